"""Diagnostic: launch time of rr_nab_hist_bwd at the training shape (512 instances x 100 x 100 edges), for the library and variant
builds: python tools/nab_hist_time.py [librrnco_hip_<name>.so ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch, bench
from rrnco_amd import _lib as L
dev = torch.device("cuda")
pol, _ = bench.make_policy(dev)
packed = pol.packed(dev)
nab = packed["blocks"][0][0].nab
M = 512 * 100 * 100
g = torch.Generator(device=dev).manual_seed(1)
xd = torch.rand(M, device=dev, generator=g); xa = (torch.rand(M, device=dev, generator=g) - 0.5) * 6.28
go = torch.randn(M, device=dev, generator=g) * 1e-4
csrc = os.path.dirname(L.LIB_PATH)
ref = None
for name in ["librrnco_hip.so"] + sys.argv[1:]:
    lib = C.CDLL(os.path.join(csrc, name))
    fn = lib.rr_nab_hist_bwd
    fn.argtypes, fn.restype = [C.c_void_p] * 5 + [C.c_long, C.c_void_p], C.c_int
    hist = torch.zeros(2 * 129 * 4 + 1, device=dev)
    fn(nab, xd.data_ptr(), xa.data_ptr(), go.data_ptr(), hist.data_ptr(), M, L.stream()); torch.cuda.synchronize()
    if ref is None: ref = hist.clone()
    err = float((hist - ref).abs().max() / ref.abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn(nab, xd.data_ptr(), xa.data_ptr(), go.data_ptr(), hist.data_ptr(), M, L.stream())
    e1.record(); torch.cuda.synchronize()
    print(f"{name:30s} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us   max rel diff to the library {err:.2e}")
