"""Diagnostic: launch time of rr_dec_attn_bwd alone at the shape of BASELINE configs[4] (512 ATSP n=100 instances, 100 sampled
starts), for the library and any number of variant builds:  python tools/attn_bwd_time.py [librrnco_hip_<name>.so ...]
(RR_ATTN_BWD_F32=1 in the environment: the fp32-MFMA kernel)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch, bench
from rrnco_amd import TensorDict, _lib as L
from rrnco_amd.envs import ATSPEnv
from rrnco_amd.models import dec_backward
dev = torch.device("cuda")
B = int(os.environ.get("PB", "512"))
pol, _ = bench.make_policy(dev)
pol.train()
env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
td = env.reset(env.generator(B, generator=torch.Generator(device=dev).manual_seed(1)))
cap = {}
with torch.no_grad():
    pol(td, env, phase="train", decode_type="multistart_sampling", num_starts=100, seed=5, capture=cap)
held = {}
real = L.lib().rr_dec_attn_bwd
def spy(ia, st):
    held["ia"], held["st"] = ia, st
    return real(ia, st)
L.lib().rr_dec_attn_bwd = spy
S, Bp = 100, td["distance_matrix"].shape[0]
gll = torch.randn(S * Bp, device=dev) / (S * Bp)
res = dec_backward.decoder_backward(pol, cap["cache"], cap["dump"], td["distance_matrix"].contiguous(), None, gll)
L.lib().rr_dec_attn_bwd = real
torch.cuda.synchronize()
ref = {k: res[k].clone() for k in ("dK", "dV", "dctxA", "dctxB")}
csrc = os.path.dirname(L.LIB_PATH)
for name in ["librrnco_hip.so"] + sys.argv[1:]:
    lib = C.CDLL(os.path.join(csrc, name))
    fn = lib.rr_dec_attn_bwd
    fn.argtypes, fn.restype = [C.POINTER(L.DecAttnIO), C.c_void_p], C.c_int
    for mode in (("0", "1") if name == "librrnco_hip.so" else ("0",)):
        os.environ["RR_ATTN_BWD_F32"] = mode
        fn(held["ia"], held["st"]); torch.cuda.synchronize()
        err = max(float((res[k] - ref[k]).abs().max() / ref[k].abs().max()) for k in ref)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn(held["ia"], held["st"])
        e1.record(); torch.cuda.synchronize()
        print(f"{name:36s} {'fp32 MFMA' if mode == '1' else 'default  '}  {e0.elapsed_time(e1) / 5:7.3f} ms   max rel diff to the first run {err:.2e}")
