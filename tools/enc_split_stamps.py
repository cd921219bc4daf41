"""Diagnostic: per-phase cycles of k_enc_mix / k_enc_tail (stamped build, csrc/librrnco_hip_stamp.so) at the headline shape. Not product."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
from rrnco_amd import _lib
_lib.LIB_PATH = os.environ.get("RR_STAMP_LIB") or _lib.LIB_PATH.replace("librrnco_hip.so", "librrnco_hip_stamp.so")
import torch
import bench
from rrnco_amd.envs import ATSPEnv, ATSPGenerator
from rrnco_amd.models.encoder import ATSPInitEmbedding
from rrnco_amd.models.transforms import StateAugmentation
dev = torch.device("cuda")
pol, w = bench.make_policy(dev)
env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
td = StateAugmentation(num_augment=8)(env.reset(ATSPGenerator(num_loc=100, device=dev)(B, generator=torch.Generator(device=dev).manual_seed(1))))
td.set("sample_idx", ATSPInitEmbedding.sample_indices(td["distance_matrix"], 25).contiguous())
packed = pol.packed(dev)
pol.encoder(td, packed=packed); torch.cuda.synchronize()
lib = _lib.lib(); lib.rr_debug_split_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
out = (ctypes.c_ulonglong * 32)()
lib.rr_debug_split_stamps(out, 1)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record(); pol.encoder(td, packed=packed); ev[1].record(); torch.cuda.synchronize()
lib.rr_debug_split_stamps(out, 0)
print(f"encoder pass (stamped build) {ev[0].elapsed_time(ev[1]):.2f} ms")
for base, name, names in ((0, "k_enc_mix", ["NAB + row softmax", "K/V loads + max fold", "exp + sum fold", "images + node sums", "image barrier", "mixing + ratio store", "pass barrier"]),
                          (16, "k_enc_tail", ["loads + norm1 + split", "start barrier", "Q projection + sigmoid", "P projection", "norm3", "ffn.norm1", "FFN", "ffn.norm2", "store", "output statistics"])):
    nw = max(out[base + 15], 1); tot = sum(out[base + i] for i in range(len(names)))
    print(f"{name}: waves {nw}, cycles per wave {tot / nw:.0f} (s_memtime ticks)")
    for i, n in enumerate(names):
        print(f"   {n:26s} {out[base + i] / nw:9.0f}  {100 * out[base + i] / max(tot, 1):5.1f} %")
