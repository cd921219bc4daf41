"""Diagnostic: per-phase cycle shares of k_rollout_w (stamped build). Not part of the product."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
from rrnco_amd import _lib
_lib.LIB_PATH = os.environ.get("RR_STAMP_LIB") or _lib.LIB_PATH.replace("librrnco_hip.so", "librrnco_hip_stamp.so")
import torch
import bench
from rrnco_amd.envs import ATSPEnv, ATSPGenerator
from rrnco_amd.models.encoder import ATSPInitEmbedding
dev = torch.device("cuda")
pol, w = bench.make_policy(dev)
env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
inst_td = ATSPGenerator(num_loc=100, device=dev)(B, generator=torch.Generator(device=dev).manual_seed(1))
inst = {"locs": inst_td["locs"], "distance_matrix": inst_td["distance_matrix"]}
sidx = ATSPInitEmbedding.sample_indices(env.reset(inst_td)["distance_matrix"], 25).repeat(8, 1, 1).contiguous()
bench.hot_path_step(pol, env, inst); torch.cuda.synchronize()
lib = _lib.lib(); lib.rr_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
out = (ctypes.c_ulonglong * 8)()
lib.rr_debug_stamps(out, 1)
lib.rr_debug_enc_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
eout = (ctypes.c_ulonglong * 8)()
lib.rr_debug_enc_stamps(eout, 1)
wout = (ctypes.c_ulonglong * 8)()
lib.rr_debug_wave_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.rr_debug_wave_cycles(wout, 1)
from rrnco_amd.models import rollout as _R
_R.TIMING = []
torch.manual_seed(7)
_best, _out = bench.hot_path_step(pol, env, inst); torch.cuda.synchronize()
_a = _out["actions"].long()
print("checksum: best", float(_best.double().sum()), "actions", int((_a * torch.arange(1, _a.shape[1] + 1, device=_a.device)).sum() % 1000000007),
      "ll", float(_out["log_likelihood"].double().sum()) if "log_likelihood" in _out else None)
print("rollout kernel ms:", [round(a.elapsed_time(b), 2) for a, b in _R.TIMING]); _R.TIMING = None
lib.rr_debug_stamps(out, 0)
names = ["loop-top", "ctx gather", "attention", "MLP", "logits MFMA", "select+step"]
waves = max(out[7], 1); tot = max(sum(out[i] for i in range(6)), 1)
print(f"waves={waves} total cycles/wave={tot/waves:.3e} per step={tot/waves/99:.0f}")
for i, n in enumerate(names):
    print(f"  {n:12s} {out[i]/waves/99:10.0f} cycles/step  {100*out[i]/tot:5.1f}%")
lib.rr_debug_wave_cycles(wout, 0)
print("  per wave number, cycles/step:", [round(wout[i] / (B * 8) / 99) for i in range(8)])
print(f"  (of the MLP: waiting at its stage barriers {out[6]/max(waves,1)/99:10.0f} cycles/step)")

lib.rr_debug_enc_stamps(eout, 0)
en = (["norm2,K,softmaxK,V", "NAB", "norm1,Q,mix", "P,M,norm3,normf1", "FFN+norm"] if os.environ.get("RR_ENC_VARIANT", "1") == "1"
      else ["load+norm1/2", "NAB+softmax", "KV,softmaxK,den,num", "Q,Y,P,M,norms", "FFN+norm"])
if os.environ.get("RR_ENC_FINE"):
    en = ["NAB stage+barrier", "NAB input wait", "NAB eval x7", "NAB softmax", "-", "KV stage", "rest"]
en = en + ["  (of phase 0) load + norm2", "  (of phase 0) K projection"]
ew = eout[7]; et = sum(eout[i] for i in range(7))
print(f"encoder block: waves={ew} cycles/wave/block={et/ew:.0f}")
for i, n in enumerate(en):
    print(f"  {n:28s} {eout[i]/ew:10.0f} cycles  {100*eout[i]/et:5.1f}%")
