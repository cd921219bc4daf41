"""Diagnostic: per-phase cycle shares of the SAMPLING rollout with the training dump (stamped build), one ATSP training step of
512 instances.  Not part of the product."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
from rrnco_amd import _lib
_lib.LIB_PATH = os.environ.get("RR_STAMP_LIB") or _lib.LIB_PATH.replace("librrnco_hip.so", "librrnco_hip_stamp.so")
import torch
import bench
from rrnco_amd.envs import ATSPEnv
from rrnco_amd.models.rl import RRNet
dev = torch.device("cuda")
pol, w = bench.make_policy(dev)
env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
pol.train()
model = RRNet(env, policy=pol)
opt = torch.optim.Adam(pol.parameters(), lr=1e-4, fused=True)
gen = torch.Generator(device=dev).manual_seed(1234)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
batches = [env.generator(B, generator=gen) for _ in range(2)]
model.training_step(batches[0], optimizer=opt, world=1, seed=1, grad_clip=1.0); torch.cuda.synchronize()
lib = _lib.lib(); lib.rr_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
out = (ctypes.c_ulonglong * 8)()
lib.rr_debug_stamps(out, 1)
model.training_step(batches[1], optimizer=opt, world=1, seed=2, grad_clip=1.0); torch.cuda.synchronize()
lib.rr_debug_stamps(out, 0)
names = ["loop-top", "ctx gather", "attention", "MLP", "logits MFMA", "select+step"]
waves = out[7]; tot = sum(out[i] for i in range(6))
print(f"waves={waves} total cycles/wave={tot/waves:.3e} per step={tot/waves/99:.0f}")
for i, n in enumerate(names):
    print(f"  {n:12s} {out[i]/waves/99:10.0f} cycles/step  {100*out[i]/tot:5.1f}%")
