"""Diagnostic: torch ops dispatched by ONE inference step of the headline workload (bench.hot_path_step), by source line."""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from rrnco_amd.envs import ATSPEnv, ATSPGenerator
dev = torch.device("cuda")
pol, w = bench.make_policy(dev)
env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
inst_td = ATSPGenerator(num_loc=100, device=dev)(512, generator=torch.Generator(device=dev).manual_seed(1))
inst = {"locs": inst_td["locs"], "distance_matrix": inst_td["distance_matrix"]}
for _ in range(2): bench.hot_path_step(pol, env, inst)
torch.cuda.synchronize()
cnt = collections.Counter()
class M(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        fr = "?"
        for f in reversed(traceback.extract_stack(limit=14)):
            if "rrnco_amd" in f.filename or f.filename.endswith("bench.py"):
                fr = f"{os.path.basename(f.filename)}:{f.lineno}"; break
        cnt[(str(func), fr)] += 1
        return func(*args, **(kwargs or {}))
with M():
    bench.hot_path_step(pol, env, inst)
torch.cuda.synchronize()
print("ops dispatched in one inference step:", sum(cnt.values()))
for (f, fr), n in cnt.most_common(60): print(f"  x{n:<4d} {f:45s} {fr}")
