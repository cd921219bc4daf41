"""Diagnostic: greedy tours of the headline workload at a small batch under another build of the library, to compare kernel
variants across processes:  python tools/ab_actions.py <out.pt> [batch] [librrnco_hip_<name>.so]; then torch.load both and compare."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
from rrnco_amd import _lib
if len(sys.argv) > 3:
    _lib.LIB_PATH = _lib.LIB_PATH.replace("librrnco_hip.so", sys.argv[3])
import torch
import bench
from rrnco_amd.envs import ATSPEnv, ATSPGenerator
dev = torch.device("cuda")
pol, w = bench.make_policy(dev)
env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
inst_td = ATSPGenerator(num_loc=100, device=dev)(B, generator=torch.Generator(device=dev).manual_seed(1))
inst = {"locs": inst_td["locs"], "distance_matrix": inst_td["distance_matrix"]}
outs = []
for k in range(3):
    torch.manual_seed(2)                                   # the step's neighbour sample
    best, out = bench.hot_path_step(pol, env, inst)
    outs.append((out["actions"].cpu(), out["log_likelihood"].cpu()))
print("repeats identical:", all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:]))
torch.save({"actions": outs[0][0], "ll": outs[0][1]}, sys.argv[1])
print("saved", sys.argv[1], "mean best cost", float(-best.mean()))
