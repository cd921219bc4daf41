"""Diagnostic: dump the greedy tours of the headline workload at a small batch (to compare kernel variants across processes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch
import bench
from rrnco_amd.envs import ATSPEnv, ATSPGenerator
from rrnco_amd.models.encoder import ATSPInitEmbedding
dev = torch.device("cuda")
pol, w = bench.make_policy(dev)
env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
inst_td = ATSPGenerator(num_loc=100, device=dev)(B, generator=torch.Generator(device=dev).manual_seed(1))
inst = {"locs": inst_td["locs"], "distance_matrix": inst_td["distance_matrix"]}
sidx = ATSPInitEmbedding.sample_indices(env.reset(inst_td)["distance_matrix"], 25, generator=torch.Generator(device=dev).manual_seed(2)) \
    if "generator" in ATSPInitEmbedding.sample_indices.__code__.co_varnames else None
if sidx is None:
    torch.manual_seed(2)
    sidx = ATSPInitEmbedding.sample_indices(env.reset(inst_td)["distance_matrix"], 25)
sidx = sidx.repeat(8, 1, 1).contiguous()
best, out = bench.hot_path_step(pol, env, inst, sidx)
torch.save({"actions": out["actions"].cpu(), "reward": out["reward"].cpu(), "ll": out["log_likelihood"].cpu()}, sys.argv[1])
print("saved", sys.argv[1], float(best.mean()))
