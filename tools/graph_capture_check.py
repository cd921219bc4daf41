"""Diagnostic: the whole ATSP hot path (augment, reset, encoder, persistent rollout, reward) captured into one hipGraph through
torch.cuda.graph and replayed: identical tours; no speed-up, the step is GPU-bound (24.05 vs 24.24 ms at B=64)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch, bench
from rrnco_amd.envs import ATSPEnv, ATSPGenerator
from rrnco_amd.models.encoder import ATSPInitEmbedding
dev = torch.device("cuda")
pol, w = bench.make_policy(dev)
env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
B = 64
inst_td = ATSPGenerator(num_loc=100, device=dev)(B, generator=torch.Generator(device=dev).manual_seed(1))
inst = {"locs": inst_td["locs"], "distance_matrix": inst_td["distance_matrix"]}
from rrnco_amd import TensorDict
from rrnco_amd.models.transforms import StateAugmentation
sidx = ATSPInitEmbedding.sample_indices(env.reset(inst_td)["distance_matrix"], 25).repeat(8, 1, 1).contiguous()


def step(pol, env, inst, sidx):
    """bench.hot_path_step with the neighbour sample pinned (its seed is a host-side draw: not capturable)."""
    td = StateAugmentation(augment_fn="dihedral8", no_aug_coords=False)(TensorDict(dict(inst), batch_size=[inst["locs"].shape[0]]))
    td = env.reset(td)
    td.set("sample_idx", sidx)
    out = pol(td, env, phase="val", decode_type="multistart_greedy", num_starts=bench.STARTS, return_actions=True)
    return out["reward"], out


best, out = step(pol, env, inst, sidx); torch.cuda.synchronize()
ref = out["actions"].clone()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    step(pol, env, inst, sidx)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        best2, out2 = step(pol, env, inst, sidx)
torch.cuda.synchronize()
out2["actions"].zero_()
g.replay(); torch.cuda.synchronize()
print("graph replay identical:", bool(torch.equal(out2["actions"], ref)))
t0 = time.perf_counter()
for _ in range(5): g.replay()
torch.cuda.synchronize(); print("graph ms/step", (time.perf_counter() - t0) / 5 * 1e3)
t0 = time.perf_counter()
for _ in range(5): step(pol, env, inst, sidx)
torch.cuda.synchronize(); print("eager ms/step", (time.perf_counter() - t0) / 5 * 1e3)
