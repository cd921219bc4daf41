"""Diagnostic: top GPU kernels of one step of BASELINE configs[3] (RCVRPTW n=100, B=256 x 8 augmentations, S=100 sampling;
PROBLEM=rcvrp: configs[2]) by torch.profiler."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from rrnco_amd import TensorDict
from rrnco_amd.envs import RCVRPEnv, RMTVRPEnv
from rrnco_amd.models import RRNetPolicy
from rrnco_amd.models.encoder import ATSPInitEmbedding
from rrnco_amd.models.transforms import StateAugmentation
dev = torch.device("cuda")
PROBLEM = os.environ.get("PROBLEM", "rcvrptw")
torch.manual_seed(1234)
pol = RRNetPolicy(env_name=PROBLEM, embed_dim=128, num_heads=8, num_encoder_layers=6, normalization="instance",
                  use_graph_context=False, nab_type="gating", init_embedding_kwargs=dict(sample_size=25)).to(dev).eval()
if PROBLEM == "rcvrptw":
    env, B, S, aug, decode = RMTVRPEnv(generator_params=dict(num_loc=100, device=dev), device=dev), 256, 100, True, "multistart_sampling"
else:
    env, B, S, aug, decode = RCVRPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev), 512, 101, False, "multistart_greedy"
inst = env.generator(B, generator=torch.Generator(device=dev).manual_seed(5))
sidx = ATSPInitEmbedding.sample_indices(env.reset(inst)["distance_matrix"], 25)
if aug:
    sidx = sidx.repeat(8, 1, 1).contiguous()
def step():
    td = TensorDict(dict(inst.items()), batch_size=[B])
    if aug:
        td = StateAugmentation(augment_fn="dihedral8", no_aug_coords=False)(td)
    td["sample_idx"] = sidx
    return pol(env.reset(td), env, phase="val", decode_type=decode, num_starts=S, seed=1)
for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    step(); torch.cuda.synchronize()
ev = sorted([e for e in prof.key_averages() if e.self_device_time_total > 0], key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in ev)
print(f"GPU time of one {PROBLEM} step: {tot / 1e3:.2f} ms over {sum(e.count for e in ev)} launches")
for e in ev[:22]:
    print(f"{e.self_device_time_total / 1e3:9.3f} ms {100 * e.self_device_time_total / tot:5.1f}%  x{e.count:<4d} {e.key[:120]}")
