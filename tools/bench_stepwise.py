"""Diagnostic: the reference-shaped per-step decode loop (policy(..., fused=False)) on the headline workload, for the
HBM-roofline figures of the per-step kernels (10N+12 = 1012 B per rollout-step for ATSP).  Run under
rocprofv3 --kernel-trace --stats to get the per-kernel durations; prints the step time itself."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch
import bench
from rrnco_amd import TensorDict
from rrnco_amd.envs import ATSPEnv, ATSPGenerator
from rrnco_amd.models.encoder import ATSPInitEmbedding
from rrnco_amd.models.transforms import StateAugmentation
dev = torch.device("cuda")
pol, w = bench.make_policy(dev)
env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
inst = ATSPGenerator(num_loc=100, device=dev)(B, generator=torch.Generator(device=dev).manual_seed(1))
sidx = ATSPInitEmbedding.sample_indices(env.reset(inst)["distance_matrix"], 25).repeat(8, 1, 1).contiguous()
def step(fused):
    td = StateAugmentation(augment_fn="dihedral8", no_aug_coords=False)(TensorDict(dict(inst.items()), batch_size=[B]))
    td["sample_idx"] = sidx
    return pol(env.reset(td), env, phase="val", decode_type="multistart_greedy", num_starts=100, fused=fused)
for fused in (False, True):
    step(fused); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = step(fused); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"fused={fused}: {dt*1e3:.1f} ms per batch of {B} instances -> {B/dt:.0f} instances/s; rollouts {out['actions'].shape[0]}")
