"""world_size-2 worker on ONE GPU (gloo; ranks share device 0): the data-parallel REINFORCE step of BASELINE configs[4] as
train.py / tools/bench_train.py run it — each rank samples and differentiates ITS shard of the instances on the HIP kernels,
one flat all-reduce (mean) — checked against the ranks' own local gradients gathered on the host."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist

import helpers as H
from rrnco_amd import TensorDict
from rrnco_amd.envs import ATSPEnv
from rrnco_amd.models.rl import RRNet
from rrnco_amd.parallel import shard_range

# RR_DIST_BACKEND=nccl (a box with >= 2 GPUs): one rank per device, the flat gradient all-reduce runs on RCCL; default gloo
BACKEND = os.environ.get("RR_DIST_BACKEND", "gloo")
if BACKEND == "nccl":
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
else:
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
rank, world = dist.get_rank(), dist.get_world_size()
fx = H.load_fixture("atsp_n20_b4_pomo")
w = H.atsp_weights(fx)
pol = H.make_policy(w).train()                      # identical weights on every rank
env = ATSPEnv(generator_params=dict(num_loc=fx["N"]), check_solution=True)
model = RRNet(env, policy=pol, num_augment=8)
st = H.fixture_state(fx)
B = st["locs"].shape[0]
lo, hi = shard_range(B, rank, world)
td = TensorDict({k: v[lo:hi].cuda() for k, v in st.items()}, batch_size=[hi - lo])
td["sample_idx"] = fx["sample_idx"][lo:hi].cuda()
# local gradient (world = 1: no collective), then the same step with the all-reduce
out1 = model.training_step(td, seed=100 + rank, world=1)
local = torch.cat([p.grad.reshape(-1) for p in pol.parameters()]).cpu()
out2 = model.training_step(td, seed=100 + rank, world=world)
red = torch.cat([p.grad.reshape(-1) for p in pol.parameters()]).cpu()
assert torch.allclose(out1["log_likelihood"], out2["log_likelihood"])            # same seed -> same tours on this rank
if BACKEND == "nccl":      # collectives on device tensors
    gathered_d = [torch.zeros_like(local, device="cuda") for _ in range(world)]
    dist.all_gather(gathered_d, local.cuda())
    gathered = [g.cpu() for g in gathered_d]
else:
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
mean = torch.stack(gathered).mean(0)
err = float((red - mean).norm() / mean.norm())
assert err < 1e-4, err                              # (float atomics make two runs of the same step differ in the last bits)
assert torch.isfinite(red).all() and float(out2["grad_norm"]) > 0
acts = out2["actions"].cpu()
assert bool((acts.sort(1).values == torch.arange(fx["N"])).all())
assert torch.allclose(out2["replay_log_likelihood"], out2["log_likelihood"], rtol=2e-5, atol=2e-3)
# every rank holds the same reduced gradient
if BACKEND == "nccl":
    same_d = [torch.zeros_like(red, device="cuda") for _ in range(world)]
    dist.all_gather(same_d, red.cuda())
    same = [t.cpu() for t in same_d]
else:
    same = [torch.zeros_like(red) for _ in range(world)]
    dist.all_gather(same, red)
assert all(torch.equal(same[0], s) for s in same)
open(os.path.join(sys.argv[1], f"rank{rank}.txt"), "w").write(f"{rank} ok {err:.2e}")
dist.destroy_process_group()
