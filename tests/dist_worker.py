"""world_size-2 gloo worker: exercises the N>1 sharding/aggregation helpers bench.py uses."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch
import torch.distributed as dist

from rrnco_amd.parallel import aggregate_throughput, allreduce_flat_gradients, shard_range

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lo, hi = shard_range(1001, rank, world)
cover = torch.zeros(1001); cover[lo:hi] = 1
dist.all_reduce(cover)
assert bool((cover == 1).all()), "shards must partition the instances exactly once"
units, tmax = aggregate_throughput(hi - lo, 0.5 + 0.25 * rank, world > 1, torch.device("cpu"))
assert units == 1001 and abs(tmax - 0.75) < 1e-9
g = [torch.full((3, 2), float(rank + 1)), torch.zeros(5), torch.arange(4.0) * (rank + 1)]
red = allreduce_flat_gradients(g, world)
assert torch.allclose(red[0], torch.full((3, 2), 1.5)) and torch.equal(red[1], torch.zeros(5)) and torch.allclose(red[2], torch.arange(4.0) * 1.5)
# CPU tensors over gloo are not timed; the device paths are (train.py prints allreduce_summary() with its step log: the first real
# multi-GPU run yields the all-reduce time without a code change)
from rrnco_amd.parallel import allreduce_summary
assert allreduce_summary() is None

# ---- data-parallel REINFORCE gradient (BASELINE configs[4]): each rank replays ITS shard of instances with the loss mean
# taken over its own rollouts, one flat all-reduce (mean) — must equal the single-process gradient over all instances
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
from oracle import restate
from rrnco_amd.models.grad_replay import replay_backward

fx = H.load_fixture("atsp_n20_b4_pomo")
w = H.atsp_weights(fx)
S, N, B = fx["S"], fx["N"], fx["B"]
st0 = restate.atsp_reset(H.fixture_state(fx))
adv = torch.from_numpy(np.random.default_rng(5).standard_normal(S * B).astype(np.float32))
pol = H.make_policy(w, device="cpu")
pol.zero_grad()
replay_backward(pol, {"distance_matrix": st0["distance_matrix"], "locs": st0["locs"]}, fx["actions"], S, -adv / (S * B), fx["sample_idx"])
full = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in pol.parameters()]
lo, hi = shard_range(B, rank, world)
pol.zero_grad()
acts = fx["actions"].view(S, B, N)[:, lo:hi].reshape(-1, N)
gl = (-adv.view(S, B)[:, lo:hi] / (S * (hi - lo))).reshape(-1)
replay_backward(pol, {"distance_matrix": st0["distance_matrix"][lo:hi], "locs": st0["locs"][lo:hi]}, acts, S, gl, fx["sample_idx"][lo:hi])
red = allreduce_flat_gradients([p.grad if p.grad is not None else torch.zeros_like(p) for p in pol.parameters()], world)
num = sum(float(((a - b) ** 2).sum()) for a, b in zip(red, full)) ** 0.5
den = sum(float((b ** 2).sum()) for b in full) ** 0.5
assert num / den < 2e-3, num / den          # fp32 summation order (and at most a ReLU-kink flip, see tests/test_cpu.py)
open(os.path.join(sys.argv[1], f"rank{rank}.txt"), "w").write(f"{rank} {units} {tmax}")
dist.destroy_process_group()
