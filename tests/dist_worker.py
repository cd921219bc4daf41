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
open(os.path.join(sys.argv[1], f"rank{rank}.txt"), "w").write(f"{rank} {units} {tmax}")
dist.destroy_process_group()
