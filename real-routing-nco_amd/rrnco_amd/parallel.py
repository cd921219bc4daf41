"""Data-parallel sharding of instances over ranks (one process per GPU).  Instances are independent, so the
inference path has NO data-path collective: each rank owns all A x S rollouts of its instances.
Only the timing/throughput aggregation uses torch.distributed (RCCL on GPUs, gloo in the CPU tests)."""
from __future__ import annotations

import torch


def shard_range(n_instances: int, rank: int, world: int):
    """Contiguous [lo, hi) block of instances for `rank` (before augmentation / multistart expansion)."""
    base, rem = divmod(n_instances, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def aggregate_throughput(local_units: int, local_seconds: float, distributed: bool, device):
    """-> (units processed by all ranks, max over ranks of the elapsed time)."""
    if not distributed:
        return local_units, local_seconds
    import torch.distributed as dist
    u = torch.tensor([float(local_units)], dtype=torch.float64, device=device)
    t = torch.tensor([float(local_seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(round(u.item())), float(t.item())


ALLREDUCE_LOG = []      # (seconds | (start event, end event), bytes) of the last gradient all-reduces: train.py prints their mean with its step log


def allreduce_summary(reset: bool = True):
    """-> (mean ms per all-reduce, MB per all-reduce, count) over ALLREDUCE_LOG's finished entries, or None."""
    ms, nbytes = [], 0
    for t, b in ALLREDUCE_LOG:
        if isinstance(t, tuple):
            if not t[1].query():
                continue
            ms.append(t[0].elapsed_time(t[1]))
        else:
            ms.append(t * 1e3)
        nbytes = b
    if reset:
        ALLREDUCE_LOG.clear()
    return (sum(ms) / len(ms), nbytes / 1e6, len(ms)) if ms else None


def allreduce_flat_gradients(grads, world: int):
    """Data-parallel gradient combine as Lightning's DDPStrategy does for the reference (configs/trainer/default.yaml:12-15):
    ONE flat fp32 buffer (13.5-17.2 MB for RRNet), one sum all-reduce over RCCL / xGMI, then / world (mean).  Parameters
    without a gradient (decoder.pointer.project_out, decoder.project_fixed_context, W_placeholder under multistart:
    SURVEY App. D-9) take part as zeros so that every rank reduces the same layout."""
    import torch.distributed as dist
    if world <= 1:          # nothing to combine: no flat copy, no ~1 400 view / slice ops on the host
        return list(grads)
    flat = torch.cat([g.reshape(-1).float() for g in grads])
    if world > 1:
        if flat.is_cuda and dist.get_backend() != "nccl":      # gloo (ranks sharing a GPU in the tests): through the host
            import time
            host = flat.cpu()
            t0 = time.perf_counter()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            ALLREDUCE_LOG.append((time.perf_counter() - t0, flat.numel() * 4))
            flat.copy_(host)
        elif flat.is_cuda:                                      # RCCL: device time of the collective by events on its stream, read lazily
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            e1.record()
            ALLREDUCE_LOG.append(((e0, e1), flat.numel() * 4))
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        del ALLREDUCE_LOG[:-256]
        flat /= world
    out, off = [], 0
    for g in grads:
        n = g.numel()
        out.append(flat[off:off + n].view_as(g))
        off += n
    return out


def run_on_streams(workers, repeats: int):
    """Throughput mode on ONE GPU: `workers` (callables, each one complete pass of the hot path over one batch, each with its OWN policy
    object — the policy keeps per-call state: range-guard word, pack cache, workspaces keyed by stream) are driven by one host thread and one
    HIP stream each, `repeats` times.  What it buys: the fused VRP rollout runs one workgroup per instance until that instance's longest
    route ends (145 +- 12 decode steps at n = 100), so the last workgroups of a launch run on a mostly idle chip, and the call ends in a
    host read of the step count; a second stream's encoder / Neural-Adaptive-Bias kernels fill both holes (RCVRPTW, BASELINE configs[3]:
    +7.5 % instances/s; ATSP, whose workgroups all take the same 99 steps: +0.9 %).  The GIL is released while a thread waits for its
    stream.  Returns seconds for everything.
    State that stays process-wide (one policy object per stream does NOT isolate it): models.rollout.TIMING (a shared event list:
    switched off here for the duration), models.rollout.SPLIT_MLP and the RR_* environment switches (read by both threads alike; the
    csrc launchers read RR_MLP_SPLIT / RR_ENC_* per call).  The range guard's fp32 override (packing.force_fp32) is per thread."""
    import threading
    import time
    from .models import rollout as _R
    timing, _R.TIMING = _R.TIMING, None
    streams = [torch.cuda.Stream() for _ in workers]

    def drive(i):
        with torch.cuda.stream(streams[i]):
            for _ in range(repeats):
                workers[i]()
            streams[i].synchronize()
    threads = [threading.Thread(target=drive, args=(i,)) for i in range(len(workers))]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    _R.TIMING = timing
    return time.perf_counter() - t0
