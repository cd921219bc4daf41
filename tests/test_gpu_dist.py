"""`-m gpu`: the N > 1 paths on a single-GPU box — two ranks sharing device 0 over gloo (RR_DIST_BACKEND=gloo): the launch
contract of bench.py (`torch.distributed.run`, barrier + max-over-ranks timing, one JSON line from rank 0), the
data-parallel REINFORCE step with its flat gradient all-reduce, and tools/bench_train.py at BASELINE configs[4] size.  RCCL
itself needs >= 2 GPUs (the driver's scaling run); everything around the collective is exercised here."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(script_args, timeout=900):
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    env = {**os.environ, "RR_DIST_BACKEND": "gloo", "OMP_NUM_THREADS": "4", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_bench_contract_under_torchrun_two_ranks():
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                    # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert abs(out["value"] - 2 * 64 * 2 / (out["ms_per_step"] * 2e-3)) / out["value"] < 1e-6     # all ranks' units / max time
    assert len(out["devices"]) == 2 and out["devices"][0].startswith("rank 0: cuda:") and "configs" not in out
    # strong scaling: the 64 instances of ONE batch partitioned over the two ranks
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64", "--scaling", "strong",
                   "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["scaling"] == "strong" and out["config"]["rollouts_per_gpu"] == 32 * 8 * 100
    assert abs(out["value"] - 64 * 2 / (out["ms_per_step"] * 2e-3)) / out["value"] < 1e-6


def test_sharded_training_step_gradient_all_reduce(tmp_path):
    r = _torchrun([os.path.join(ROOT, "tests", "dist_worker_gpu.py"), str(tmp_path)])
    assert r.returncode == 0, r.stderr[-3000:]
    assert sorted(os.listdir(tmp_path)) == ["rank0.txt", "rank1.txt"]


def test_bench_train_full_size_two_ranks():
    """BASELINE configs[4] per-GPU size (512 instances x 100 sampled starts per rank), two ranks: finite loss and gradient norm."""
    r = _torchrun([os.path.join(ROOT, "tools", "bench_train.py"), "--steps", "1"], timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["loss"] == out["loss"] and 0 < out["grad_norm"] < 1e4


def test_rccl_paths_when_two_devices_are_visible(tmp_path):
    """On a box with >= 2 GPUs the nccl (= RCCL) path of allreduce_flat_gradients and aggregate_throughput runs under
    torch.distributed.run, one rank per device: the sharded training gradient must equal the single-rank one and bench.py must
    report both ranks' devices.  Skipped on the single-GPU test boxes (the gloo twin above covers the logic there)."""
    import json
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (RCCL over xGMI); single-GPU box")
    env = dict(os.environ, RR_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29613"]
    r = subprocess.run(cmd + [os.path.join(ROOT, "tests", "dist_worker_gpu.py"), str(tmp_path)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run(cmd + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and len(line["devices"]) == 2 and "cuda:0" in line["devices"][0] and "cuda:1" in line["devices"][1]
    assert len({d.split(":", 1)[1].split("(")[0].strip() for d in line["devices"]}) == 2          # two DISTINCT devices, one per rank
    assert abs(line["value"] - 2 * 64 * 2 / (line["ms_per_step"] * 2e-3)) / line["value"] < 1e-6   # whole-job units / max-over-ranks time
    # strong scaling over RCCL: ONE batch of 64 partitioned by parallel.shard_range
    r = subprocess.run(cmd + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64", "--scaling", "strong",
                              "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["config"]["rollouts_per_gpu"] == 32 * 8 * 100 and len(line["devices"]) == 2
