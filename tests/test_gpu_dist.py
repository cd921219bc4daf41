"""`-m gpu`: the N > 1 paths on a single-GPU box — two ranks sharing device 0 over gloo (RR_DIST_BACKEND=gloo): the launch
contract of bench.py (`torch.distributed.run`, barrier + max-over-ranks timing, one JSON line from rank 0), the
data-parallel REINFORCE step with its flat gradient all-reduce, and tools/bench_train.py at BASELINE configs[4] size.  RCCL
itself needs >= 2 GPUs (the driver's scaling run); everything around the collective is exercised here."""
import json
import os
import subprocess
import sys

import pytest

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
