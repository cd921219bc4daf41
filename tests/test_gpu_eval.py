"""`-m gpu`: the test.py-shaped evaluation driver (evaluate.py) on npz datasets in the reference's schema, all three problems."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import restate
from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _write(tmp_path, problem, n, count):
    if problem == "atsp":
        inst = restate.atsp_synthetic(count, n, 7)
    elif problem == "rcvrp":
        inst = restate.rcvrp_synthetic(count, n, 7)
        inst["demand"] = inst["demand"] * 50.0
        inst["capacity"] = torch.full((count,), 50.0)
    else:
        inst = restate.rcvrptw_synthetic(count, n, 7)
        inst["speed"] = torch.ones(count, 1)                      # present in generate_data.py:354-372 files, unused
        inst["vehicle_capacity"] = torch.ones(count, 1)
    p = str(tmp_path / f"{problem}{n}.npz")
    np.savez(p, **{k: v.numpy() for k, v in inst.items()})
    return p


@pytest.mark.parametrize("problem", ["atsp", "rcvrp", "rcvrptw"])
def test_evaluate_driver_matches_direct_policy_call(tmp_path, problem):
    import evaluate
    from rrnco_amd import data
    n, count = 20, 6
    path = _write(tmp_path, problem, n, count)
    dev = torch.device("cuda:0")
    policy, env = evaluate.build(problem, None, n, dev, seed=5)
    S = n if problem != "rcvrp" else n + 1
    lines = []
    avg, times = evaluate.evaluate_dataset(path, problem, policy, env, batch_size=4, n_aug=1, n_start=S, device=dev, log=lines.append)
    assert len(times) == 2 and lines[0].startswith("Average cost:") and np.isfinite(avg) and avg > 0
    # same number from one direct call on the whole set (the neighbour sample is drawn on device, so pin it in both)
    td = data.prepare_for_env(data.load_npz_to_tensordict(path), problem).to(dev)
    out = policy(env.reset(td), env, phase="val", num_starts=S)
    best = out["reward"].view(S, count).max(0).values
    assert abs(float(-best.mean()) - avg) / avg < 0.05          # different random neighbour samples: close, not equal


@pytest.mark.parametrize("problem", ["atsp", "rcvrp", "rcvrptw"])
def test_evaluate_driver_hipgraph_replay_gives_the_eager_costs(tmp_path, problem, monkeypatch):
    """--hipgraph: the policy call of each batch shape captured once and replayed per batch (two full batches + a ragged one = two
    graphs).  With the neighbour sample pinned the replayed costs are the eager ones."""
    import evaluate
    from rrnco_amd.models.encoder import ATSPInitEmbedding
    n, count = 20, 10
    path = _write(tmp_path, problem, n, count)
    dev = torch.device("cuda:0")
    policy, env = evaluate.build(problem, None, n, dev, seed=5)
    S = n if problem != "rcvrp" else n + 1

    def fixed(distance, k):                      # (deterministic stand-in for the multinomial draw: the k nearest columns by index order)
        B, N = distance.shape[0], distance.shape[-1]
        return (torch.arange(k, device=distance.device)[None, None, :] + torch.arange(N, device=distance.device)[None, :, None] + 1).remainder(N).expand(B, N, k).contiguous()
    monkeypatch.setattr(ATSPInitEmbedding, "sample_indices", staticmethod(fixed))
    a = evaluate.evaluate_dataset(path, problem, policy, env, batch_size=4, n_aug=8, n_start=S, device=dev, log=lambda *_: None)[0]
    b = evaluate.evaluate_dataset(path, problem, policy, env, batch_size=4, n_aug=8, n_start=S, device=dev, log=lambda *_: None, hipgraph=True)[0]
    assert np.isfinite(a) and abs(a - b) <= 1e-6 * abs(a), (a, b)


def test_evaluate_cli_with_checkpoint_and_augmentation(tmp_path):
    import evaluate
    path = _write(tmp_path, "atsp", 20, 3)
    w = H.atsp_weights(15, layers=6, seed=12)
    ck = str(tmp_path / "epoch_199.ckpt")
    torch.save({"state_dict": {"policy." + k: v for k, v in w.items()}}, ck)
    res = evaluate.main(["--problem", "atsp", "--datasets", path, "--checkpoint", ck, "--problem_size", "20", "--batch_size", "2"])
    assert list(res) == [path] and np.isfinite(res[path])
    res_na = evaluate.main(["--problem", "atsp", "--datasets", path, "--checkpoint", ck, "--problem_size", "20", "--no_aug"])
    assert res[path] <= res_na[path] + 1e-2                       # best-of-8 augmentations cannot be (noticeably) worse


def test_device_sampler_matches_reference_indexing_bit_exact():
    """rr_submatrix_gather == data[key][idx[:, :, None], idx[:, None, :]] (atsp/sampler.py:83-90, rmtvrp/sampler.py:80)."""
    from rrnco_amd.envs import ATSPEnv, RealWorldSampler
    g = torch.Generator().manual_seed(4)
    M, B, n = 700, 33, 100
    city = {"points": torch.rand(M, 2, generator=g).numpy(), "distance": (torch.rand(M, M, generator=g) * 5e3).numpy(),
            "duration": (torch.rand(M, M, generator=g) * 9e2).numpy()}
    smp = RealWorldSampler(with_duration=True)
    smp.load_city(city)
    out = smp.sample(B, n, "uniform", generator=torch.Generator(device="cuda").manual_seed(1))
    # recover the indices from the points (all distinct) and redo the slicing the reference's way on the host
    pts = torch.from_numpy(city["points"])
    idx = torch.cdist(out["points"].cpu().reshape(-1, 2), pts).argmin(1).view(B, n)
    assert all(len(set(r.tolist())) == n for r in idx)
    assert torch.equal(out["points"].cpu(), pts[idx])
    for key in ("distance", "duration"):
        ref = torch.from_numpy(city[key])[idx[:, :, None], idx[:, None, :]]
        assert torch.equal(out[key + "_matrix"].cpu(), ref)
    for mode in ("mixed", "single_cluster"):
        o = smp.sample(5, 20, mode, generator=torch.Generator(device="cuda").manual_seed(2))
        assert o["distance_matrix"].shape == (5, 20, 20) and bool((torch.diagonal(o["distance_matrix"], dim1=1, dim2=2) >= 0).all())
    # the sampled batch feeds the env directly
    from rrnco_amd import TensorDict
    td = ATSPEnv(generator_params=dict(num_loc=n)).reset(
        TensorDict({"locs": out["points"], "distance_matrix": out["distance_matrix"]}, batch_size=[B]))
    assert td["distance_matrix"].shape == (B, n, n) and float(td["distance_matrix"].max()) <= 1.0
    with pytest.raises(ValueError):
        smp.sample(2, M + 1)


def test_bench_multi_process_contract_two_ranks_on_one_gpu(tmp_path):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one JSON line from rank 0, aggregate over ranks),
    with two ranks sharing this box's GPU over gloo (RR_DIST_BACKEND): exercises barrier / SUM-of-units / MAX-of-time."""
    import json
    import subprocess
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "64"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env={**os.environ, "RR_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                            # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["unit"] == "instances/s" and out["value"] > 0 and abs(out["value"] * out["ms_per_step"] / 1e3 - 128) < 1e-6 * 128
    assert out["roofline"]["bound"] == "mfma" and "cpu_baseline" not in out


@pytest.mark.parametrize("problem", ["atsp", "rcvrp", "rcvrptw"])
def test_train_driver_writes_checkpoints_that_evaluate_reads(tmp_path, problem):
    """train.py (configs/experiment/rrnet.yaml hyper-parameters, tiny sizes) -> Lightning-layout checkpoint -> resume ->
    evaluate.py on an npz set; the policy must have moved and the loop must stay finite."""
    import evaluate
    import train
    ck = str(tmp_path / "ck")
    common = ["--problem", problem, "--problem_size", "20", "--batch_size", "8", "--train_data_size", "32", "--val_data_size", "8",
              "--checkpoint_dir", ck, "--log_every", "2"]
    v1 = train.main(common + ["--epochs", "1"])
    last = os.path.join(ck, problem, "last.ckpt")
    blob = torch.load(last, map_location="cpu", weights_only=False)
    assert blob["epoch"] == 0 and all(k.startswith("policy.") for k in blob["state_dict"]) and np.isfinite(v1)
    v2 = train.main(common + ["--epochs", "2", "--resume", last])                  # continues at epoch 1
    assert torch.load(last, map_location="cpu", weights_only=False)["epoch"] == 1 and np.isfinite(v2)
    moved = sum(int(not torch.equal(blob["state_dict"][k], v)) for k, v in
                torch.load(last, map_location="cpu", weights_only=False)["state_dict"].items())
    assert moved > 100
    path = _write(tmp_path, problem, 20, 4)
    res = evaluate.main(["--problem", problem, "--datasets", path, "--checkpoint", last, "--problem_size", "20", "--batch_size", "4", "--no_aug"])
    assert np.isfinite(res[path])
