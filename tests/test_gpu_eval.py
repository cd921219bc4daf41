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
