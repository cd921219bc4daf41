"""Headline-shaped calls of BASELINE.json configs[2] and [3] and of the trained ATSP policy against the oracle run LIVE
(VERDICT r04, next #6): the goldens of the VRPs are B = 2 .. 4 at N = 100, the full sizes are property-checked only.  Here a few
instances go through the policy exactly as bench.py's C3 / C4 steps drive it and oracle/restate.py evaluates the same instances:

  C3  RCVRP n = 100, S = 101 greedy POMO: tours against the oracle's, a parting only where the oracle's own top-1 / top-2 gap is
      below GAP_TOL; costs and log-likelihoods of the tours kept.
  C4  RCVRPTW n = 100, x8 augmentation, S = 100 SAMPLING: the engine's sampled routes are teacher-forced through the oracle's
      evaluate mode (decoding.py:386-399) — every step's log-probability of the action the engine took, the route costs, and that
      the oracle's feasibility mask admits every sampled action.
  ATSP on the weights of tests/golden/atsp_trained_weights.npz at the headline shape (x8 aug, S = 100 greedy), gap rule.
GAP_TOL = 3e-4 (the largest gap at a parting observed on any build is 1.5e-4)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import restate
from tests import helpers as H

pytestmark = pytest.mark.gpu
GAP_TOL = 3e-4
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEV = "cuda"


def _gap_rule(acts, racts, logp_trace, t0):
    """-> (fraction of identical tours, first-difference index per rollout); asserts the gap rule at every parting."""
    T = min(acts.shape[1], racts.shape[1])
    neq = acts[:, :T] != racts[:, :T]
    first = torch.where(neq.any(1), neq.float().argmax(1), torch.full((acts.shape[0],), -1))
    same = first < 0
    worst = 0.0
    for r in torch.nonzero(~same).flatten().tolist():
        lp = torch.nan_to_num(logp_trace[int(first[r]) - t0][r], neginf=-1e9)
        top2 = lp.topk(2).values
        g = float(top2[0] - top2[1])
        worst = max(worst, g)
        assert g < GAP_TOL, f"rollout {r} parts from the oracle at step {int(first[r])} where the oracle's gap is {g:.3e}"
    return float(same.float().mean()), first, worst


@pytest.mark.parametrize("weights", ["random", "trained"])
def test_c3_rcvrp_headline_shaped_call_matches_the_live_oracle(weights):
    from rrnco_amd.envs import RCVRPEnv
    from rrnco_amd.models import RRNetPolicy
    from rrnco_amd.models.encoder import ATSPInitEmbedding
    B, N, S = 4, 100, 101
    if weights == "trained":
        z = np.load(H.fixture_path("rcvrp_trained_weights"))
        w = {k: torch.from_numpy(z[k]).float() for k in z.files}
    else:
        w = restate.make_weights(restate.rcvrp_weight_template(), 77)
    pol = H.make_policy(w, env_name="rcvrp", device=DEV)
    env = RCVRPEnv(generator_params=dict(num_loc=N, device=DEV), check_solution=True, device=DEV)
    inst = env.generator(B, generator=torch.Generator(device=DEV).manual_seed(31))
    td = env.reset(inst)
    torch.manual_seed(5)
    td["sample_idx"] = ATSPInitEmbedding.sample_indices(td["distance_matrix"], 25)
    out = pol(td, env, phase="val", decode_type="multistart_greedy", num_starts=S, return_actions=True)
    oinst = {k: inst[k].cpu() for k in ("locs", "depot", "distance_matrix", "demand")}
    st = restate.rcvrp_reset(oinst)
    assert torch.equal(st["distance_matrix"], td["distance_matrix"].cpu())
    tr = {}
    with torch.inference_mode():
        ref = restate.rcvrp_policy(w, st, td["sample_idx"].cpu(), S, "greedy", trace=tr)
    acts, racts = out["actions"].cpu(), ref["actions"]
    frac, first, worst = _gap_rule(acts, racts, tr["logp"], 1)
    same = first < 0
    dll = float((out["log_likelihood"].cpu()[same] - ref["log_likelihood"][same]).abs().max())
    drw = float((out["reward"].cpu()[same] - ref["reward"][same]).abs().max())
    print(f"\n[C3 {weights}] {B} instances x {S} starts: tours identical to the oracle {frac:.5f} ({int((~same).sum())} of {acts.shape[0]} part, "
          f"largest oracle gap at a parting {worst:.2e}); |LL - oracle| {dll:.2e}; |cost - oracle| {drw:.2e}")
    assert frac >= 0.98 and drw < 1e-4 and dll < 2e-3
    if frac == 1.0:
        assert acts.shape == racts.shape


def test_c4_rcvrptw_sampled_routes_teacher_forced_through_the_live_oracle():
    from rrnco_amd.envs import RMTVRPEnv
    from rrnco_amd.models.encoder import ATSPInitEmbedding
    from rrnco_amd.models.transforms import StateAugmentation
    B, N, S = 2, 100, 100
    z = np.load(H.fixture_path("rcvrptw_trained_weights"))
    w = {k: torch.from_numpy(z[k]).float() for k in z.files}
    pol = H.make_policy(w, env_name="rcvrptw", device=DEV)
    env = RMTVRPEnv(generator_params=dict(num_loc=N, device=DEV), device=DEV)
    inst = env.generator(B, generator=torch.Generator(device=DEV).manual_seed(17))
    td = env.reset(StateAugmentation(augment_fn="dihedral8", no_aug_coords=False)(inst))
    torch.manual_seed(6)
    td["sample_idx"] = ATSPInitEmbedding.sample_indices(td["distance_matrix"], 25)
    out = pol(td, env, phase="val", decode_type="multistart_sampling", num_starts=S, seed=11, return_actions=True)
    acts = out["actions"].cpu()
    keys = ("locs", "distance_matrix", "duration_matrix", "demand_linehaul", "time_windows", "service_time")
    oinst = restate.augment_state({k: inst[k].cpu() for k in keys if k in inst.keys()})
    st = restate.rmtvrp_reset(oinst)
    assert torch.equal(st["distance_matrix"], td["distance_matrix"].cpu()) and torch.allclose(st["locs"], td["locs"].cpu())
    tr = {}
    with torch.inference_mode():
        ref = restate.rcvrptw_policy(w, st, td["sample_idx"].cpu(), S, "evaluate", actions=acts[:, 1:], trace=tr)
    # the oracle's loop ends when every teacher-forced rollout is done: exactly the engine's route length
    assert ref["actions"].shape == acts.shape and torch.equal(ref["actions"], acts)
    R, T = acts.shape
    lp_or = ref["logprobs"]                                     # [R, T]: log-probability of the forced action at every step
    assert torch.isfinite(lp_or).all(), "the oracle's mask forbids an action the engine sampled"
    # per-step log-probabilities: the engine keeps one per step too (summed in log_likelihood); compare the sums and the rewards
    dll = float((out["log_likelihood"].cpu() - ref["log_likelihood"]).abs().max())
    drw = float((out["reward"].cpu() - ref["reward"]).abs().max())
    # steps behind a rollout's last customer (waiting at the depot until the batch's longest route ends) carry log-probability 0 on both sides
    print(f"\n[C4] {B} instances x 8 aug x {S} starts sampled: {R} routes of {T} steps teacher-forced through the oracle: "
          f"|LL - oracle| {dll:.2e} (|LL| up to {float(ref['log_likelihood'].abs().max()):.1f}); |cost - oracle| {drw:.2e}")
    assert drw < 1e-4 and dll < 2e-3


def test_atsp_trained_weights_at_the_headline_shape_match_the_live_oracle(monkeypatch):
    import bench
    from rrnco_amd.envs import ATSPEnv, ATSPGenerator
    from rrnco_amd.models.encoder import ATSPInitEmbedding
    B, N, S, A = 8, bench.N_NODES, bench.STARTS, bench.AUG
    z = np.load(H.fixture_path("atsp_trained_weights"))
    w = {k: torch.from_numpy(z[k]).float() for k in z.files}
    pol = H.make_policy(w, device=DEV)
    env = ATSPEnv(generator_params=dict(num_loc=N, device=DEV), check_solution=True, device=DEV)
    td = ATSPGenerator(num_loc=N, device=DEV)(B, generator=torch.Generator(device=DEV).manual_seed(2027))
    inst = {"locs": td["locs"], "distance_matrix": td["distance_matrix"]}
    seen = {}
    orig = ATSPInitEmbedding.sample_indices

    def spy(distance, k):
        o = orig(distance, k)
        seen["sidx"] = o.clone()
        return o
    monkeypatch.setattr(ATSPInitEmbedding, "sample_indices", staticmethod(spy))
    torch.manual_seed(98)
    best, out = bench.hot_path_step(pol, env, inst)
    pol.check_range()
    st = restate.atsp_reset(restate.augment_state({k: v.cpu() for k, v in inst.items()}))
    tr = {}
    with torch.inference_mode():
        ref = restate.atsp_policy(w, st, seen["sidx"].cpu(), S, "greedy", trace=tr)
    acts, racts = out["actions"].cpu(), ref["actions"]
    frac, first, worst = _gap_rule(acts, racts, tr["logp"], 1)
    same = first < 0
    rb = ref["reward"].view(S, A, B).amax(dim=(0, 1))
    db = float((best.cpu() - rb).abs().max())
    dll = float((out["log_likelihood"].cpu()[same] - ref["log_likelihood"][same]).abs().max())
    print(f"\n[ATSP trained] headline shape, {B} instances: tours identical to the oracle {frac:.5f} ({int((~same).sum())} of {acts.shape[0]} part, "
          f"largest oracle gap at a parting {worst:.2e}); |best-of-800 cost - oracle| {db:.2e}; |LL - oracle| {dll:.2e}")
    assert frac >= 0.995 and db < 1e-5 * float(rb.abs().max()) + 1e-5 and dll < 2e-3
