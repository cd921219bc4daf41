"""`-m gpu`: the opt-in precision="16-mixed" rollout (csrc/rr_rollout_w.inc, HALF): one fp16 piece per operand, fp32 accumulation,
fp32 softmax and logits — the arithmetic the reference itself runs on a GPU (torch.autocast in test.py:183, Lightning precision
16-mixed in configs/trainer/default.yaml:8, logits cast back to fp32 in rrnco/models/decoder.py:195-196).

Its yardstick is NOT the fp32 tolerance of the default build but the reference's own mixed-precision deviation:
tests/golden/*_autocast.npz hold the REAL reference run under torch.autocast (fp16 and bf16) on the instances, weights and
neighbour samples of two fp32 fixtures (oracle/gen_golden.py autocast).  Under autocast the reference keeps 0.5 % .. 1.5 % of its
own fp32 tours and moves the log-likelihoods by 1.8 .. 3.0 on average (it also runs the ENCODER in half precision; the variant
here keeps the encoder and the decoder cache fp32-equivalent, so it has to stay well inside that band)."""
import numpy as np
import pytest
import torch

from oracle import restate
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _run(fx, w, precision):
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    from rrnco_amd.models import RRNetPolicy
    pol = H.make_policy(w, device="cuda:0")
    pol.precision = precision
    env = ATSPEnv(generator_params=dict(num_loc=fx["N"]))
    td = TensorDict({"locs": fx["locs"].cuda(), "distance_matrix": fx["distance_matrix"].cuda(),
                     "sample_idx": fx["sample_idx"].cuda()}, batch_size=[fx["B"]])
    out = pol(env.reset(td), env, phase="val", decode_type="multistart_greedy", num_starts=fx["S"], return_actions=True, range_guard="sync")
    torch.cuda.synchronize()
    assert pol.last_range_flags == 0
    return {k: v.cpu() for k, v in out.items() if torch.is_tensor(v)}


def test_policy_rejects_unknown_precision():
    from rrnco_amd.models import RRNetPolicy
    with pytest.raises(ValueError):
        RRNetPolicy(env_name="atsp", precision="8-bit")
    assert RRNetPolicy(env_name="atsp", precision="16-mixed", num_encoder_layers=1, normalization="instance").precision == "16-mixed"


@pytest.mark.parametrize("name", ["atsp_n100_b2_pomo", "atsp_n100_b2_pomo_trained"])
def test_sixteen_mixed_rollout_stays_inside_the_references_own_autocast_deviation(name):
    fx = H.load_fixture(name)
    ac = dict(np.load(H.fixture_path(name + "_autocast")))
    w = H.atsp_weights(fx)
    S, N = fx["S"], fx["N"]
    full = _run(fx, w, "32")
    half = _run(fx, w, "16-mixed")
    ref_a, ref_ll, ref_r = fx["actions"], fx["log_likelihood"], fx["reward"]
    Bp = ref_r.shape[0] // S
    best = lambda r: r.view(S, Bp).max(0).values                                     # noqa: E731
    assert restate.atsp_check(half["actions"])                                        # valid tours whatever the precision
    assert torch.isfinite(half["log_likelihood"]).all() and torch.isfinite(half["reward"]).all()
    # the variant really ran other arithmetic: its log-likelihoods are not the default build's
    assert not torch.equal(half["log_likelihood"], full["log_likelihood"])
    same_ref = (half["actions"] == ref_a).all(1).float().mean().item()
    same_full = (half["actions"] == full["actions"]).all(1).float().mean().item()
    dll = (half["log_likelihood"] - ref_ll).abs()
    gap = (best(ref_r) - best(half["reward"])).mean().item()                          # > 0: the variant's best tour is worse
    r16_same = float(ac["fp16_tours_identical"])
    r16_dll = np.abs(ac["fp16_log_likelihood"] - ref_ll.numpy())
    r16_gap = float((best(ref_r) - best(torch.from_numpy(ac["fp16_reward"]))).mean())
    print(f"\n[{name}] 16-mixed rollout vs the reference's fp32 golden: tours identical {same_ref:.4f} (default build: "
          f"{(full['actions'] == ref_a).all(1).float().mean().item():.4f}; 16-mixed vs default build {same_full:.4f}), |LL - fp32| mean {dll.mean():.3e} "
          f"max {dll.max():.3e}, best-of-{S} cost gap {gap:+.3e}")
    print(f"[{name}] the reference under torch.autocast(fp16) vs its own fp32: tours identical {r16_same:.4f}, |LL - fp32| mean {r16_dll.mean():.3e} "
          f"max {r16_dll.max():.3e}, best-of-{S} cost gap {r16_gap:+.3e}")
    # the tolerance of this variant: inside the reference's own autocast band, with room to spare (its encoder stays fp32-equivalent)
    assert same_ref >= r16_same
    assert float(dll.mean()) <= 0.5 * float(r16_dll.mean()) and float(dll.max()) <= float(r16_dll.max())
    assert gap <= max(r16_gap, 0.0) + 5e-3
    # the reward it reports is the exact cost of the tours it returns (k_tour_cost is fp32 whatever the policy's arithmetic)
    if "normalized_reward" in half:
        D = restate.atsp_reset(H.fixture_state(fx))["distance_matrix"]
        a = half["actions"]
        b = torch.arange(a.shape[0]) % D.shape[0]
        cost = D[b[:, None], a, a.roll(-1, 1)].sum(1)
        assert torch.allclose(-half["normalized_reward"], cost, atol=1e-4)
