"""Run-to-run determinism of the fused rollout.  The greedy decode is a pure function of its inputs: repeated launches must
return bit-identical tours and log-likelihoods.  (A write-after-read hazard between a VALU write and a just-issued
v_mfma_f32_16x16x32_f16 once made the RCVRPTW rollout's P.V differ between runs in the low bits while every parity test
passed most of the time: csrc/rr_common.h, RR_MFMA_SRC_FENCE.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu

REPEATS = 25


def _repeat(pol, env, td_in, S, **kw):
    ref = None
    for i in range(REPEATS):
        out = pol(env.reset(td_in), env, phase="val", decode_type="multistart_greedy", num_starts=S, return_actions=True, fused=True, **kw)
        a, ll, rw = out["actions"], out["log_likelihood"], out["reward"]
        if ref is None:
            ref = (a.clone(), ll.clone(), rw.clone())
            continue
        assert torch.equal(a, ref[0]), f"tours differ in repeat {i}"
        assert torch.equal(ll, ref[1]), f"log-likelihoods differ in repeat {i}: max {float((ll - ref[1]).abs().max()):.3e}"
        assert torch.equal(rw, ref[2])


@pytest.mark.parametrize("which", ["rcvrptw_n20_b4_pomo", "variants"])
def test_rcvrptw_rollout_is_deterministic(which):
    from tests import test_gpu_rcvrptw as T
    fx, w, pol, inst, env, td_in = T._setup(T.VARIANTS if which == "variants" else which)
    _repeat(pol, env, td_in, fx["S"])


def test_rcvrp_rollout_is_deterministic():
    from tests import test_gpu_rcvrp as T
    fx, w, pol, inst, env, td_in = T._setup("rcvrp_n20_b4_pomo")
    _repeat(pol, env, td_in, fx["S"])


@pytest.mark.parametrize("name", ["atsp_n20_b4_pomo", "atsp_n100_b2_pomo"])
def test_atsp_rollout_is_deterministic(name):
    from tests import helpers as H
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    fx = H.load_fixture(name)
    pol = H.make_policy(H.atsp_weights(fx), "atsp")
    env = ATSPEnv(check_solution=False, device=torch.device("cuda"))
    st = H.fixture_state(fx)
    td_in = TensorDict({k: v.cuda() for k, v in st.items()}, batch_size=[st["locs"].shape[0]])
    td_in["sample_idx"] = fx["sample_idx"].cuda()
    _repeat(pol, env, td_in, fx["S"])


def test_full_size_atsp_rollout_is_deterministic():
    """512 instances x 8 augmentations x 100 starts (BASELINE.json configs[1]): every workgroup shape of the headline launch."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from rrnco_amd.envs import ATSPEnv, ATSPGenerator
    dev = torch.device("cuda")
    pol, _ = bench.make_policy(dev)
    env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
    td = ATSPGenerator(num_loc=100, device=dev)(256, generator=torch.Generator(device=dev).manual_seed(3))
    inst = {"locs": td["locs"], "distance_matrix": td["distance_matrix"]}
    outs = []
    for _ in range(3):
        torch.manual_seed(11)
        best, out = bench.hot_path_step(pol, env, inst)
        outs.append((out["actions"].clone(), out["log_likelihood"].clone()))
    for a, ll in outs[1:]:
        assert torch.equal(a, outs[0][0]) and torch.equal(ll, outs[0][1])
