"""`-m gpu`: the ablation bias modules (nab_type "heuristic" / "naive") against golden vectors of the
real reference built with the same nab_type (oracle/gen_golden.py ablation)."""
import pytest
import torch

from oracle import restate
from tests import helpers as H

pytestmark = pytest.mark.gpu
ENC_ATOL, LL_RTOL, LL_ATOL, COST_ATOL, GAP_TOL = 5e-4, 2e-5, 4e-3, 1e-4, 1e-3


def _run_plain(name):
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv, RMTVRPEnv
    fx = H.load_fixture(name)
    if fx["kind"] == "atsp":
        w, inst = H.atsp_weights(fx), H.fixture_state(fx)
        env, env_name = ATSPEnv(generator_params=dict(num_loc=fx["N"])), "atsp"
    else:
        w, inst = H.rcvrptw_weights(fx), H.rcvrptw_instance(fx)
        env, env_name = RMTVRPEnv(generator_params=dict(num_loc=fx["N"])), "rcvrptw"
    pol = H.make_policy(w, env_name=env_name)
    td_in = TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[fx["B"]])
    td_in["sample_idx"] = fx["sample_idx"].cuda()
    return fx, w, pol, inst, env, td_in


def _run(name):
    out = _run_plain(name)
    assert out[2].encoder.net.layers[0].row_encoding_block.neural_adaptive_bias.__class__.__name__ in ("_HeuristicNAB", "_NaiveNAB")
    return out


@pytest.mark.parametrize("name", ["atsp_n20_b4_pomo_heuristic", "rcvrptw_n20_b4_pomo_heuristic", "rcvrptw_n20_b4_pomo_naive"])
def test_ablation_nab_encoder_and_tours_match_reference(name):
    fx, w, pol, inst, env, td_in = _run(name)
    row, col = pol.encoder(env.reset(td_in), packed=pol.packed(torch.device("cuda")))
    assert torch.allclose(row.cpu(), fx["row_emb"], atol=ENC_ATOL) and torch.allclose(col.cpu(), fx["col_emb"], atol=ENC_ATOL)
    S = fx["S"]
    for fused in (True, False):
        out = pol(env.reset(td_in), env, phase="val", decode_type="multistart_greedy", num_starts=S, return_actions=True, fused=fused)
        acts = out["actions"].cpu()
        T = min(acts.shape[1], fx["actions"].shape[1])
        frac, first = H.tour_agreement(acts[:, :T], fx["actions"][:, :T])
        assert frac >= 0.97
        same = first < 0
        assert torch.allclose(out["reward"].cpu()[same], fx["reward"][same], atol=COST_ATOL)
        assert torch.allclose(out["log_likelihood"].cpu()[same], fx["log_likelihood"][same], rtol=LL_RTOL, atol=LL_ATOL)
        if fx["kind"] == "atsp":
            assert frac == 1.0


def test_naive_nab_without_duration_matrix_is_rejected_like_the_reference():
    """NaiveNeuralAdaptiveBias concatenates duration_mat (attn_freenet.py:193-195): ATSP / RCVRP cannot use it."""
    t = restate.ablation_template(restate.atsp_weight_template(128, 1, 512, 15), "naive", use_duration=False)
    pol = H.make_policy(restate.make_weights(t, 3))
    with pytest.raises(NotImplementedError, match="duration matrix"):
        pol.packed(torch.device("cuda"))


def test_batchnorm_eval_policy_matches_reference_and_train_mode_uses_batch_statistics():
    """normalization='batch' (the constructor default of RRNetPolicy, 3 layers): running statistics folded into per-feature
    affine maps; golden vectors from the reference in eval mode.  Train mode: the encoder kernels refuse (one instance per
    workgroup), the policy serves it with batch statistics through torch ops (tests/test_gpu_train.py)."""
    fx, w, pol, inst, env, td_in = _run_plain("atsp_n20_b4_pomo_batchnorm")
    assert pol.encoder.normalization == "batch" and len(pol.encoder.net.layers) == 3
    row, col = pol.encoder(env.reset(td_in), packed=pol.packed(torch.device("cuda")))
    assert torch.allclose(row.cpu(), fx["row_emb"], atol=ENC_ATOL) and torch.allclose(col.cpu(), fx["col_emb"], atol=ENC_ATOL)
    out = pol(env.reset(td_in), env, phase="val", decode_type="multistart_greedy", num_starts=fx["S"], return_actions=True)
    assert torch.equal(out["actions"].cpu(), fx["actions"])
    assert torch.allclose(out["reward"].cpu(), fx["reward"], atol=COST_ATOL)
    pol.train()
    with pytest.raises(NotImplementedError, match="running statistics"):
        pol.encoder(env.reset(td_in), packed=pol.packed(torch.device("cuda")))
    out_t = pol(env.reset(td_in), env, phase="train", num_starts=fx["S"])           # batch statistics: a different network function
    assert bool(torch.isfinite(out_t["reward"]).all()) and bool(torch.isfinite(out_t["log_likelihood"]).all())


@pytest.mark.parametrize("name,kind", [("atsp_n20_b4_pomo_rmsnorm", "rms"), ("atsp_n20_b4_pomo_layernorm", "layer")])
def test_rms_and_layer_normalization_policies_match_reference(name, kind):
    """The other two kinds of Normalization (attn_freenet.py:85, 92-93, 106-111): RMSNorm (per node over the features, weight
    only) and the parameter-free 'layer' (one mean / unbiased variance per instance); golden vectors from the reference built
    with those options."""
    fx, w, pol, inst, env, td_in = _run_plain(name)
    assert pol.encoder.normalization == kind
    row, col = pol.encoder(env.reset(td_in), packed=pol.packed(torch.device("cuda")))
    assert torch.allclose(row.cpu(), fx["row_emb"], atol=ENC_ATOL) and torch.allclose(col.cpu(), fx["col_emb"], atol=ENC_ATOL)
    out = pol(env.reset(td_in), env, phase="val", decode_type="multistart_greedy", num_starts=fx["S"], return_actions=True)
    frac, first = H.tour_agreement(out["actions"].cpu(), fx["actions"])
    assert frac >= 0.98
    same = first < 0
    assert torch.allclose(out["reward"].cpu()[same], fx["reward"][same], atol=COST_ATOL)

