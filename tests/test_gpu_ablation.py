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



@pytest.mark.parametrize("name", ["atsp_n20_b8_pomo_random_idx", "atsp_n20_b4_pomo_coords_only", "atsp_n20_b4_pomo_dist_only",
                                  "atsp_n100_b2_pomo_dist_only"])
def test_init_embedding_branches_match_reference(name):
    """The non-default branches of ATSPInitEmbedding (env_embeddings/atsp.py) against golden vectors of the real
    reference built with the same switches (oracle/gen_golden.py initvariants): sample_type="random" (:38-54: index rows shared by all
    nodes of an augmentation block — the published kernel given those indices), use_dist=False (:92: row = col = init_embed(locs)) and
    use_coords=False (:94-104: unsorted gathers into row_embed / col_embed) on rr_init_embed_plain."""
    fx, w, pol, inst, env, td_in = _run_plain(name)
    ie = pol.encoder.init_embedding
    assert (ie.init_embed is None) == ("dist_only" in name) and hasattr(ie, "gating_network_row") == ("random_idx" in name)
    assert sorted(pol.state_dict().keys()) == sorted(w.keys())                 # the same parameters as the reference builds (atsp.py:29-35)
    packed = pol.packed(torch.device("cuda"))
    assert packed["init_mode"] == (2 if "dist_only" in name else 1 if "coords_only" in name else 0)
    row, col = pol.encoder(env.reset(td_in), packed=packed)
    dev_emb = max(float((row.cpu() - fx["row_emb"]).abs().max()), float((col.cpu() - fx["col_emb"]).abs().max()))
    S = fx["S"]
    out = pol(env.reset(td_in), env, phase="val", decode_type="multistart_greedy", num_starts=S, return_actions=True)
    frac, first = H.tour_agreement(out["actions"].cpu(), fx["actions"])
    print(f"\n[{name}] max |embedding - reference| {dev_emb:.2e}; tours identical on {frac * 100:.2f} % of the rollouts")
    assert dev_emb < 2e-4
    assert frac >= 0.99
    same = first < 0
    assert torch.allclose(out["reward"].cpu()[same], fx["reward"][same], atol=COST_ATOL)
    assert torch.allclose(out["log_likelihood"].cpu()[same], fx["log_likelihood"][same], rtol=LL_RTOL, atol=LL_ATOL)


def test_shared_random_indices_law_and_default_draw():
    """sample_type="random" without an explicit td["sample_idx"]: one index row for the whole batch in training, one per block of
    B / 8 instances otherwise (atsp.py:38-54) — same torch.randint calls in the same order as the reference, so the same generator state
    gives the same indices; the VRP embeddings share the law (rcvrp.py:153-169).  The forward runs on it."""
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    from rrnco_amd.models import RRNetPolicy
    from rrnco_amd.models.encoder import shared_random_indices
    dev = torch.device("cuda")
    torch.manual_seed(5)
    a = shared_random_indices("train", 6, 20, 15, dev)
    assert a.shape == (6, 20, 15) and bool((a == a[0, 0]).all()) and int(a.min()) >= 0 and int(a.max()) < 20
    torch.manual_seed(5)
    assert torch.equal(a[0, 0], torch.randint(0, 20, (1, 15), device=dev)[0])
    b = shared_random_indices("val", 16, 20, 15, dev)
    assert b.shape == (16, 20, 15)
    blocks = b.view(8, 2, 20, 15)
    assert bool((blocks == blocks[:, :1, :1]).all()) and len({tuple(blocks[i, 0, 0].tolist()) for i in range(8)}) > 1
    with pytest.raises(RuntimeError):
        shared_random_indices("val", 12, 20, 15, dev)            # the reference's reshape fails too unless 8 | B
    torch.manual_seed(1)
    pol = RRNetPolicy(env_name="atsp", embed_dim=128, num_heads=8, num_encoder_layers=2, normalization="instance", use_graph_context=False,
                      init_embedding_kwargs=dict(sample_type="random", sample_size=15)).to(dev).eval()
    env = ATSPEnv(generator_params=dict(num_loc=20, device=dev), check_solution=True, device=dev)
    td = env.generator(16, generator=torch.Generator(device=dev).manual_seed(2))
    out = pol(env.reset(TensorDict(dict(td.items()), batch_size=[16])), env, phase="val", decode_type="multistart_greedy", num_starts=20,
              return_actions=True)
    assert bool(torch.isfinite(out["reward"]).all()) and out["actions"].shape == (320, 20)
    for env_name in ("rcvrp", "rcvrptw"):
        from rrnco_amd.models.vrp_embeddings import make_vrp_init_embedding
        e = make_vrp_init_embedding(env_name, 128, sample_type="random", sample_size=10)
        i = e.indices_for(torch.rand(8, 21, 21, device=dev), "val")
        assert i.shape == (8, 21, 10) and bool((i == i[:, :1]).all())
        with pytest.raises(NotImplementedError, match="fail inside the reference"):
            make_vrp_init_embedding(env_name, 128, use_dist=False)
