"""`-m gpu` parity tests of the MatNet baseline encoder (csrc/rr_matnet.hip through the C-ABI) against the
golden outputs of the reference's own MatNetEncoder (tests/golden/matnet_*.npz, oracle/gen_golden.py matnet) and against
the oracle run live on other shapes.  fp32 MFMA accumulation order differs from torch's GEMMs: tolerance 5e-4 on the
instance-normalised embeddings (the same bar as the RRNet encoder), 2e-4 after one layer."""
import pytest
import torch

from oracle import restate
from tests import helpers as H

pytestmark = pytest.mark.gpu
ENC_ATOL, L1_ATOL = 5e-4, 2e-4


def _encoder(fx_or_kw, w):
    from rrnco_amd.baselines import MatNetEncoder
    enc = MatNetEncoder(embed_dim=fx_or_kw["embed_dim"], num_heads=fx_or_kw["heads"], num_layers=fx_or_kw["layers"],
                        env_name=fx_or_kw["env_name"])
    enc.load_state_dict(w, strict=True)
    return enc.cuda().eval()


@pytest.mark.parametrize("name", ["matnet_atsp_n20_b4", "matnet_rcvrp_n20_b4", "matnet_atsp_n100_b2", "matnet_rcvrp_n100_b2"])
def test_matnet_encoder_matches_reference_golden(name):
    fx = H.load_fixture(name)
    w = H.matnet_weights(fx)
    td = {"distance_matrix": fx["distance_matrix"].cuda()}
    if fx["env_name"] == "rcvrp":
        td["demand"] = fx["demand"].cuda()
    (row, col), init = _encoder(fx, w)(td, rand_idx=fx["rand_idx"].cuda())
    assert init is None
    assert torch.allclose(row.cpu(), fx["row_emb"], atol=ENC_ATOL) and torch.allclose(col.cpu(), fx["col_emb"], atol=ENC_ATOL)
    one = _encoder(dict(fx, layers=1), {k: v for k, v in w.items() if not k.startswith("layers.") or k.startswith("layers.0.")})
    (r1, c1), _ = one(td, rand_idx=fx["rand_idx"].cuda())
    assert torch.allclose(r1.cpu(), fx["row_l1"], atol=L1_ATOL) and torch.allclose(c1.cpu(), fx["col_l1"], atol=L1_ATOL)


@pytest.mark.parametrize("env_name,n_nodes,batch", [("atsp", 33, 5), ("atsp", 64, 3), ("rcvrp", 50, 4), ("atsp", 112, 2), ("rcvrp", 7, 3)])
def test_matnet_encoder_matches_oracle_on_other_shapes(env_name, n_nodes, batch):
    cfg = dict(embed_dim=256, heads=16, layers=2, env_name=env_name)
    w = restate.make_weights(restate.matnet_weight_template(256, 16, 2, 512, env_name), 900 + n_nodes)
    g = torch.Generator().manual_seed(n_nodes)
    D = torch.rand(batch, n_nodes, n_nodes, generator=g)
    td = {"distance_matrix": D}
    if env_name == "rcvrp":
        td["demand"] = torch.rand(batch, n_nodes - 1, generator=g) * 0.3
    rand_idx = torch.rand(batch, n_nodes, generator=g).argsort(dim=1)
    with torch.inference_mode():
        row, col = restate.matnet_encoder(w, td, rand_idx, 2, 16, env_name, 256)
    (r, c), _ = _encoder(cfg, w)({k: v.cuda() for k, v in td.items()}, rand_idx=rand_idx.cuda())
    assert torch.allclose(r.cpu(), row, atol=ENC_ATOL) and torch.allclose(c.cpu(), col, atol=ENC_ATOL)


def test_matnet_encoder_rejects_what_it_does_not_implement():
    from rrnco_amd.baselines import MatNetEncoder
    with pytest.raises(NotImplementedError):
        MatNetEncoder(embed_dim=128, num_heads=16)
    with pytest.raises(NotImplementedError):
        MatNetEncoder(normalization="batch")
    enc = MatNetEncoder(num_layers=1, env_name="atsp").cuda()
    with pytest.raises(Exception):
        enc({"distance_matrix": torch.rand(2, 10, 10)})          # CPU tensors: there is no CPU path


@pytest.mark.parametrize("name", ["matnet_policy_atsp_n20_b4", "matnet_policy_atsp_n50_b2", "matnet_policy_atsp_n100_b2"])
def test_matnet_policy_tours_match_reference(name):
    """MatNetPolicy end to end on ATSP (HIP encoder, rr_matnet_linear / rr_matnet_dec_step decoder, rr_select_matnet with the
    baseline's clamped process_logits, rr_atsp_step) against the reference's MatNetPolicy.forward.  The clamp makes every action
    within 1e-4 of the best one an exact tie (lowest index wins), so divergences are only accepted where the oracle's own top-2
    gap BEFORE the clamp is within 2e-4 of that threshold or below 1e-3 in total."""
    from rrnco_amd import TensorDict
    from rrnco_amd.baselines import MatNetPolicy
    from rrnco_amd.envs import ATSPEnv
    fx = H.load_fixture(name)
    w = restate.make_weights(restate.matnet_policy_template(fx["embed_dim"], fx["heads"], fx["layers"], 512, "atsp"), fx["seed"])
    pol = MatNetPolicy(env_name="atsp", num_encoder_layers=fx["layers"])
    pol.load_state_dict(w, strict=True)
    pol = pol.cuda().eval()
    env = ATSPEnv(generator_params=dict(num_loc=fx["N"]), check_solution=True)
    td = TensorDict({"locs": fx["locs"].cuda(), "distance_matrix": fx["distance_matrix"].cuda()}, batch_size=[fx["B"]])
    out = pol(env.reset(td), env, phase="val", decode_type="multistart_greedy", num_starts=fx["S"], return_actions=True,
              rand_idx=fx["rand_idx"].cuda())
    acts = out["actions"].cpu()
    assert restate.atsp_check(acts)
    frac, first = H.tour_agreement(acts, fx["actions"])
    assert frac >= 0.95
    same = first < 0
    assert torch.allclose(out["reward"].cpu()[same], fx["reward"][same], atol=2e-5)
    assert torch.allclose(out["log_likelihood"].cpu()[same], fx["log_likelihood"][same], rtol=2e-5, atol=3e-3)
    if frac < 1.0:
        st0 = restate.atsp_reset({"locs": fx["locs"], "distance_matrix": fx["distance_matrix"]})
        tr = {}
        with torch.inference_mode():
            restate.matnet_policy_atsp(w, st0, fx["rand_idx"], fx["S"], fx["layers"], fx["heads"], fx["embed_dim"], trace=tr)
        for r in torch.nonzero(first >= 0).flatten().tolist():
            lg = torch.tanh(tr["logits"][int(first[r]) - 1][r]) * 10.0
            top = lg.masked_fill(tr["logp"][int(first[r]) - 1][r] < -40, float("-inf")).topk(2).values
            assert float(top[0] - top[1]) < 1.2e-3


@pytest.mark.parametrize("name", ["matnet_policy_rcvrp_n20_b4", "matnet_policy_rcvrp_n50_b2", "matnet_policy_rcvrp_n100_b2"])
def test_matnet_policy_rcvrp_routes_match_reference(name):
    """MatNetPolicy on RCVRP, the environment configs/experiment/matnet.yaml trains on (RVRPInitEmbedding without coordinates,
    rl4co VRPContext, RCVRPEnv masks), against the reference's MatNetPolicy.forward."""
    from rrnco_amd import TensorDict
    from rrnco_amd.baselines import MatNetPolicy
    from rrnco_amd.envs import RCVRPEnv
    fx = H.load_fixture(name)
    w = restate.make_weights(restate.matnet_policy_template(fx["embed_dim"], fx["heads"], fx["layers"], 512, "rcvrp"), fx["seed"])
    pol = MatNetPolicy(env_name="rcvrp", num_encoder_layers=fx["layers"])
    pol.load_state_dict(w, strict=True)
    pol = pol.cuda().eval()
    env = RCVRPEnv(generator_params=dict(num_loc=fx["N"]), check_solution=True)
    td = TensorDict({k: fx[k].cuda() for k in ("locs", "depot", "distance_matrix", "demand")}, batch_size=[fx["B"]])
    out = pol(env.reset(td), env, phase="val", decode_type="multistart_greedy", num_starts=fx["S"], return_actions=True,
              rand_idx=fx["rand_idx"].cuda())
    acts = out["actions"].cpu()
    n = fx["N"]
    assert (acts.sort(1).values[:, -n:] == torch.arange(1, n + 1)).all()
    T = min(acts.shape[1], fx["actions"].shape[1])
    frac, first = H.tour_agreement(acts[:, :T], fx["actions"][:, :T])
    assert frac >= 0.95
    same = first < 0
    assert torch.allclose(out["reward"].cpu()[same], fx["reward"][same], atol=1e-4)
    assert torch.allclose(out["log_likelihood"].cpu()[same], fx["log_likelihood"][same], rtol=2e-5, atol=4e-3)

