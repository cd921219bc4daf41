"""`-m gpu` parity tests for RCVRPTW (RMTVRPEnv, vrptw preset; BASELINE configs[3] shape at N=100) against the golden
vectors of the real reference: NAB with the duration matrix (MFMA kernel), time-window masks, duration inductive bias."""
import pytest
import torch

from oracle import restate
from tests import helpers as H

pytestmark = pytest.mark.gpu
ENC_ATOL, LL_RTOL, LL_ATOL, COST_ATOL, GAP_TOL = 5e-4, 2e-5, 4e-3, 1e-4, 1e-3
# *_trained: a policy trained for 1 600 REINFORCE steps on this engine (tools/train_fixture_weights.py --problem rcvrptw; chosen-action
# probability 0.93 .. 0.96), run through the REAL reference by oracle/gen_golden.py rcvrptw_trained
FIXTURES = ["rcvrptw_n20_b4_pomo", "rcvrptw_n20_b4_greedy", "rcvrptw_n100_b2_pomo", "rcvrptw_n100_b2_pomo_trained", "rcvrptw_n50_b3_pomo_trained"]


def _setup(name):
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import RMTVRPEnv
    fx = H.load_fixture(name)
    w = H.rcvrptw_weights(fx)
    pol = H.make_policy(w, env_name="rcvrptw")
    inst = H.rcvrptw_instance(fx)
    env = RMTVRPEnv(generator_params=dict(num_loc=fx["N"]))
    td_in = TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[fx["B"]])
    td_in["sample_idx"] = fx["sample_idx"].cuda()
    return fx, w, pol, inst, env, td_in


def test_rmtvrp_env_step_and_time_window_mask_match_oracle():
    from rrnco_amd.ops import batchify
    fx, w, pol, inst, env, td_in = _setup("rcvrptw_n20_b4_pomo")
    S = fx["S"]
    td0 = env.reset(td_in)
    o0 = restate.rmtvrp_reset(inst)
    assert torch.equal(td0["distance_matrix"].cpu(), fx["norm_distance"]) and torch.equal(td0["action_mask"].cpu(), o0["action_mask"])
    assert env.get_num_starts(td0) == fx["N"] and torch.equal(env.select_start_nodes(td0, S).cpu(), fx["actions"][:, 0])
    td = batchify(td0, S)
    otd = restate.batchify_state({k: v for k, v in o0.items() if k not in ("locs", "min_distance", "max_distance")}, S)
    for t in range(fx["actions"].shape[1]):
        a = fx["actions"][:, t]
        td.set("action", a.cuda()); td = env.step(td)["next"]
        otd["action"] = a; otd = restate.rmtvrp_step(otd)
        assert torch.equal(td["action_mask"].cpu(), otd["action_mask"]), t          # TW / capacity feasibility, bit-exact
        assert torch.equal(td["visited"].cpu(), otd["visited"]) and torch.equal(td["done"].cpu(), otd["done"])
        assert torch.equal(td["current_time"].cpu(), otd["current_time"])           # same fp32 op order: exact
        assert torch.equal(td["used_capacity_linehaul"].cpu(), otd["used_capacity_linehaul"])
        assert torch.equal(td["current_route_length"].cpu(), otd["current_route_length"])
    assert td["done"].all()
    real, nd = env.get_reward(td, fx["actions"].cuda())
    assert torch.allclose(real.cpu(), fx["reward"], atol=COST_ATOL)


VARIANTS = "rmtvrp_n20_b8_pomo_variants"     # backhauls (classes 1 and 2), open routes, distance limits, mixed per instance


def test_rmtvrp_variant_env_step_and_masks_bit_exact_vs_oracle():
    """rmtvrp/env.py:155-215, 343-428 beyond the vrptw preset (rr_rmtvrp_step with MtvrpExtra) along the reference's tours."""
    from rrnco_amd.ops import batchify
    fx, w, pol, inst, env, td_in = _setup(VARIANTS)
    assert bool(fx["open_route"].any()) and bool(torch.isfinite(fx["distance_limit"]).any()) and bool((fx["demand_backhaul"] > 0).any())
    assert set(fx["backhaul_class"].flatten().tolist()) == {1, 2}
    S = fx["S"]
    td0 = env.reset(td_in)
    assert td0.meta["mtvrp_variant"] is True
    o0 = restate.rmtvrp_reset(inst)
    assert torch.equal(td0["action_mask"].cpu(), o0["action_mask"])
    td = batchify(td0, S)
    otd = restate.batchify_state({k: v for k, v in o0.items() if k not in ("locs", "min_distance", "max_distance")}, S)
    for t in range(fx["actions"].shape[1]):
        a = fx["actions"][:, t]
        td.set("action", a.cuda()); td = env.step(td)["next"]
        otd["action"] = a; otd = restate.rmtvrp_step(otd)
        assert torch.equal(td["action_mask"].cpu(), otd["action_mask"]), t
        assert torch.equal(td["visited"].cpu(), otd["visited"]) and torch.equal(td["done"].cpu(), otd["done"])
        for k in ("current_time", "used_capacity_linehaul", "used_capacity_backhaul", "current_route_length"):
            assert torch.equal(td[k].cpu(), otd[k]), (k, t)
    assert td["done"].all()
    real, nd = env.get_reward(td, fx["actions"].cuda())                    # open routes: arcs into the depot are free
    assert torch.allclose(real.cpu(), fx["reward"], atol=COST_ATOL) and torch.allclose(nd.cpu(), fx["normalized_reward"], atol=COST_ATOL)


@pytest.mark.parametrize("fused", [True, False])
def test_rmtvrp_variant_policy_routes_match_reference(fused):
    """Greedy multistart decode on the mixed-variant batch (backhaul classes 1 / 2, open routes, distance limits), on the
    fused rollout (general mask inside the kernel) and on the step-wise loop (rr_rmtvrp_step with MtvrpExtra): MTVRP context
    with open-route flag / remaining distance / backhaul load; both reproduce the reference's tours."""
    fx, w, pol, inst, env, td_in = _setup(VARIANTS)
    S = fx["S"]
    out = pol(env.reset(td_in), env, phase="val", decode_type="multistart_greedy", num_starts=S, return_actions=True, fused=fused)
    acts = out["actions"].cpu()
    T = min(acts.shape[1], fx["actions"].shape[1])
    frac, first = H.tour_agreement(acts[:, :T], fx["actions"][:, :T])
    if frac < 1.0:
        tr = {}
        with torch.inference_mode():
            restate.rcvrptw_policy(w, restate.rmtvrp_reset(inst), fx["sample_idx"], S, "greedy", trace=tr)
        lp = torch.nan_to_num(torch.stack(tr["logp"], 1), neginf=-1e9).topk(2, -1).values
        gap = lp[..., 0] - lp[..., 1]
        for r in torch.nonzero(first >= 0).flatten().tolist():
            t = int(first[r]) - 1
            assert t >= gap.shape[1] or gap[r, t] < GAP_TOL
    name = VARIANTS
    print(f"\n[{name} fused={fused}] tours identical to the reference {frac:.4f}; |LL - ref| max {float((out['log_likelihood'].cpu()[first < 0] - fx['log_likelihood'][first < 0]).abs().max()):.2e}")
    assert frac >= 0.97
    same = first < 0
    assert torch.allclose(out["reward"].cpu()[same], fx["reward"][same], atol=COST_ATOL)
    assert torch.allclose(out["log_likelihood"].cpu()[same], fx["log_likelihood"][same], rtol=LL_RTOL, atol=LL_ATOL)


@pytest.mark.parametrize("n_nodes", [21, 51, 101])
def test_rmtvrp_variant_fused_rollout_agrees_with_stepwise_loop(n_nodes):
    """Sampling decode of random multi-task instances: the in-kernel general env.step and the step kernel see the same
    masks, so both paths draw the same tours (same counter-based noise) up to near-tie decisions."""
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import RMTVRPEnv
    fx, w, pol, _, _, _ = _setup(VARIANTS)
    n = n_nodes - 1
    env = RMTVRPEnv(generator_params=dict(num_loc=n))
    inst = restate.rmtvrp_variant_synthetic(6, n, 500 + n)
    sidx = restate.sample_neighbor_indices(restate.rmtvrp_reset(inst)["distance_matrix"], fx["sample_size"],
                                           generator=torch.Generator().manual_seed(n))
    outs = []
    for fused in (True, False):
        td = TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[6])
        td["sample_idx"] = sidx.cuda()
        td = env.reset(td)
        assert td.meta.get("mtvrp_variant", False)
        outs.append(pol(td, env, phase="val", decode_type="multistart_sampling", num_starts=n, return_actions=True, fused=fused, seed=5))
    a, b = outs[0]["actions"], outs[1]["actions"]
    T = min(a.shape[1], b.shape[1])
    same = (a[:, :T] == b[:, :T]).all(1)
    assert same.float().mean() >= 0.95
    assert torch.allclose(outs[0]["reward"][same], outs[1]["reward"][same], atol=COST_ATOL)
    cust = a.sort(1).values[:, -n:]
    assert (cust == torch.arange(1, n + 1, device=a.device)).all()


@pytest.mark.parametrize("name", FIXTURES)
def test_rcvrptw_encoder_with_duration_nab_matches_reference(name):
    fx, w, pol, inst, env, td_in = _setup(name)
    row, col = pol.encoder(env.reset(td_in), packed=pol.packed(torch.device("cuda")))
    assert torch.allclose(row.cpu(), fx["row_emb"], atol=ENC_ATOL) and torch.allclose(col.cpu(), fx["col_emb"], atol=ENC_ATOL)


@pytest.mark.parametrize("n_nodes", [51, 72, 101, 110])
def test_duration_nab_three_kernel_generations_agree(n_nodes, monkeypatch):
    """rr_nab_dur: the LDS-resident piecewise-linear kernel (default for N*N >= 2048), the L2-gather one and the MFMA
    contraction evaluate the same folded formula; they may differ by fp32 association only."""
    from rrnco_amd import _lib as L
    fx, w, pol, inst, env, td_in = _setup("rcvrptw_n20_b4_pomo")
    packed = pol.packed(torch.device("cuda"))
    g = torch.Generator().manual_seed(n_nodes)
    Bp, N = 3, n_nodes
    D = torch.rand(Bp, N, N, generator=g).cuda(); T = torch.rand(Bp, N, N, generator=g).cuda()
    D[0, 0, :5] = 0.0; T[1, 2, :5] = 1.0
    locs = torch.rand(Bp, N, 2, generator=g).cuda()
    outs = []
    for variant in ("1", "2", "0"):
        monkeypatch.setenv("RR_NABDUR_VARIANT", variant)
        bias = torch.full((Bp, 2, N * N), float("nan"), device="cuda")
        for nr, nc in packed["nabdur"]:
            L.check(L.lib().rr_nab_dur(nr, nc, L.ptr(D), L.ptr(T), L.ptr(locs), L.ptr(bias), Bp, N, L.stream()), "rr_nab_dur")
            outs.append(bias.clone())
    nl = len(packed["nabdur"])
    for l in range(nl):
        a, b, c = outs[l], outs[nl + l], outs[2 * nl + l]
        assert torch.isfinite(a).all()
        assert (a - b).abs().max() < 1e-5 * (1 + b.abs().max())
        assert (a - c).abs().max() < 2e-5 * (1 + c.abs().max())


@pytest.mark.parametrize("n_nodes", [51, 101])
def test_duration_nab_shared_over_the_eight_augmentations_agrees(n_nodes):
    """rr_nab_dur_aug (x8 dihedral augmentation: distance / duration part of an edge evaluated once for the 8 copies) against rr_nab_dur
    on the replicated batch: the same folded formula, the three families summed in another order; and the encoder takes it exactly
    when StateAugmentation's note says the batch is such a batch."""
    from rrnco_amd import _lib as L
    from rrnco_amd import TensorDict
    from rrnco_amd.models.transforms import StateAugmentation
    fx, w, pol, inst, env, td_in = _setup("rcvrptw_n20_b4_pomo")
    packed = pol.packed(torch.device("cuda"))
    g = torch.Generator().manual_seed(7 * n_nodes)
    B, N = 3, n_nodes
    base = TensorDict({"distance_matrix": torch.rand(B, N, N, generator=g).cuda(), "duration_matrix": torch.rand(B, N, N, generator=g).cuda(),
                       "locs": torch.rand(B, N, 2, generator=g).cuda()}, batch_size=[B])
    td = StateAugmentation(augment_fn="dihedral8", no_aug_coords=False)(base)
    assert td.meta.get("num_augment") == 8
    D, T, locs = td["distance_matrix"].contiguous(), td["duration_matrix"].contiguous(), td["locs"].contiguous()
    Bp = 8 * B
    for nr, nc in packed["nabdur"]:
        a = torch.full((Bp, 2, N * N), float("nan"), device="cuda"); b = torch.full_like(a, float("nan"))
        L.check(L.lib().rr_nab_dur(nr, nc, L.ptr(D), L.ptr(T), L.ptr(locs), L.ptr(a), Bp, N, L.stream()), "rr_nab_dur")
        L.check(L.lib().rr_nab_dur_aug(nr, nc, L.ptr(D), L.ptr(T), L.ptr(locs), L.ptr(b), Bp, N, 8, L.stream()), "rr_nab_dur_aug")
        assert torch.isfinite(b).all()
        assert (a - b).abs().max() < 1e-5 * (1 + a.abs().max()), float((a - b).abs().max())
    # anything that is not the x8 form is refused (callers fall back to rr_nab_dur)
    nr, nc = packed["nabdur"][0]
    assert L.lib().rr_nab_dur_aug(nr, nc, L.ptr(D), L.ptr(T), L.ptr(locs), L.ptr(a), Bp, N, 4, L.stream()) != 0


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", FIXTURES)
def test_rcvrptw_policy_greedy_routes_match_reference(name, fused):
    fx, w, pol, inst, env, td_in = _setup(name)
    S = fx["S"]
    out = pol(env.reset(td_in), env, phase="val", decode_type="multistart_greedy" if S > 1 else "greedy",
              num_starts=S if S > 1 else None, return_actions=True, fused=fused)
    acts = out["actions"].cpu()
    n = fx["N"]
    cust = acts.sort(1).values[:, -n:]
    assert (cust == torch.arange(1, n + 1)).all()                     # every customer exactly once
    T = min(acts.shape[1], fx["actions"].shape[1])
    frac, first = H.tour_agreement(acts[:, :T], fx["actions"][:, :T])
    if frac < 1.0:
        tr = {}
        with torch.inference_mode():
            restate.rcvrptw_policy(w, restate.rmtvrp_reset(inst), fx["sample_idx"], S, "greedy", trace=tr)
        lp = torch.nan_to_num(torch.stack(tr["logp"], 1), neginf=-1e9).topk(2, -1).values
        gap = lp[..., 0] - lp[..., 1]
        for r in torch.nonzero(first >= 0).flatten().tolist():
            t = int(first[r]) - (1 if S > 1 else 0)
            assert t >= gap.shape[1] or gap[r, t] < GAP_TOL
    print(f"\n[{name} fused={fused}] tours identical to the reference {frac:.4f}; |LL - ref| max {float((out['log_likelihood'].cpu()[first < 0] - fx['log_likelihood'][first < 0]).abs().max()):.2e}")
    assert frac >= 0.97
    same = first < 0
    assert torch.allclose(out["reward"].cpu()[same], fx["reward"][same], atol=COST_ATOL)
    assert torch.allclose(out["log_likelihood"].cpu()[same], fx["log_likelihood"][same], rtol=LL_RTOL, atol=LL_ATOL)


@pytest.mark.parametrize("preset", ["cvrp", "ovrp", "vrpb", "vrpl", "ovrpbl", "vrpbltw", "ovrpbltw", "all"])
def test_rmtvrp_generator_presets_env_dynamics_match_oracle(preset):
    """rmtvrp/generator.py:37-58, 352-432: every feature combination the presets produce (infinite time windows and limits,
    backhaul demands folded into linehaul, open routes) through reset / step / get_action_mask on the kernels against the
    oracle, along random feasible walks.  (The RRNet policy itself needs finite time windows: RVRPTWInitEmbedding feeds the raw
    windows to a Linear, in the reference as here — the non-TW presets are an environment feature.)"""
    from rrnco_amd.envs import RMTVRPEnv
    dev = torch.device("cuda")
    env = RMTVRPEnv(generator_params=dict(num_loc=20, variant_preset=preset, device=dev, sample_backhaul_class=True), device=dev)
    td_in = env.generator(8, generator=torch.Generator(device=dev).manual_seed(11))
    inst = {k: td_in[k].cpu() for k in td_in.keys()}
    td = env.reset(td_in)
    otd = restate.rmtvrp_reset(inst)
    assert torch.equal(td["action_mask"].cpu(), otd["action_mask"])
    otd = {k: v for k, v in otd.items() if k not in ("locs", "min_distance", "max_distance")}
    g = torch.Generator().manual_seed(3)
    for t in range(80):
        m = otd["action_mask"]
        a = torch.multinomial(m.float() + 1e-9 * (m.sum(-1, keepdim=True) == 0), 1, generator=g)[:, 0]
        td.set("action", a.cuda()); td = env.step(td)["next"]
        otd["action"] = a; otd = restate.rmtvrp_step(otd)
        assert torch.equal(td["action_mask"].cpu(), otd["action_mask"]), (preset, t)
        assert torch.equal(td["visited"].cpu(), otd["visited"]) and torch.equal(td["done"].cpu(), otd["done"])
        for k in ("current_time", "used_capacity_linehaul", "used_capacity_backhaul", "current_route_length"):
            assert torch.equal(td[k].cpu(), otd[k]), (preset, k, t)
        if bool(otd["done"].all()):
            break
    assert bool(otd["done"].all())


def test_fused_rollout_sampling_draws_from_the_policy_distribution_rcvrptw():
    """... and the RCVRPTW instantiation (time-window mask): tests/helpers.sampling_law_check."""
    fx, w, pol, inst, env, td_in = _setup("rcvrptw_n20_b4_pomo")
    seen, worst = H.sampling_law_check(pol, env, inst, fx["sample_idx"], fx["S"])
    assert seen >= fx["S"] and worst < 5.0, (seen, worst)


def test_two_streams_with_two_host_threads_give_the_single_stream_results():
    """rrnco_amd.parallel.run_on_streams (throughput mode of BASELINE configs[3]: one host thread, one HIP stream and one policy object per
    worker): every pass returns exactly what the same call returns alone."""
    from rrnco_amd import TensorDict
    from rrnco_amd.parallel import run_on_streams
    fx, w, pol, inst, env, td_in = _setup("rcvrptw_n20_b4_pomo")
    pol2 = H.make_policy(w, env_name="rcvrptw")
    S = fx["S"]

    def call(p, seed):
        td = TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[fx["B"]])
        td["sample_idx"] = fx["sample_idx"].cuda()
        return p(env.reset(td), env, phase="val", decode_type="multistart_sampling", num_starts=S, seed=seed, return_actions=True)
    ref = [call(pol, 5), call(pol, 6)]
    got = [[], []]
    workers = [lambda: got[0].append(call(pol, 5)), lambda: got[1].append(call(pol2, 6))]
    sec = run_on_streams(workers, 3)
    assert sec > 0 and len(got[0]) == 3 and len(got[1]) == 3
    for i in range(2):
        for o in got[i]:
            assert torch.equal(o["actions"], ref[i]["actions"]) and torch.equal(o["reward"], ref[i]["reward"])
            assert torch.equal(o["log_likelihood"], ref[i]["log_likelihood"])
