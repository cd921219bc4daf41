"""`-m gpu`: shapes the golden fixtures do not cover — other tile-count templates (N <= 32 / <= 64 / <= 103), ragged
start counts (S not a multiple of 16, S > 112, S = 1), the N = 103 limit — each against the oracle run on the fly."""
import pytest
import torch

from oracle import restate
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _run(N, B, S, ss, seed, layers=2):
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    w = restate.make_weights(restate.atsp_weight_template(128, layers, 512, ss), seed)
    pol = H.make_policy(w)
    inst = restate.atsp_synthetic(B, N, seed)
    st0 = restate.atsp_reset(inst)
    sidx = restate.sample_neighbor_indices(st0["distance_matrix"], ss, generator=torch.Generator().manual_seed(seed))
    tr = {}
    with torch.inference_mode():
        ref = restate.atsp_policy(w, st0, sidx, S, "greedy", trace=tr)
    env = ATSPEnv(generator_params=dict(num_loc=N))
    td = TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[B])
    td["sample_idx"] = sidx.cuda()
    outs = [pol(env.reset(td), env, phase="val", decode_type="multistart_greedy" if S > 1 else "greedy",
                num_starts=S if S > 1 else None, fused=f) for f in (True, False)]
    for out in outs:
        acts = out["actions"].cpu()
        assert acts.shape == ref["actions"].shape and restate.atsp_check(acts)
        frac, first = H.tour_agreement(acts, ref["actions"])
        if frac < 1.0:
            lp = torch.nan_to_num(torch.stack(tr["logp"], 1), neginf=-1e9).topk(2, -1).values
            gap = lp[..., 0] - lp[..., 1]
            for r in torch.nonzero(first >= 0).flatten().tolist():
                assert gap[r, int(first[r]) - (1 if S > 1 else 0)] < 1e-3
        assert frac >= 0.97
        same = first < 0
        assert torch.allclose(out["reward"].cpu()[same], ref["reward"][same], atol=5e-5)
        assert torch.allclose(out["log_likelihood"].cpu()[same], ref["log_likelihood"][same], rtol=2e-5, atol=3e-3)
    # fused and step-wise agree except where fp32 noise decides a near-tie (each was checked against the oracle's gaps above;
    # the two paths evaluate tanh(log u) differently: (u^2-1)/(u^2+1) in the rollout, tanh(log) in rr_select)
    R = outs[0]["actions"].shape[0]
    assert float((outs[0]["actions"] == outs[1]["actions"]).all(1).float().mean()) >= min(0.99, 1.0 - 2.0 / R)   # two near-ties allowed on tiny batches


@pytest.mark.parametrize("N,B,S,ss", [
    (50, 3, 50, 25),      # 4-tile template, S not a multiple of 16
    (64, 2, 7, 25),       # 4-tile template upper edge, one partial rollout tile
    (33, 3, 33, 20),      # just above the 2-tile template
    (12, 5, 12, 8),       # small instance, 2-tile template
    (20, 2, 1, 15),       # num_starts = 1 -> plain greedy path of get_decoding_strategy
    (20, 9, 20, 15),      # tail packing: 4 left-over rollouts x 4 instances per tile, last tail tile holds one instance
    (24, 20, 17, 15),     # tail packing at its widest: 1 left-over rollout x 16 instances per tile (16 passes), then 4
])
def test_atsp_other_shapes_match_oracle(N, B, S, ss):
    _run(N, B, S, ss, seed=100 + N + S)


def test_tail_packed_rollout_still_matches_the_oracle():
    """Tail packing (several instances' left-over rollouts in one tile) is off by default (csrc/rr_decode.hip); RR_TAIL_PACK=1 is
    read once per process, so the packed shapes run in a child process with it set."""
    import os, subprocess, sys
    env = dict(os.environ, RR_TAIL_PACK="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "test_atsp_other_shapes_match_oracle and (20-9-20-15 or 24-20-17-15)"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_eight_tile_form_of_the_seven_tile_shapes_still_matches_the_oracle():
    """Launches with 7 rollout tiles per instance run in instance mode (csrc/rr_rollout_w.inc: one instance per workgroup, K + L in
    LDS, loader wave); RR_ROLLOUT_INST=0 (read once per process) sends the same shapes through the 8-tile form, which other
    shapes still use: golden tours, log-likelihoods and determinism in a child process with it set."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RR_ROLLOUT_INST="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(root, "tests", "test_gpu_atsp.py"),
                        os.path.join(root, "tests", "test_gpu_rcvrptw.py"), os.path.join(root, "tests", "test_gpu_determinism.py"),
                        "-k", "greedy_tours_match_reference or split_bf16_mlp or determin or repeat"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout and "no tests ran" not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_atsp_max_nodes_103_and_more_starts_than_a_workgroup_tile_set():
    _run(103, 2, 103, 25, seed=7)


def test_streamed_rows_equal_the_register_forms_bit_for_bit(monkeypatch):
    """csrc/rr_bign.hip: rows of more than 208 keys are streamed (k_aft_mix_stream, k_dec_fwd_big<0>, k_select_big<16>); RR_BIGN_STREAM=1
    sends every N through those forms.  Same matrix instructions, same order of every sum: embeddings, tours and log-likelihoods of
    a 150- and a 200-node instance are identical to the register forms' bit for bit."""
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    w = restate.make_weights(restate.atsp_weight_template(128, 2, 512, 25), 3)
    pol = H.make_policy(w)
    for N, S in ((150, 20), (200, 9)):
        inst = restate.atsp_synthetic(2, N, 13)
        env = ATSPEnv(generator_params=dict(num_loc=N))
        sidx = restate.sample_neighbor_indices(restate.atsp_reset(inst)["distance_matrix"], 25, generator=torch.Generator().manual_seed(5)).cuda()
        res = {}
        for sw in ("0", "1"):
            monkeypatch.setenv("RR_BIGN_STREAM", sw)
            td = TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[2]); td["sample_idx"] = sidx
            td = env.reset(td)
            row, col = pol.encoder(td, packed=pol.packed(torch.device("cuda")))
            g = pol(td, env, phase="val", decode_type="multistart_greedy", num_starts=S)
            td2 = TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[2]); td2["sample_idx"] = sidx
            smp = pol(env.reset(td2), env, phase="val", decode_type="multistart_sampling", num_starts=S, seed=4)
            res[sw] = (row, col, g["actions"], g["log_likelihood"], smp["actions"], smp["log_likelihood"])
        monkeypatch.delenv("RR_BIGN_STREAM")
        for a, b in zip(res["0"], res["1"]):
            assert torch.equal(a, b)


@pytest.mark.parametrize("N,S", [(128, 16), (200, 24), (300, 12)])
def test_atsp_more_than_103_nodes_matches_the_live_oracle(N, S):
    """N > 103: the row-parallel kernels of csrc/rr_bign.hip under the same policy API (step-by-step decode loop), against the
    oracle run live: embeddings, greedy POMO tours (every divergence at an oracle decision gap < GAP_TOL), costs, log-likelihoods."""
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    ss, B = 25, 2
    w = restate.make_weights(restate.atsp_weight_template(128, 2, 512, ss), 3)
    pol = H.make_policy(w)
    inst = restate.atsp_synthetic(B, N, 11)
    st0 = restate.atsp_reset(inst)
    sidx = restate.sample_neighbor_indices(st0["distance_matrix"], ss, generator=torch.Generator().manual_seed(5))
    trace = {}
    with torch.inference_mode():
        ref = restate.atsp_policy(w, st0, sidx, S, "greedy", trace=trace)
    env = ATSPEnv(generator_params=dict(num_loc=N))
    td = TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[B])
    td["sample_idx"] = sidx.cuda()
    td = env.reset(td)
    row, col = pol.encoder(td, packed=pol.packed(torch.device("cuda")))
    assert torch.allclose(row.cpu(), trace["row_emb"], atol=2e-4) and torch.allclose(col.cpu(), trace["col_emb"], atol=2e-4)
    out = pol(td, env, phase="val", decode_type="multistart_greedy", num_starts=S)
    acts = out["actions"].cpu()
    assert restate.atsp_check(acts)
    frac, first = H.tour_agreement(acts, ref["actions"])
    if frac < 1.0:
        lp = torch.stack(trace["logp"], 1)
        top2 = torch.nan_to_num(lp, neginf=-1e9).topk(2, -1).values
        gap = top2[..., 0] - top2[..., 1]
        for r in torch.nonzero(first >= 0).flatten().tolist():
            assert gap[r, int(first[r]) - 1] < 1e-3
    assert frac >= 0.9
    same = first < 0
    assert torch.allclose(out["reward"].cpu()[same], ref["reward"][same], atol=1e-4)
    assert torch.allclose(out["log_likelihood"].cpu()[same], ref["log_likelihood"][same], rtol=2e-5, atol=2e-3)
    # sampling and evaluate on the same path
    smp = pol(env.reset(TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[B])), env, phase="val",
              decode_type="multistart_sampling", num_starts=S, seed=9)
    assert restate.atsp_check(smp["actions"].cpu())
    with torch.inference_mode():
        ev = restate.atsp_policy(w, st0, sidx, S, "evaluate", actions=acts[:, 1:])
    td2 = TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[B]); td2["sample_idx"] = sidx.cuda()
    mine = pol(env.reset(td2), env, phase="val", actions=acts[:, 1:].cuda(), num_starts=S)
    assert torch.allclose(mine["log_likelihood"].cpu(), ev["log_likelihood"], rtol=2e-5, atol=2e-3)


def test_rcvrp_more_than_103_nodes_roundtrip_properties():
    """RCVRP with 150 customers on the row-parallel path: every customer once, capacity never exceeded, cost = route length."""
    from rrnco_amd.envs import RCVRPEnv
    w = restate.make_weights(restate.rcvrp_weight_template(128, 2, 512, 20), 5)
    pol = H.make_policy(w, env_name="rcvrp")
    env = RCVRPEnv(generator_params=dict(num_loc=150), check_solution=True)
    td_in = env.generator(3, generator=torch.Generator(device="cuda").manual_seed(2))
    out = pol(env.reset(td_in), env, phase="val", decode_type="multistart_greedy", num_starts=12)
    a = out["actions"]
    served = torch.zeros(a.shape[0], 151, dtype=torch.bool, device="cuda").scatter_(1, a, True)
    assert bool(served[:, 1:].all()) and bool(torch.isfinite(out["reward"]).all())


def test_rcvrp_500_nodes_roundtrip_properties():
    """RCVRP with 500 customers (streamed rows): every customer once, capacity kept (the env's own check), finite costs, and the greedy
    tours' cost recomputed from the real distance matrix."""
    from rrnco_amd.envs import RCVRPEnv
    w = restate.make_weights(restate.rcvrp_weight_template(128, 2, 512, 20), 5)
    pol = H.make_policy(w, env_name="rcvrp")
    env = RCVRPEnv(generator_params=dict(num_loc=500), check_solution=True)
    td_in = env.generator(2, generator=torch.Generator(device="cuda").manual_seed(2))
    out = pol(env.reset(td_in), env, phase="val", decode_type="multistart_greedy", num_starts=6)
    a = out["actions"]
    served = torch.zeros(a.shape[0], 501, dtype=torch.bool, device="cuda").scatter_(1, a, True)
    assert bool(served[:, 1:].all()) and bool(torch.isfinite(out["reward"]).all())
    D = td_in["distance_matrix"]
    R = a.shape[0]
    b = torch.arange(R, device="cuda") % 2
    path = torch.cat([torch.zeros(R, 1, dtype=torch.int64, device="cuda"), a, torch.zeros(R, 1, dtype=torch.int64, device="cuda")], 1)
    cost = D[b[:, None], path[:, :-1], path[:, 1:]].sum(1)
    assert torch.allclose(-out["reward"], cost, rtol=2e-5, atol=1e-3)


def test_atsp_1000_nodes_roundtrip_properties():
    """The reference's generators go to 1 000 nodes (rcvrp/generator.py:21-37): one ATSP instance of 1 000 nodes end to end (streamed
    rows, 999 decode steps): every start yields a permutation, the reported cost is the tour's length on the real matrix, greedy and
    evaluate agree on the log-likelihood."""
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    N, S = 1000, 5
    w = restate.make_weights(restate.atsp_weight_template(128, 1, 512, 25), 3)
    pol = H.make_policy(w)
    inst = restate.atsp_synthetic(1, N, 17)
    env = ATSPEnv(generator_params=dict(num_loc=N))
    td = env.reset(TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[1]))
    sidx = restate.sample_neighbor_indices(td["distance_matrix"].cpu(), 25, generator=torch.Generator().manual_seed(5)).cuda()
    td["sample_idx"] = sidx
    out = pol(td, env, phase="val", decode_type="multistart_greedy", num_starts=S)
    acts = out["actions"].cpu()
    assert acts.shape == (S, N) and restate.atsp_check(acts)
    D = inst["distance_matrix"][0]
    cost = D[acts, acts.roll(-1, 1)].sum(1)
    assert torch.allclose(-out["reward"].cpu(), cost, rtol=2e-5, atol=1e-3)
    td2 = env.reset(TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[1])); td2["sample_idx"] = sidx
    ev = pol(td2, env, phase="val", actions=out["actions"][:, 1:], num_starts=S)
    assert torch.allclose(ev["log_likelihood"], out["log_likelihood"], rtol=1e-5, atol=2e-3)


def test_rcvrptw_300_nodes_roundtrip_properties():
    """RCVRPTW with 300 customers on the streamed forms (duration NAB, MTVRP context): routes the env's own checker accepts."""
    from rrnco_amd.envs import RMTVRPEnv
    w = restate.make_weights(restate.rcvrptw_weight_template(128, 2, 512, 20), 7)
    pol = H.make_policy(w, env_name="rcvrptw")
    env = RMTVRPEnv(generator_params=dict(num_loc=300))
    td_in = env.generator(2, generator=torch.Generator(device="cuda").manual_seed(4))
    out = pol(env.reset(td_in), env, phase="val", decode_type="multistart_greedy", num_starts=5)
    a = out["actions"]
    served = torch.zeros(a.shape[0], 301, dtype=torch.bool, device="cuda").scatter_(1, a, True)
    assert bool(served[:, 1:].all()) and bool(torch.isfinite(out["reward"]).all())
    # time windows and capacity re-walked on the host along every route (the reference has no checker: rmtvrp/env.py:457-461)
    ac, st = a.cpu(), env.reset(td_in)          # (the reset state: the quantities the mask works on, rmtvrp/env.py:289-340)
    for r in range(ac.shape[0]):
        b = r % 2
        T, tw = st["duration_matrix"][b].cpu(), st["time_windows"][b].cpu()
        sv, dem, cap = st["service_time"][b].cpu(), st["demand_linehaul"][b].cpu(), float(st["vehicle_capacity"].reshape(-1)[b])
        t, load, cur = 0.0, 0.0, 0
        for n in ac[r].tolist():
            if n == 0:
                t, load, cur = 0.0, 0.0, 0
                continue
            arr = t + float(T[cur, n])
            assert arr < float(tw[n, 1]) + 1e-5, (r, n)
            t = max(arr, float(tw[n, 0])) + float(sv[n])
            load += float(dem[n])
            assert load <= cap + 1e-5, (r, n)
            cur = n


def test_rcvrptw_more_than_103_nodes_matches_the_live_oracle():
    """RCVRPTW with 120 customers (N = 121) on the row-parallel path (duration NAB by rr_nab_dur, MTVRP context, duration inductive
    bias, step-wise loop on rr_rmtvrp_step / rr_select) against the oracle run live: embeddings, greedy tours (divergences only at
    oracle gaps < 1e-3), rewards."""
    from rrnco_amd.envs import RMTVRPEnv
    N, S, B, ss = 120, 16, 2, 20
    w = restate.make_weights(restate.rcvrptw_weight_template(128, 2, 512, ss), 7)
    pol = H.make_policy(w, env_name="rcvrptw")
    env = RMTVRPEnv(generator_params=dict(num_loc=N))
    td_in = env.generator(B, generator=torch.Generator(device="cuda").manual_seed(4))
    inst = {k: td_in[k].cpu() for k in td_in.keys()}
    st0 = restate.rmtvrp_reset(inst)
    sidx = restate.sample_neighbor_indices(st0["distance_matrix"], ss, generator=torch.Generator().manual_seed(5))
    trace = {}
    with torch.inference_mode():
        ref = restate.rcvrptw_policy(w, st0, sidx, S, "greedy", trace=trace)
    td_in["sample_idx"] = sidx.cuda()
    td = env.reset(td_in)
    row, col = pol.encoder(td, packed=pol.packed(torch.device("cuda")))
    assert torch.allclose(row.cpu(), trace["row_emb"], atol=3e-4) and torch.allclose(col.cpu(), trace["col_emb"], atol=3e-4)
    out = pol(td, env, phase="val", decode_type="multistart_greedy", num_starts=S)
    acts = out["actions"].cpu()
    T = min(acts.shape[1], ref["actions"].shape[1])
    frac, first = H.tour_agreement(acts[:, :T], ref["actions"][:, :T])
    if frac < 1.0:
        lp = torch.nan_to_num(torch.stack(trace["logp"], 1), neginf=-1e9).topk(2, -1).values
        gap = lp[..., 0] - lp[..., 1]
        for r in torch.nonzero(first >= 0).flatten().tolist():
            t = int(first[r]) - 1
            assert t >= gap.shape[1] or gap[r, t] < 1e-3
    assert frac >= 0.8
    same = first < 0
    assert torch.allclose(out["reward"].cpu()[same], ref["reward"][same], atol=2e-4)


def test_more_than_1024_nodes_is_rejected_loudly():
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    w = restate.make_weights(restate.atsp_weight_template(128, 1, 512, 25), 1)
    pol = H.make_policy(w)
    inst = restate.atsp_synthetic(1, 1025, 1)
    env = ATSPEnv(generator_params=dict(num_loc=1025))
    td = TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[1])
    with pytest.raises(NotImplementedError, match="nodes"):
        pol(env.reset(td), env, phase="val", decode_type="greedy")


def test_rcvrp_generator_instances_roundtrip_properties_n50():
    """Size-independent properties on generator-made RCVRP instances (no fixture): every customer once, capacity never
    exceeded, cost equals the recomputed route length, fused == step-wise."""
    from rrnco_amd.envs import RCVRPEnv
    w = restate.make_weights(restate.rcvrp_weight_template(128, 2, 512, 20), 5)
    pol = H.make_policy(w, env_name="rcvrp")
    env = RCVRPEnv(generator_params=dict(num_loc=50), check_solution=True)
    td_in = env.generator(6, generator=torch.Generator(device="cuda").manual_seed(2))
    from rrnco_amd.models.encoder import ATSPInitEmbedding
    td0 = env.reset(td_in)
    td_in["sample_idx"] = ATSPInitEmbedding.sample_indices(td0["distance_matrix"], 20)
    a = pol(env.reset(td_in), env, phase="val", decode_type="multistart_greedy", num_starts=51, fused=True)
    b = pol(env.reset(td_in), env, phase="val", decode_type="multistart_greedy", num_starts=51, fused=False)
    assert torch.equal(a["actions"], b["actions"]) and torch.allclose(a["reward"], b["reward"])
    acts = a["actions"].cpu()
    R = acts.shape[0]
    chk = {"demand": td_in["demand"].cpu()[torch.arange(R) % 6], "vehicle_capacity": torch.ones(R, 1)}
    assert restate.rcvrp_check(chk, acts)
    D = td0["distance_matrix"].cpu()
    for r in (0, 17, R - 1):
        path = torch.cat([torch.zeros(1, dtype=torch.long), acts[r], torch.zeros(1, dtype=torch.long)])
        cost = D[r % 6, path[:-1], path[1:]].double().sum()
        assert abs(float(a["normalized_reward"][r]) + float(cost)) < 1e-4


def _run_vrp(problem, N, B, S, ss, seed, layers=2):
    """RCVRP / RCVRPTW at shapes the fixtures do not cover, against the oracle run on the fly (tolerances of the
    fixture tests: every divergence must sit at an oracle decision gap < 1e-3)."""
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import RCVRPEnv, RMTVRPEnv
    if problem == "rcvrp":
        w = restate.make_weights(restate.rcvrp_weight_template(128, layers, 512, ss), seed)
        inst = restate.rcvrp_synthetic(B, N, seed, capacity=restate.vrptw_capacity(N))
        st0 = restate.rcvrp_reset(inst)
        env, run = RCVRPEnv(generator_params=dict(num_loc=N)), restate.rcvrp_policy
    else:
        w = restate.make_weights(restate.rcvrptw_weight_template(128, layers, 512, ss), seed)
        inst = restate.rcvrptw_synthetic(B, N, seed)
        st0 = restate.rmtvrp_reset(inst)
        env, run = RMTVRPEnv(generator_params=dict(num_loc=N)), restate.rcvrptw_policy
    pol = H.make_policy(w, env_name=problem)
    sidx = restate.sample_neighbor_indices(st0["distance_matrix"], ss, generator=torch.Generator().manual_seed(seed))
    tr = {}
    with torch.inference_mode():
        ref = run(w, st0, sidx, S, "greedy", trace=tr)
    td = TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[B])
    td["sample_idx"] = sidx.cuda()
    outs = [pol(env.reset(td), env, phase="val", decode_type="multistart_greedy", num_starts=S, fused=f) for f in (True, False)]
    # fused ~ step-wise (see _run): the two paths round differently, so a near-tie of the oracle may fall either way in each — both
    # are held to the oracle below (a fuzzed case had five rollouts of one instance share a 1.3e-5 tie: 96 % between the paths)
    T2 = min(outs[0]["actions"].shape[1], outs[1]["actions"].shape[1])
    assert float((outs[0]["actions"][:, :T2] == outs[1]["actions"][:, :T2]).all(1).float().mean()) >= 0.9
    lp = torch.nan_to_num(torch.stack(tr["logp"], 1), neginf=-1e9).topk(2, -1).values
    gap = lp[..., 0] - lp[..., 1]
    for o in reversed(outs):                                                       # (ends on the fused call: `first` is used below)
        acts = o["actions"].cpu()
        assert bool((acts.sort(1).values[:, -N:] == torch.arange(1, N + 1)).all())     # every customer exactly once
        T = min(acts.shape[1], ref["actions"].shape[1])
        frac, first = H.tour_agreement(acts[:, :T], ref["actions"][:, :T])
        for r in torch.nonzero(first >= 0).flatten().tolist():
            t = int(first[r]) - 1
            assert t >= gap.shape[1] or gap[r, t] < 1e-3
        assert frac >= 0.95
    same = first < 0
    assert torch.allclose(outs[0]["reward"].cpu()[same], ref["reward"][same], atol=1e-4)
    assert torch.allclose(outs[0]["log_likelihood"].cpu()[same], ref["log_likelihood"][same], rtol=2e-5, atol=4e-3)


@pytest.mark.parametrize("problem,N,B,S,ss", [
    ("rcvrp", 37, 3, 38, 20),       # 38 nodes: 4-tile template, ragged S, odd batch
    ("rcvrp", 63, 2, 64, 25),       # 64 nodes: upper edge of the 4-tile template, S a multiple of 16 (no tail tile)
    ("rcvrp", 102, 2, 50, 25),      # 103 nodes: the kernels' maximum
    ("rcvrptw", 37, 3, 37, 20),
    ("rcvrptw", 70, 2, 70, 25),     # 71 nodes: 7-tile template below 100
    ("rcvrptw", 102, 1, 33, 25),    # 103 nodes, one instance (tail packing needs two: disabled)
])
def test_vrp_other_shapes_match_oracle(problem, N, B, S, ss):
    _run_vrp(problem, N, B, S, ss, seed=300 + N)


def test_mtvrp_variants_n50_match_oracle_live():
    """Backhauls / open routes / distance limits at N=50 (no golden tours exist there: the reference and its restatement
    already differ on 11 % of the rollouts at decision gaps < 1e-3) against the oracle run on the fly."""
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import RMTVRPEnv
    N, B, S, ss, seed = 50, 3, 50, 25, 77
    w = restate.make_weights(restate.rcvrptw_weight_template(128, 2, 512, ss), seed)
    inst = restate.rmtvrp_variant_synthetic(B, N, seed)
    st0 = restate.rmtvrp_reset(inst)
    pol = H.make_policy(w, env_name="rcvrptw")
    sidx = restate.sample_neighbor_indices(st0["distance_matrix"], ss, generator=torch.Generator().manual_seed(seed))
    tr = {}
    with torch.inference_mode():
        ref = restate.rcvrptw_policy(w, st0, sidx, S, "greedy", trace=tr)
    env = RMTVRPEnv(generator_params=dict(num_loc=N))
    td = TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[B])
    td["sample_idx"] = sidx.cuda()
    out = pol(env.reset(td), env, phase="val", decode_type="multistart_greedy", num_starts=S)
    acts = out["actions"].cpu()
    assert bool((acts.sort(1).values[:, -N:] == torch.arange(1, N + 1)).all())
    T = min(acts.shape[1], ref["actions"].shape[1])
    frac, first = H.tour_agreement(acts[:, :T], ref["actions"][:, :T])
    lp = torch.nan_to_num(torch.stack(tr["logp"], 1), neginf=-1e9).topk(2, -1).values
    gap = lp[..., 0] - lp[..., 1]
    for r in torch.nonzero(first >= 0).flatten().tolist():
        t = int(first[r]) - 1
        assert t >= gap.shape[1] or gap[r, t] < 1e-3
    assert frac >= 0.85
    same = first < 0
    assert torch.allclose(out["reward"].cpu()[same], ref["reward"][same], atol=1e-4)
