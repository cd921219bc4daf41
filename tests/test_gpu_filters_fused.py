"""`-m gpu`: top-k / top-p filtering (rrnco/models/decoding.py:37-63, 352-358) INSIDE the fused rollout (csrc/rr_rollout_w.inc: FILT
builds of the two-piece greedy / sampling kernels; round 5 — before, a strategy with a filter dropped to the per-step loop).

Two layers:
* the filter functions themselves, through `rr_filter_rows` (the same device functions in the same lane layout): kept set and
  renormalised log-probabilities against the oracle's `process_logits` key by key, with exact ties across the top-k boundary, tanh-saturated
  ties at the TOP of a top-p row (the reference removes the first of them while the cumulative sum stays <= 1 - top_p), a single
  feasible key, rows of every length 1 .. 112;
* the policy: a filtered sampling / greedy rollout in ONE rr_rollout launch (no rr_select call), its sampled actions teacher-forced
  through the oracle (evaluate mode, per-step logits and masks from `trace`), the per-step log-probabilities compared with the
  oracle's FILTERED log-softmax at those actions; the per-step loop (fused=False) draws the same tours from the same seed."""
import pytest
import torch

from oracle import restate
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _filter_rows(x, top_k, top_p):
    from rrnco_amd import _lib as L
    xc = x.cuda().contiguous()
    out = torch.empty_like(xc)
    L.check(L.lib().rr_filter_rows(L.ptr(xc), L.ptr(out), xc.shape[0], xc.shape[1], int(top_k), float(top_p), L.stream()), "rr_filter_rows")
    return out.cpu()


def _stable_top_p_reference(x, top_p):
    """modify_logits_for_top_p_filtering (decoding.py:45-63) with the tie order written out: ascending, stable (equal values in key order)."""
    sl, si = torch.sort(x, descending=False, stable=True)
    rm = sl.softmax(dim=-1).cumsum(dim=-1) <= (1 - top_p)
    return x.masked_fill(rm.scatter(-1, si, rm), float("-inf"))


@pytest.mark.parametrize("N", [1, 5, 20, 64, 100, 112])
@pytest.mark.parametrize("top_k,top_p", [(5, 0.0), (0, 0.9), (12, 0.7), (0, 0.3), (1, 0.0), (3, 0.5)])
def test_filter_functions_match_process_logits(N, top_k, top_p):
    g = torch.Generator().manual_seed(1000 * N + 7 * top_k + int(100 * top_p))
    R = 777
    lg = torch.randn(R, N, generator=g) * 3
    mk = torch.rand(R, N, generator=g) > 0.4
    mk[:, 0] = True
    if N > 17:
        mk[5] = False; mk[5, 17] = True                     # a single feasible key
    x = torch.tanh(lg) * 10.0
    x[~mk] = float("-inf")
    x = x / 1.3
    got = _filter_rows(x, top_k, top_p)
    ref = x.clone()
    if 0 < top_k < N:
        ref = ref.masked_fill(ref < torch.topk(ref, top_k)[0][..., -1, None], float("-inf"))
    if 0.0 < top_p < 1.0:
        ref = _stable_top_p_reference(ref, top_p)
    assert bool(torch.isfinite(got).any(1).all())                                   # the row maximum always survives
    same = (torch.isfinite(got) == torch.isfinite(ref)).all(1)
    # top-p compares a float cumulative sum with 1 - top_p: a row may differ only where that sum lands within rounding of the threshold
    assert float(same.float().mean()) >= (1.0 if not (0.0 < top_p < 1.0) else 0.995), float(same.float().mean())
    assert torch.equal(got[same][torch.isfinite(got[same])], x[same][torch.isfinite(got[same])])      # kept keys keep their value
    # and the reference's own formulation (torch.sort's tie order is unspecified; random rows have no ties)
    ref2 = torch.exp(restate.process_logits(lg, mk, temperature=1.3, tanh_clipping=10.0, top_p=top_p, top_k=top_k)) > 0
    assert float(((torch.isfinite(got) == ref2).all(1)).float().mean()) >= (1.0 if not (0.0 < top_p < 1.0) else 0.995)


def test_filter_functions_with_ties():
    N, R = 100, 256
    g = torch.Generator().manual_seed(5)
    x = torch.randn(R, N, generator=g)
    x[:, 40:47] = x[:, 47:48]                               # seven copies of one value in every row
    x[:64, 40:48] = 9.0                                     # ... at the top of the first rows (a saturated tanh: 8 keys at the maximum)
    x[64:128, 40:48] = -9.0                                 # ... at the bottom of the next ones
    x[200:, 3] = float("-inf")
    # top-k: `logits < k-th value` keeps every copy of the k-th value
    for k in (1, 3, 8, 20):
        got = _filter_rows(x, k, 0.0)
        ref = x.masked_fill(x < torch.topk(x, k)[0][..., -1, None], float("-inf"))
        assert torch.equal(got, ref), k
    # top-p: ties go in key order while the cumulative sum stays <= 1 - top_p; the last of them (the sort's last element) never
    for p in (0.2, 0.5, 0.9, 0.97):
        got = _filter_rows(x, 0, p)
        ref = _stable_top_p_reference(x, p)
        same = (torch.isfinite(got) == torch.isfinite(ref)).all(1)
        assert float(same.float().mean()) >= 0.99, (p, float(same.float().mean()))
        if p <= 0.5:     # rows 0 .. 63: eight keys share the maximum and ~all the mass: the first of them go (1/8 each against 1 - p)
            kept_top = torch.isfinite(got[:64, 40:48]).sum(1)
            assert bool((kept_top == torch.isfinite(ref[:64, 40:48]).sum(1)).all()) and int(kept_top.max()) < 8
            assert bool(torch.isfinite(got[:64, 47]).all())                          # the last in key order stays
    # both: top-k first
    got = _filter_rows(x, 10, 0.6)
    ref = _stable_top_p_reference(x.masked_fill(x < torch.topk(x, 10)[0][..., -1, None], float("-inf")), 0.6)
    assert float((torch.isfinite(got) == torch.isfinite(ref)).all(1).float().mean()) >= 0.99


class _Spy:
    def __init__(self, lib, names):
        self.lib, self.names, self.calls, self.saved = lib, names, {n: 0 for n in names}, {}

    def __enter__(self):
        for n in self.names:
            fn = getattr(self.lib, n)
            self.saved[n] = fn

            def wrap(*a, _fn=fn, _n=n):
                self.calls[_n] += 1
                return _fn(*a)
            setattr(self.lib, n, wrap)
        return self

    def __exit__(self, *exc):
        for n, fn in self.saved.items():
            setattr(self.lib, n, fn)
        return False


def _filtered_logp_of(trace, acts, t0, temperature, top_k, top_p):
    """the oracle's filtered log-softmax at the given actions, per decode step (teacher-forced trace)."""
    cols = []
    for k, (lg, mk) in enumerate(zip(trace["logits"], trace["mask"])):
        lp = restate.process_logits(lg, mk, temperature=temperature, tanh_clipping=10.0, top_p=top_p, top_k=top_k)
        cols.append(lp.gather(1, acts[:, t0 + k:t0 + k + 1])[:, 0])
    return torch.stack(cols, 1)


@pytest.mark.parametrize("name,top_k,top_p,temp", [("atsp_n100_b2_pomo_trained", 5, 0.0, 1.0), ("atsp_n100_b2_pomo_trained", 0, 0.8, 1.5),
                                                  ("atsp_n100_b2_pomo", 8, 0.9, 1.0), ("atsp_n20_b4_pomo", 4, 0.7, 1.2)])
def test_fused_filtered_sampling_matches_the_oracle_atsp(name, top_k, top_p, temp):
    from rrnco_amd import _lib as L
    from tests.test_gpu_atsp import _setup
    fx, w, pol, st, env, td_in = _setup(name)
    S = fx["S"]
    with _Spy(L.lib(), ("rr_rollout", "rr_select")) as spy:
        out = pol(env.reset(td_in), env, phase="test", decode_type="multistart_sampling", num_starts=S, top_k=top_k, top_p=top_p,
                  temperature=temp, seed=11, return_actions=True)
    assert spy.calls["rr_select"] == 0 and spy.calls["rr_rollout"] == 1, spy.calls       # one launch, no per-step selection
    acts = out["actions"].cpu()
    assert restate.atsp_check(acts)
    tr = {}
    with torch.inference_mode():
        restate.atsp_policy(w, restate.atsp_reset(st), fx["sample_idx"], S, "evaluate", actions=acts[:, 1:], trace=tr)
        ref = _filtered_logp_of(tr, acts, 1, temp, top_k, top_p)
    # per-step log-probabilities are not returned by the policy; its log-likelihood is their sum
    fin = torch.isfinite(ref).all(1)
    assert float(fin.float().mean()) >= 0.99, float(fin.float().mean())              # a drawn action outside the oracle's kept set: only at a threshold within rounding
    ll = out["log_likelihood"].cpu()
    d = (ll[fin] - ref[fin].sum(1)).abs()
    print(f"\n[{name} top_k={top_k} top_p={top_p}] rollouts inside the oracle's kept sets {float(fin.float().mean()):.4f}, |LL - oracle filtered LL| max {float(d.max()):.2e}")
    # a kept SET that differs at a threshold within rounding moves the renormalisation by the dropped key's mass: allow a few such rows
    assert float((d < 2e-3).float().mean()) >= 0.98 and float(d.median()) < 5e-4
    # filtering changes the distribution: the unfiltered log-likelihood of the same tours is lower
    with torch.inference_mode():
        unf = _filtered_logp_of(tr, acts, 1, temp, 0, 0.0).sum(1)
    assert float((ll[fin] - unf[fin]).min()) > -1e-3 and float((ll[fin] - unf[fin]).mean()) > 1e-3
    # the per-step loop draws the same tours from the same seed (same uniforms, same inverse-CDF rule, same filters: rr_select)
    with _Spy(L.lib(), ("rr_select",)) as spy2:
        out2 = pol(env.reset(td_in), env, phase="test", decode_type="multistart_sampling", num_starts=S, top_k=top_k, top_p=top_p,
                   temperature=temp, seed=11, return_actions=True, fused=False)
    assert spy2.calls["rr_select"] > 0
    eq = (out2["actions"].cpu() == acts).all(1)
    assert float(eq.float().mean()) >= 0.97, float(eq.float().mean())
    d2 = (out2["log_likelihood"].cpu()[eq] - ll[eq]).abs()          # (a kept set that differs at a threshold within rounding renormalises a step differently)
    assert float((d2 < 2e-3).float().mean()) >= 0.98 and float(d2.median()) < 3e-4, (float(d2.max()), float(d2.median()))


def test_fused_filtered_greedy_keeps_the_tours_and_renormalises():
    from rrnco_amd import _lib as L
    from tests.test_gpu_atsp import _setup
    fx, w, pol, st, env, td_in = _setup("atsp_n100_b2_pomo_trained")
    S = fx["S"]
    base = pol(env.reset(td_in), env, phase="test", decode_type="multistart_greedy", num_starts=S, return_actions=True)
    with _Spy(L.lib(), ("rr_rollout", "rr_select")) as spy:
        out = pol(env.reset(td_in), env, phase="test", decode_type="multistart_greedy", num_starts=S, top_k=3, top_p=0.9, return_actions=True)
    assert spy.calls["rr_select"] == 0 and spy.calls["rr_rollout"] == 1
    assert torch.equal(out["actions"], base["actions"])                 # the row maximum survives both filters
    assert bool((out["log_likelihood"] >= base["log_likelihood"] - 1e-4).all()) and float((out["log_likelihood"] - base["log_likelihood"]).mean()) > 1e-3
    acts = out["actions"].cpu()
    tr = {}
    with torch.inference_mode():
        restate.atsp_policy(w, restate.atsp_reset(st), fx["sample_idx"], S, "evaluate", actions=acts[:, 1:], trace=tr)
        ref = _filtered_logp_of(tr, acts, 1, 1.0, 3, 0.9).sum(1)
    assert float((out["log_likelihood"].cpu() - ref).abs().max()) < 2e-3


@pytest.mark.parametrize("problem", ["rcvrp", "rcvrptw", "rcvrp_n20", "rcvrptw_n20", "rcvrp_n50", "rcvrptw_n50", "rcvrptw_variants"])
def test_fused_filtered_sampling_vrp_matches_the_step_loop(problem):
    """RCVRP / RCVRPTW: the filtered fused rollout against the per-step loop (rr_select's filters, pinned to process_logits in
    test_gpu_atsp.py) from the same seed: same uniforms, same rule — tours equal except where fp32 noise moves a boundary."""
    from rrnco_amd import _lib as L
    # n = 100: instance mode (7 key tiles); n = 50 / 20: the 4- and 2-tile builds of the workgroup-shared form
    names = {"rcvrp": "rcvrp_n100_b2_pomo_trained", "rcvrptw": "rcvrptw_n100_b2_pomo_trained", "rcvrp_n20": "rcvrp_n20_b4_pomo",
             "rcvrptw_n20": "rcvrptw_n20_b4_pomo", "rcvrp_n50": "rcvrp_n50_b3_pomo_trained", "rcvrptw_n50": "rcvrptw_n50_b3_pomo_trained",
             "rcvrptw_variants": "rmtvrp_n20_b8_pomo_variants"}       # (backhauls, open routes, distance limits: the general mask's build)
    if problem.startswith("rcvrptw"):
        from tests.test_gpu_rcvrptw import _setup
    else:
        from tests.test_gpu_rcvrp import _setup
    fx, w, pol, inst, env, td_in = _setup(names[problem])
    S = fx["S"]
    env.check_solution = False
    kw = dict(phase="test", decode_type="multistart_sampling", num_starts=S, top_k=6, top_p=0.85, temperature=1.1, seed=4, return_actions=True)
    with _Spy(L.lib(), ("rr_rollout", "rr_select")) as spy:
        a = pol(env.reset(td_in), env, **kw)
    assert spy.calls["rr_select"] == 0 and spy.calls["rr_rollout"] == 1
    b = pol(env.reset(td_in), env, fused=False, **kw)
    T = min(a["actions"].shape[1], b["actions"].shape[1])
    eq = (a["actions"][:, :T] == b["actions"][:, :T]).all(1).cpu()
    assert float(eq.float().mean()) >= 0.95, float(eq.float().mean())
    d2 = (a["log_likelihood"].cpu()[eq] - b["log_likelihood"].cpu()[eq]).abs()
    assert float((d2 < 3e-3).float().mean()) >= 0.97 and float(d2.median()) < 5e-4, (float(d2.max()), float(d2.median()))
    assert torch.allclose(a["reward"].cpu()[eq], b["reward"].cpu()[eq], atol=1e-5)
    # and the filters did something: the unfiltered sampling run from the same seed scores its tours lower
    c = pol(env.reset(td_in), env, **{**kw, "top_k": 0, "top_p": 0.0})
    assert float(a["log_likelihood"].mean()) > float(c["log_likelihood"].mean())


def test_filters_with_the_fp32_build_take_the_step_loop():
    """The FILT builds exist for the two-piece kernels; under packing.force_fp32() (the range guard's retry path) a filtered strategy
    runs the per-step loop as before."""
    from rrnco_amd import _lib as L, packing
    from tests.test_gpu_atsp import _setup
    fx, w, pol, st, env, td_in = _setup("atsp_n20_b4_pomo")
    with packing.force_fp32(), _Spy(L.lib(), ("rr_select",)) as spy:
        out = pol(env.reset(td_in), env, phase="test", decode_type="multistart_sampling", num_starts=fx["S"], top_k=4, seed=2, return_actions=True)
    assert spy.calls["rr_select"] > 0 and restate.atsp_check(out["actions"].cpu())
