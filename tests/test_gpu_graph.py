"""`-m gpu`: the ATSP hot path (reset, encoder, fused rollout, reward) captures into one hipGraph and replays with identical tours:
every launcher is asynchronous on the caller's stream, none allocates through the runtime, and the deferred range guard reads
nothing back while a capture is in progress."""
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_hot_path_captures_into_a_hip_graph_and_replays_identically():
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    from rrnco_amd.models.transforms import StateAugmentation
    fx = H.load_fixture("atsp_n20_b4_pomo")
    pol = H.make_policy(H.atsp_weights(fx)).eval()
    env = ATSPEnv(generator_params=dict(num_loc=fx["N"]), check_solution=False)
    st = {k: v.cuda() for k, v in H.fixture_state(fx).items()}
    B = st["locs"].shape[0]
    sidx = fx["sample_idx"].cuda().repeat(8, 1, 1).contiguous()

    def step():
        td = StateAugmentation(augment_fn="dihedral8", no_aug_coords=False)(TensorDict(dict(st), batch_size=[B]))
        td = env.reset(td)
        td.set("sample_idx", sidx)
        return pol(td, env, phase="val", decode_type="multistart_greedy", num_starts=fx["S"], return_actions=True)

    with torch.no_grad():
        ref = step()
        torch.cuda.synchronize()
        g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        with torch.cuda.stream(s):
            step()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                out = step()
        torch.cuda.synchronize()
        out["actions"].zero_(); out["reward"].zero_()
        g.replay()
        torch.cuda.synchronize()
    assert torch.equal(out["actions"], ref["actions"]) and torch.equal(out["reward"], ref["reward"])
    pol.check_range()          # the word the captured call left pending is clean


def test_replayed_graph_keeps_a_range_flag_raised_by_a_middle_batch():
    """the guard word of a captured call is created and zeroed INSIDE the capture, so every replay resets it and
    check_range() used to see the LAST replay only — an out-of-range batch in the middle of a dataset showed up as `Average cost: nan`
    and nothing else.  Now every replay ORs its word into a persistent device word (RRNetPolicy.prepare_graph_capture); evaluate.py's
    _GraphedPolicy replays clean / poisoned / clean batches: the poisoned one's rewards are NaN-marked inside the graph, the clean
    ones are untouched, and check_range() raises although the LAST replay was clean."""
    import evaluate
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    fx = H.load_fixture("atsp_n20_b4_pomo")
    pol = H.make_policy(H.atsp_weights(fx)).eval()
    env = ATSPEnv(generator_params=dict(num_loc=fx["N"]), check_solution=False)
    st = {k: v.cuda() for k, v in H.fixture_state(fx).items()}
    B = st["locs"].shape[0]
    graphed = evaluate._GraphedPolicy(pol, env, fx["S"])

    def batch(poison):
        d = {k: v.clone() for k, v in st.items()}
        if poison:
            d["locs"][0, 0, 0] = float("nan")            # the encoder's embeddings of instance 0 become non-finite: the K / V / L images raise bit 0
        return env.reset(TensorDict(d, batch_size=[B]))

    def run(poison):
        torch.manual_seed(0)                              # the same neighbour sample for every call (drawn per forward, outside the graph)
        return graphed(batch(poison))["reward"].clone()

    with torch.no_grad():
        run(False)                                       # (captures; its own neighbour sample comes second in the seeded stream)
        clean0 = run(False)
        pol.check_range()                                # a clean replay: nothing raised, the persistent word stays pending
        bad = run(True)
        clean1 = run(False)
        torch.cuda.synchronize()
    assert len(graphed.graphs) == 1                       # one capture, three replays
    assert bool(torch.isfinite(clean0).all()) and torch.equal(clean0, clean1)
    assert bool(torch.isnan(bad).all())                   # device-side poison of the flagged replay
    with pytest.raises(FloatingPointError):
        pol.check_range()
    assert pol.last_range_flags & 1
    pol.check_range()                                     # read and cleared


@pytest.mark.parametrize("problem", ["rcvrp", "rcvrptw"])
def test_vrp_policy_call_replays_from_a_hip_graph_with_the_reference_output_shape(problem):
    """bench.py's `hipgraph_replay_exact_shape` variant (evaluate.py --hipgraph per batch): the VRP policy call in its padded form
    (policy.lazy_trim: no host read inside) captured once, replayed, then ONE host read of the step count and the trim as a view — tours,
    rewards and the trimmed length equal the eager call's (the reference's output shape)."""
    if problem == "rcvrp":
        from tests import test_gpu_rcvrp as T
        fx, w, pol, inst, env, td_in = T._setup("rcvrp_n20_b4_pomo")
    else:
        from tests import test_gpu_rcvrptw as T
        fx, w, pol, inst, env, td_in = T._setup("rcvrptw_n20_b4_pomo")
    S = fx["S"]
    kw = dict(phase="val", decode_type="multistart_greedy", num_starts=S, return_actions=True)
    with torch.no_grad():
        ref = pol(env.reset(td_in), env, **kw)                      # eager: actions trimmed to the longest route
        static = env.reset(td_in)                                   # (RMTVRPEnv.reset reads a flag back: outside the capture)
        pol.lazy_trim = True
        checked, env.check_solution = env.check_solution, False     # (the validity check of get_reward reads back: the eager call above ran it)
        try:
            g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                pol(static.clone(), env, **kw)
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=s):
                    out = pol(static.clone(), env, **kw)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            out["actions"].zero_(); out["reward"].zero_()
            g.replay()
            T_used = int(out["steps"].item()) + 1
        finally:
            pol.lazy_trim, env.check_solution = False, checked
    acts = out["actions"][:, :T_used]
    assert acts.shape == ref["actions"].shape, (acts.shape, ref["actions"].shape)
    assert torch.equal(acts, ref["actions"]) and torch.equal(out["reward"], ref["reward"])
    pol.check_range()
