"""Diagnostic: repeats one greedy policy call and reports run-to-run differences (a race shows up as non-determinism)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
if os.environ.get('RR_LIB'):
    from rrnco_amd import _lib as _L
    _L.LIB_PATH = os.environ['RR_LIB']
which = sys.argv[1] if len(sys.argv) > 1 else "variants"
if which in ("variants", "plain"):
    import test_gpu_rcvrptw as T
    fx, w, pol, inst, env, td_in = T._setup(T.VARIANTS if which == "variants" else "rcvrptw_n20_b4_pomo")
    S = fx["S"]
elif which == "rcvrp":
    import test_gpu_rcvrp as T
    fx, w, pol, inst, env, td_in = T._setup("rcvrp_n20_b4_pomo")
    S = fx["S"]
else:
    import test_gpu_atsp as T
    import helpers as H
    from rrnco_amd.envs import ATSPEnv
    from rrnco_amd import TensorDict
    fx = H.load_fixture("atsp_n20_b4_pomo"); w = H.atsp_weights(fx); pol = H.make_policy(w, "atsp")
    env = ATSPEnv(check_solution=False, device=torch.device("cuda"))
    st = H.fixture_state(fx)
    td_in = TensorDict({k: v.cuda() for k, v in st.items()}, batch_size=[st["locs"].shape[0]])
    td_in["sample_idx"] = fx["sample_idx"].cuda()
    S = fx["S"]
ref = None
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    out = pol(env.reset(td_in), env, phase="val", decode_type="multistart_greedy", num_starts=S, return_actions=True, fused=True)
    a, ll = out["actions"].cpu(), out["log_likelihood"].cpu()
    if ref is None:
        ref = (a, ll); print("shapes", a.shape, ll.shape, "S", S)
        continue
    da = (a != ref[0]).any(1); dl = (ll - ref[1]).abs()
    if da.any() or dl.max() > 0:
        rows = torch.nonzero((dl > 0) | da).flatten().tolist()
        print("run", i, "rows differing", rows[:20], "max |dLL| %.3e" % dl.max().item(), "tour diffs", int(da.sum()))
print("done")
