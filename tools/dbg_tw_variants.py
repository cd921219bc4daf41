"""Diagnostic: RCVRPTW n=100 fixture through library variants built with -DRR_TW_BUF=<mask> (tools/build_full_variant.sh)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
    from rrnco_amd import _lib
    _lib.LIB_PATH = _lib.LIB_PATH.replace("librrnco_hip.so", sys.argv[1])
    import torch
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import RMTVRPEnv
    from tests import helpers as H
    fx = H.load_fixture("rcvrptw_n100_b2_pomo")
    pol = H.make_policy(H.rcvrptw_weights(fx), env_name="rcvrptw")
    inst = H.rcvrptw_instance(fx)
    env = RMTVRPEnv(generator_params=dict(num_loc=fx["N"]))
    td = TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[fx["B"]]); td["sample_idx"] = fx["sample_idx"].cuda()
    out = pol(env.reset(td), env, phase="val", decode_type="multistart_greedy", num_starts=fx["S"], return_actions=True)
    a = out["actions"].cpu(); n = fx["N"]
    T = min(a.shape[1], fx["actions"].shape[1])
    same = (a[:, :T] == fx["actions"][:, :T]).all(1).float().mean().item()
    cust_ok = bool((a.sort(1).values[:, -n:] == torch.arange(1, n + 1)).all())
    missing = sorted(set(range(1, n + 1)) - set(a[0].tolist()))
    print(f"{sys.argv[1]}: T {a.shape[1]} (fixture {fx['actions'].shape[1]}), tours identical {same:.3f}, every customer once {cust_ok}, rollout 0 misses {missing[:10]}")
else:
    for m in ("librrnco_hip.so", "librrnco_hip_exact.so"):
        r = subprocess.run([sys.executable, __file__, m], capture_output=True, text=True)
        print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1])
