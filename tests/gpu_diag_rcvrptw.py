"""RCVRPTW stage diagnostics on the GPU box."""
import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "real-routing-nco_amd"))
import torch
from oracle import restate
from tests import helpers as H
from tests.gpu_diag import err
from rrnco_amd import TensorDict
from rrnco_amd.envs import RMTVRPEnv
from rrnco_amd.ops import batchify


def run(name):
    fx = H.load_fixture(name)
    N, S, B = fx["N"], fx["S"], fx["B"]
    print(f"===== {name}: B={B} N={N} S={S}")
    w = H.rcvrptw_weights(fx)
    pol = H.make_policy(w, env_name="rcvrptw")
    inst = H.rcvrptw_instance(fx)
    env = RMTVRPEnv(generator_params=dict(num_loc=N), check_solution=False)
    td_in = TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[B])
    td_in["sample_idx"] = fx["sample_idx"].cuda()
    td = env.reset(td_in)
    st0 = restate.rmtvrp_reset(inst)
    print("reset D exact:", torch.equal(td["distance_matrix"].cpu(), fx["norm_distance"]), " mask eq:", torch.equal(td["action_mask"].cpu(), st0["action_mask"]))
    packed = pol.packed(td.device)
    dbg = torch.zeros(B, 2, 8, N + 1, 128, device="cuda")
    pol.encoder._debug_buffer = dbg
    row, col = pol.encoder(td, packed=packed)
    ir, ic = pol.encoder._last_init
    with torch.inference_mode():
        feats = torch.cat([st0["time_windows"], st0["service_time"][..., None]], -1)
        r0, c0 = restate.rcvrp_init_embedding(restate._tw_init_names(w), st0["locs"], st0["demand_linehaul"][:, 1:], st0["distance_matrix"], fx["sample_idx"], feats)
    print("init row   :", err(ir, r0)); print("init col   :", err(ic, c0))
    print("enc row    :", err(row, fx["row_emb"])); print("enc col    :", err(col, fx["col_emb"]))
    pol.encoder._debug_buffer = None
    for fused in (True, False):
        out = pol(env.reset(td_in), env, phase="val", decode_type="multistart_greedy" if S > 1 else "greedy",
                  num_starts=S if S > 1 else None, return_actions=True, fused=fused)
        torch.cuda.synchronize()
        acts = out["actions"].cpu()
        print(f"fused={fused}: T={acts.shape[1]} ref T={fx['actions'].shape[1]}")
        T = min(acts.shape[1], fx["actions"].shape[1])
        frac, first = H.tour_agreement(acts[:, :T], fx["actions"][:, :T])
        print(f"   tours identical {frac*100:.2f}%  first-divergence steps: {first[first>=0][:10].tolist()}")
        same = first < 0
        print("   reward(same):", err(out["reward"].cpu()[same], fx["reward"][same]), " ll:", err(out["log_likelihood"].cpu()[same], fx["log_likelihood"][same]))
    if "trace_logits" in fx and S > 1:
        tdb = batchify(env.reset(td_in), S)
        tdb.set("action", fx["actions"][:, 0].cuda()); tdb = env.step(tdb)["next"]
        cache = pol.decoder._precompute_cache((fx["row_emb"].cuda(), fx["col_emb"].cuda()), packed=packed)
        for k in range(min(4, fx["trace_logits"].shape[0])):
            lg, mk = pol.decoder(tdb, cache, S, packed=packed)
            print(f"dec logits step {k}:", err(lg, fx["trace_logits"][k]), " mask eq:", torch.equal(mk.cpu(), fx["trace_mask"][k]))
            tdb.set("action", fx["actions"][:, k + 1].cuda()); tdb = env.step(tdb)["next"]


if __name__ == "__main__":
    for n in sys.argv[1:] or ["rcvrptw_n20_b4_pomo", "rcvrptw_n20_b4_greedy", "rcvrptw_n100_b2_pomo"]:
        try:
            run(n)
        except Exception:
            traceback.print_exc()
