"""Stage-by-stage diagnostic of the HIP path against the oracle (run on the GPU box, prints max errors)."""
import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "real-routing-nco_amd"))
import torch
from oracle import restate
from tests import helpers as H
from rrnco_amd import TensorDict, _lib as L
from rrnco_amd.envs import ATSPEnv
from rrnco_amd.ops import batchify

torch.set_printoptions(precision=6, sci_mode=False, linewidth=160)


def err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    d = (a - b).abs()
    return f"max|d|={d.max().item():.3e} mean|d|={d.mean().item():.3e} max|ref|={b.abs().max().item():.3e} nan={int(torch.isnan(a).sum())}"


def run(name):
    fx = H.load_fixture(name)
    print(f"===== {name}: B={fx['B']} N={fx['N']} S={fx['S']} aug={fx['aug']}")
    w = H.atsp_weights(fx)
    pol = H.make_policy(w)
    N, S = fx["N"], fx["S"]
    st = H.fixture_state(fx)
    env = ATSPEnv(generator_params=dict(num_loc=N), check_solution=True)
    td_in = TensorDict({k: v.cuda() for k, v in st.items()}, batch_size=[st["locs"].shape[0]])
    td_in["sample_idx"] = fx["sample_idx"].cuda()
    td = env.reset(td_in)
    print("reset D    :", err(td["distance_matrix"], fx["norm_distance"]), "exact" if torch.equal(td["distance_matrix"].cpu(), fx["norm_distance"]) else "NOT exact")
    print("reset min  :", torch.equal(td["min_distance"].cpu(), fx["min_distance"]), torch.equal(td["max_distance"].cpu(), fx["max_distance"]))
    packed = pol.packed(td.device)
    # --- oracle intermediates
    st0 = restate.atsp_reset(st)
    with torch.inference_mode():
        r0, c0 = restate.atsp_init_embedding(w, st0["locs"], st0["distance_matrix"], fx["sample_idx"])
    # --- encoder with stage dumps of layer 0
    Bp = st0["locs"].shape[0]
    dbg = torch.zeros(Bp, 2, 8, N, 128, device="cuda")
    pol.encoder._debug_buffer = dbg
    row, col = pol.encoder(td, packed=packed)
    torch.cuda.synchronize()
    ir, ic = pol.encoder._last_init
    print("init row   :", err(ir, r0)); print("init col   :", err(ic, c0))
    with torch.inference_mode():
        p = "encoder.net.layers.0.row_encoding_block"
        r = restate.instance_norm(w, p + ".norm1", r0); c = restate.instance_norm(w, p + ".norm2", c0)
        bias = restate.nab_gating(w, p + ".angle_distance_fusion", st0["locs"].float(), st0["distance_matrix"], None) * w[p + ".alpha"]
        ea = torch.exp(torch.softmax(bias, -1))
        Q = restate.lin(w, p + ".attn_free.to_q", r); K = restate.lin(w, p + ".attn_free.to_k", c); V = restate.lin(w, p + ".attn_free.to_v", c)
        Ks = torch.softmax(K, 1)
        Y = torch.sigmoid(Q) * ((ea @ (torch.exp(Ks) * V)) / (ea @ torch.exp(Ks)))
        out = restate.instance_norm(w, p + ".norm3", restate.lin(w, p + ".multi_head_combine", restate.lin(w, p + ".attn_free.project", Y)))
        x1 = restate.instance_norm(w, p + ".feed_forward.ops.norm1", r + out)
    d = dbg[:, 0].cpu()
    print("L0 row r   :", err(d[:, 0], r)); print("L0 row c   :", err(d[:, 1], c))
    print("L0 row ea  :", err(d[:, 2].reshape(Bp, -1)[:, :N * N].reshape(Bp, N, N), ea))
    print("L0 row Y   :", err(d[:, 3], Y)); print("L0 row x1  :", err(d[:, 4], x1))
    print("enc row    :", err(row, fx["row_emb"])); print("enc col    :", err(col, fx["col_emb"]))
    pol.encoder._debug_buffer = None
    # --- decoder cache
    cache = pol.decoder._precompute_cache((fx["row_emb"].cuda(), fx["col_emb"].cuda()), packed=packed)
    oc = restate.precompute_cache(w, fx["row_emb"], fx["col_emb"])
    print("cache K    :", err(cache.glimpse_key, oc["glimpse_key"])); print("cache V    :", err(cache.glimpse_val, oc["glimpse_val"]))
    print("cache L    :", err(cache.logit_key, oc["logit_key"]))
    # --- full policy (fused)
    for fused in (True, False):
        td2 = env.reset(td_in)
        out = pol(td2, env, phase="val", decode_type="multistart_greedy" if S > 1 else "greedy",
                  num_starts=S if S > 1 else None, return_actions=True, fused=fused)
        torch.cuda.synchronize()
        acts = out["actions"].cpu()
        frac, first = H.tour_agreement(acts, fx["actions"])
        print(f"policy fused={fused}: tours identical {frac*100:.2f}%  first-divergence steps: {first[first>=0][:10].tolist()}")
        print("   reward  :", err(out["reward"], fx["reward"]), " ll:", err(out["log_likelihood"], fx["log_likelihood"]))
        same = first < 0
        if same.any():
            print("   reward(same tours):", err(out["reward"].cpu()[same], fx["reward"][same]), " ll:", err(out["log_likelihood"].cpu()[same], fx["log_likelihood"][same]))
    # --- decoder.forward logits against the golden trace (state from the oracle trace)
    if "trace_logits" in fx and S > 1:
        with torch.inference_mode():
            tr = {}
            restate.atsp_policy(w, st0, fx["sample_idx"], S, "greedy", trace=tr)
        # step k state: replay actions
        td3 = env.reset(td_in)
        tdb = batchify(td3, S)
        a0 = fx["actions"][:, 0].cuda()
        tdb.set("action", a0); tdb = env.step(tdb)["next"]
        cache2 = pol.decoder._precompute_cache((fx["row_emb"].cuda(), fx["col_emb"].cuda()), packed=packed)
        for k in range(min(3, fx["trace_logits"].shape[0])):
            lg, mk = pol.decoder(tdb, cache2, S, packed=packed)
            torch.cuda.synchronize()
            print(f"dec logits step {k}:", err(lg, fx["trace_logits"][k]), " mask eq:", torch.equal(mk.cpu(), fx["trace_mask"][k]))
            tdb.set("action", fx["actions"][:, k + 1].cuda()); tdb = env.step(tdb)["next"]


if __name__ == "__main__":
    names = sys.argv[1:] or ["atsp_n20_b4_pomo", "atsp_n20_b4_greedy", "atsp_n100_b2_pomo", "atsp_n20_b2_pomo_aug8"]
    for n in names:
        try:
            run(n)
        except Exception:
            traceback.print_exc()
