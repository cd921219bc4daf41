"""one layer through rr_enc_layer_train (saves) and rr_enc_layer_split: where do they part?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "real-routing-nco_amd")]
import torch
from rrnco_amd import _lib as L
from tests.test_gpu_atsp import _setup
fx, w, pol, st, env, td_in = _setup("atsp_n100_b2_pomo")
dev = torch.device("cuda")
packed = pol.packed(dev)
td = env.reset(td_in)
lib = L.lib()
D = td["distance_matrix"].contiguous(); locs = td["locs"].float().contiguous()
Bp, N = D.shape[0], D.shape[-1]
row = torch.empty(Bp, N, 128, device=dev); col = torch.empty_like(row)
sidx = td["sample_idx"].contiguous()
L.check(lib.rr_init_embed(packed["init"], 0, L.ptr(D), L.ptr(locs), L.ptr(sidx), None, L.ptr(row), L.ptr(col), Bp, N, sidx.shape[-1], L.stream()), "init")
theta = torch.empty(Bp, N, N, device=dev)
L.check(lib.rr_edge_angles(L.ptr(locs), L.ptr(theta), Bp, N, L.stream()), "ang")
wr, wc = packed["blocks"][0]
# reference: training forward with saves
sv = []
for _ in range(2):
    d = {n: torch.zeros(Bp, N, 128, device=dev) for n in L.EncSave.NAMES[:-1]}
    d["eaT"] = torch.zeros(Bp, 112, 112, device=dev)
    s_ = L.EncSave()
    for n in L.EncSave.NAMES: setattr(s_, n, L.ptr(d[n]))
    sv.append((d, s_))
r1, c1 = torch.empty_like(row), torch.empty_like(col)
L.check(lib.rr_enc_layer_train(wr, wc, L.ptr(row), L.ptr(col), L.ptr(r1), L.ptr(c1), L.ptr(D), L.ptr(theta), None, Bp, N, sv[0][1], sv[1][1], L.stream()), "train")
# re-cut
stats = torch.zeros(2, 2, Bp, 2, 128, device=dev)
work = torch.zeros(6, Bp, N, 128, device=dev)
L.check(lib.rr_enc_stats(L.ptr(row), L.ptr(col), L.ptr(stats[0]), Bp, N, L.stream()), "stats")
r2, c2 = torch.empty_like(row), torch.empty_like(col)
L.check(lib.rr_enc_layer_split(wr, wc, L.ptr(row), L.ptr(col), L.ptr(r2), L.ptr(c2), L.ptr(D), L.ptr(theta), None, L.ptr(stats[0]), L.ptr(stats[1]), L.ptr(work), None, 0, Bp, N, None, L.stream()), "split")
torch.cuda.synchronize()
def cmp(name, a, b):
    d = (a - b).abs().max().item()
    print(f"{name:12s} max|diff| {d:.3e}  equal {torch.equal(a, b)}  (|ref| max {a.abs().max().item():.3f})")
# stats vs torch
for t, x in enumerate((row, col)):
    m = x.mean(1); v = x.var(1, unbiased=False)
    print("stats tensor", t, "mean err", (stats[0, t, :, 0] - m).abs().max().item(), "rstd err", (stats[0, t, :, 1] - (v + 1e-5).rsqrt()).abs().max().item())
for side in (0, 1):
    s = sv[side][0]
    print("side", side)
    cmp("V", s["v"], work[2 + side])      # V[2]: work[2], work[3]
    cmp("ratio", s["num"] * (1.0 / s["den"]), work[4 + side])
    # K is not saved raw; ek = exp(softmax(K)) over nodes
    K = work[side]
    ek = torch.exp(torch.softmax(K, dim=1))
    cmp("ek(torch)", s["ek"], ek)
cmp("row_out", r1, r2); cmp("col_out", c1, c2)
stage = int(os.environ.get("DBG_STAGE", "0"))
if stage in (1, 2):
    for side in (0, 1): cmp(("num", "den")[stage - 1], sv[side][0][("num", "den")[stage - 1]], work[4 + side])
if stage >= 3:
    nm = {3: "y", 4: "o", 5: "u1", 6: "x1"}[stage]
    cmp(nm + " row", sv[0][0][nm], r2); cmp(nm + " col", sv[1][0][nm], c2)
if stage == 2:
    for side in (0, 1):
        df = (sv[side][0]["den"] - work[4 + side]) != 0
        print("side", side, "den differs at", int(df.sum()), "elements; per-feature counts (inst 0):", df[0].sum(0).nonzero().flatten().tolist()[:20], df[0].sum(0)[df[0].sum(0) > 0].tolist()[:20])
        # node sums of ek the kernel's way vs torch
        ek = sv[side][0]["ek"]
        print("   sum_nodes ek (fp32 torch) vs implied:", ek.sum(1)[0, :6].tolist())
if stage == 2:
    for side in (0, 1):
        s_ = sv[side][0]
        ea = s_["eaT"][:, :N, :N].transpose(1, 2).double()        # [b][i][j]
        den64 = ea @ s_["ek"].double()
        idx = ((s_["den"] - work[4 + side]) != 0).nonzero()
        for b_, n_, f_ in idx.tolist():
            print(f"side {side} b {b_} node {n_} feat {f_}: old {s_['den'][b_, n_, f_].item():.9f} new {work[4 + side][b_, n_, f_].item():.9f} f64 {den64[b_, n_, f_].item():.9f}")
