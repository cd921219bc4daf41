"""Init embeddings differentiated on kernels — host side of csrc/rr_train_enc.hip's k_linear_smallk / k_gate_bwd.

ATSP (rrnco/models/env_embeddings/atsp.py:69-121), per (instance, node) row m:
    node = init_embed(loc),  dist_rc = {row,col}_embed(sorted sampled distances)
    h = relu(gating_fc.0([node | dist])),  g = sigmoid(gating_fc.2(h)),  out_rc = g node + (1 - g) dist          (ContextualGating :108-121)
RCVRP / RCVRPTW (rcvrp.py:88-150, rcvrptw.py:44-56): the same gate between
    node = init_embed_depot(depot) for the depot row, init_embed([x, y, atan2(y - y0, x - x0)]) for a customer      (CoordinateExpert :105-124)
    dist_rc = distance_expert.{row,col}_embed(sorted sampled distances)                                            (DistanceExpert :127-150)
and then  out_rc = combine_{rc}_embed([g node + (1 - g) dist | attr(features)])  with attr = demand_init (demand) / init_embed (demand,
time window, service time).
Given d loss / d out of both sides the chain runs on the library's kernels only: the narrow Linear maps recomputed by rr_linear_smallk,
every 128 x 128 block of the wide ones by rr_linear_rows, the scalar gate and everything elementwise around it by rr_gate_bwd, every
weight gradient an rr_gemm_tn product with fixed-order split reduction.  Torch is left with the gather / sort of the sampled distances
(as in the forward), the polar angle and a few column sums.  Before round 4 this was torch autograd over hipBLASLt GEMMs inside the
REINFORCE step."""
from __future__ import annotations

import torch

from .. import _lib as L
from .. import packing

E = 128
_P = "encoder.init_embedding"


def _names(env_name: str) -> dict:
    if env_name == "atsp":
        return {"node": _P + ".init_embed", "row": _P + ".row_embed", "col": _P + ".col_embed"}
    return {"node": _P + ".coord_expert.init_embed", "depot": _P + ".coord_expert.init_embed_depot",
            "row": _P + ".distance_expert.row_embed", "col": _P + ".distance_expert.col_embed",
            "attr": _P + (".demand_init" if env_name == "rcvrp" else ".init_embed")}


def supported(policy, sidx) -> bool:
    P = policy.param_index()["P"]
    nm = _names(policy.env_name) if policy.env_name in ("atsp", "rcvrp", "rcvrptw") else None
    if nm is None or sidx.shape[-1] > 32 or f"{_P}.gating_network_row.gating_fc.0.weight" not in P:
        return False
    need = [v + s for v in nm.values() for s in (".weight", ".bias")]
    if policy.env_name != "atsp":
        need += [f"{_P}.combine_{rc}_embed{s}" for rc in ("row", "col") for s in (".weight", ".bias")]
    return all(n in P for n in need) and P[nm["node"] + ".weight"].shape[1] == (2 if policy.env_name == "atsp" else 3)


def gate_packs(P, vrp: bool = False) -> dict:
    """fp32 MFMA A-operand packs (packing.pack_a) per side: 'f' the four 128 x 128 blocks W[o][i] of gating_fc.0 [256, 256] for the
    forward (h_o = W[o][0] node + W[o][1] dist), 't' their transposes for the input gradient (dcat_i = sum_o dh_o W[o][i]); VRPs: 'c' the
    two transposed halves of combine_{rc}_embed [128, 256] (d [mix | attr] = dout W)."""
    mats, per = [], 10 if vrp else 8
    for rc in ("row", "col"):
        W0 = P[f"{_P}.gating_network_{rc}.gating_fc.0.weight"].detach().float()
        blocks = [W0[E * o:E * (o + 1), E * i:E * (i + 1)] for o in (0, 1) for i in (0, 1)]
        mats += blocks + [b.t() for b in blocks]
        if vrp:
            Wc = P[f"{_P}.combine_{rc}_embed.weight"].detach().float()
            mats += [Wc[:, :E].t(), Wc[:, E:].t()]
    pk = packing.pack_a(torch.stack([m.contiguous() for m in mats]))
    return {rc: {"f": pk[per * s:per * s + 4], "t": pk[per * s + 4:per * s + 8], "c": pk[per * s + 8:per * s + 10]}
            for s, rc in enumerate(("row", "col"))}


def init_embedding_backward(env_name, P, G, gp, locs, D, sidx, d_row, d_col, ws_tn, msplit, feats=None):
    """Adds the gradients of every init-embedding parameter into G's buffers.  d_row / d_col [Bp,N,128] = d loss / d (row, col embedding
    before the first layer); gp = gate_packs(P, vrp); feats [Bp,N,F] = the VRPs' node attributes (depot row included), F <= 4."""
    lib, st = L.lib(), L.stream()
    vrp = env_name != "atsp"
    nm = _names(env_name)
    Bp, N, K = sidx.shape
    M, dev = Bp * N, D.device
    Kp = (K + 3) // 4 * 4
    new = lambda: torch.empty(M, E, device=dev)                                      # noqa: E731

    def lin(wp, x, out, acc=0, colsum=None, bias=None):
        L.check(lib.rr_linear_rows(L.ptr(wp), L.ptr(bias), L.ptr(x), L.ptr(out), M, acc, L.ptr(colsum), st), "rr_linear_rows")

    def smallk(x, ldx, k, name, out):
        L.check(lib.rr_linear_smallk(L.ptr(x), ldx, k, L.ptr(P[name + ".weight"].detach()), L.ptr(P[name + ".bias"].detach()),
                                     L.ptr(out), M, st), "rr_linear_smallk")

    def tn(a, p, lda, b, c, ldc, off=0):
        """c (+ off floats) [p][ldc] += a[:, :p]^T b"""
        L.check(lib.rr_gemm_tn(L.ptr(a), L.ptr(b), c.data_ptr() + 4 * off, 1, M, p, lda, E, ldc, 0, 0, 0, msplit, 1, L.ptr(ws_tn), st), "rr_gemm_tn")

    def narrow_wgrad(x, k, ldx, dy, name, bias_rows=None):
        """gradients of a narrow Linear (in-features k) from its input rows x [M, ldx] and output gradient dy [M,128]"""
        gT = torch.zeros(k, E, device=dev)
        tn(x, k, ldx, dy, gT, E)
        G.buf(name + ".weight").add_(gT.t())
        G.buf(name + ".bias").add_((dy if bias_rows is None else dy.view(Bp, N, E)[:, bias_rows].reshape(-1, E)).sum(0))

    # ---- inputs of the narrow maps (zero-padded to 16-byte rows: rr_gemm_tn reads float4)
    xs = {}
    for rc, Dm in (("row", D), ("col", D.transpose(1, 2))):
        x = torch.zeros(M, Kp, device=dev)
        x[:, :K] = Dm.gather(2, sidx).sort(dim=-1).values.reshape(M, K)
        xs[rc] = x
    node, dnode = new(), new()
    xn = torch.zeros(Bp, N, 4, device=dev)
    if not vrp:
        xn[..., :2] = locs
        xn = xn.view(M, 4)
        smallk(xn, 4, 2, nm["node"], node)
    else:
        # customers: (x, y, polar angle about the depot) with a zero depot row; the depot row through its own Linear, blended in
        c = locs[:, 1:] - locs[:, :1]
        xn[:, 1:, :2] = locs[:, 1:]
        xn[:, 1:, 2] = torch.atan2(c[..., 1], c[..., 0])
        xd = torch.zeros(Bp, N, 4, device=dev)
        xd[:, 0, :2] = locs[:, 0]
        xn, xd = xn.view(M, 4), xd.view(M, 4)
        nodeD = new()
        smallk(xn, 4, 3, nm["node"], node)
        smallk(xd, 4, 2, nm["depot"], nodeD)
        node.view(Bp, N, E)[:, 0] = nodeD.view(Bp, N, E)[:, 0]
        F_ = feats.shape[-1]
        f4 = torch.zeros(M, 4, device=dev)
        f4[:, :F_] = feats.reshape(M, F_).float()
        de, dde = new(), new()
        smallk(f4, 4, F_, nm["attr"], de)
    for si, (rc, dout) in enumerate((("row", d_row), ("col", d_col))):
        q = f"{_P}.gating_network_{rc}.gating_fc"
        f, t = gp[rc]["f"], gp[rc]["t"]
        dist, ddist, hA, hB = new(), new(), new(), new()
        smallk(xs[rc], Kp, K, nm[rc], dist)
        b0 = P[q + ".0.bias"].detach()
        lin(f[0], node, hA, bias=b0[:E]); lin(f[1], dist, hA, acc=1)
        lin(f[2], node, hB, bias=b0[E:]); lin(f[3], dist, hB, acc=1)
        dout = dout.contiguous().view(M, E)
        mix = None
        if vrp:        # combine_{rc}_embed([mix | attr]) (rcvrp.py:96-101): its input gradient first, its weights once mix is known
            cn = f"{_P}.combine_{rc}_embed"
            dmix = new()
            lin(gp[rc]["c"][0], dout, dmix, colsum=G.buf(cn + ".bias"))
            lin(gp[rc]["c"][1], dout, dde, acc=si)
            mix = new()
        io = L.GateBwdIO()
        io.hA, io.hB, io.w2, io.b2 = L.ptr(hA), L.ptr(hB), L.ptr(P[q + ".2.weight"].detach()), L.ptr(P[q + ".2.bias"].detach())
        io.node, io.dist, io.dout = L.ptr(node), L.ptr(dist), L.ptr(dmix if vrp else dout)
        io.dnode, io.ddist = L.ptr(dnode), L.ptr(ddist)
        io.dw2, io.db2 = L.ptr(G.buf(q + ".2.weight")), L.ptr(G.buf(q + ".2.bias"))
        io.M, io.acc_node, io.mix = M, si, L.ptr(mix)
        L.check(lib.rr_gate_bwd(io, st), "rr_gate_bwd")
        if vrp:
            gWc = G.buf(cn + ".weight")                                           # [128, 256] = dout^T [mix | attr]
            tn(dout, E, E, mix, gWc, 2 * E, 0); tn(dout, E, E, de, gWc, 2 * E, E)
        # gating_fc.0: weight blocks dW[o][i] = dh_o^T cat_i, bias = column sums of dh, input gradient dcat_i = sum_o dh_o W[o][i]
        gW0, gb0 = G.buf(q + ".0.weight"), G.buf(q + ".0.bias")
        for o, dh in enumerate((hA, hB)):
            tn(dh, E, E, node, gW0, 2 * E, o * E * 2 * E)
            tn(dh, E, E, dist, gW0, 2 * E, o * E * 2 * E + E)
        lin(t[0], hA, dnode, acc=1, colsum=gb0[:E]); lin(t[2], hB, dnode, acc=1, colsum=gb0[E:])
        lin(t[1], hA, ddist, acc=1); lin(t[3], hB, ddist, acc=1)
        narrow_wgrad(xs[rc], K, Kp, ddist, nm[rc])
    if not vrp:
        narrow_wgrad(xn, 2, 4, dnode, nm["node"])
    else:
        narrow_wgrad(xn, 3, 4, dnode, nm["node"], bias_rows=slice(1, None))       # zero depot rows of xn: no weight contribution from them
        narrow_wgrad(xd, 2, 4, dnode, nm["depot"], bias_rows=slice(0, 1))
        narrow_wgrad(f4, F_, 4, dde, nm["attr"])
