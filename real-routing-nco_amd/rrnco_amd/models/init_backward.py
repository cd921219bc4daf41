"""ATSP init embedding (rrnco/models/env_embeddings/atsp.py:69-121) differentiated on kernels — host side of csrc/rr_train_enc.hip's
k_linear_smallk / k_gate_bwd.

Forward, per (instance, node) row m:   node = init_embed(loc),  dist_rc = {row,col}_embed(sorted sampled distances)
                                       h = relu(gating_fc.0([node | dist])),  g = sigmoid(gating_fc.2(h)),  out = g node + (1 - g) dist
(ContextualGating :108-121, one gating network per side).  Given d loss / d out of both sides the chain runs on the library's
kernels only: the narrow Linear maps recomputed by rr_linear_smallk, the 256 -> 256 layer as four 128 x 128 rr_linear_rows blocks
each way, the scalar gate and everything elementwise around it by rr_gate_bwd, every weight gradient an rr_gemm_tn product with
fixed-order split reduction.  Torch is left with the gather / sort of the sampled distances (as in the forward) and three column
sums.  Before round 4 this was torch autograd over hipBLASLt GEMMs inside the REINFORCE step."""
from __future__ import annotations

import torch

from .. import _lib as L
from .. import packing

E = 128
_P = "encoder.init_embedding"


def supported(policy, sidx) -> bool:
    ie = policy.encoder.init_embedding
    return (policy.env_name == "atsp" and sidx.shape[-1] <= 32 and getattr(ie, "init_embed", None) is not None
            and ie.init_embed.in_features == 2 and f"{_P}.gating_network_row.gating_fc.0.weight" in dict(policy.named_parameters()))


def gate_packs(P) -> dict:
    """fp32 MFMA A-operand packs of the four 128 x 128 blocks of gating_fc.0 [256, 256] of both sides: 'f' W[o][i] for the forward
    (h_o = W[o][0] node + W[o][1] dist) and 't' W[o][i]^T for the input gradient (dcat_i = sum_o dh_o W[o][i])."""
    mats = []
    for rc in ("row", "col"):
        W0 = P[f"{_P}.gating_network_{rc}.gating_fc.0.weight"].detach().float()
        blocks = [W0[E * o:E * (o + 1), E * i:E * (i + 1)] for o in (0, 1) for i in (0, 1)]
        mats += blocks + [b.t() for b in blocks]
    pk = packing.pack_a(torch.stack([m.contiguous() for m in mats]))
    return {rc: {"f": pk[8 * s:8 * s + 4], "t": pk[8 * s + 4:8 * s + 8]} for s, rc in enumerate(("row", "col"))}


def init_embedding_backward_atsp(P, G, gp, locs, D, sidx, d_row, d_col, ws_tn, msplit):
    """Adds the gradients of every init-embedding parameter into G's buffers.  d_row / d_col [Bp,N,128] = d loss / d (row, col
    embedding before the first layer); gp = gate_packs(P)."""
    lib, st = L.lib(), L.stream()
    Bp, N, K = sidx.shape
    M, dev = Bp * N, D.device
    Kp = (K + 3) // 4 * 4
    new = lambda: torch.empty(M, E, device=dev)                                      # noqa: E731

    def lin(wp, x, out, acc=0, colsum=None, bias=None):
        L.check(lib.rr_linear_rows(L.ptr(wp), L.ptr(bias), L.ptr(x), L.ptr(out), M, acc, L.ptr(colsum), st), "rr_linear_rows")

    def smallk(x, ldx, k, name, out):
        L.check(lib.rr_linear_smallk(L.ptr(x), ldx, k, L.ptr(P[name + ".weight"].detach()), L.ptr(P[name + ".bias"].detach()) if (name + ".bias") in P else None,
                                     L.ptr(out), M, st), "rr_linear_smallk")

    def tn(a, p, lda, b, c, ldc, off=0):
        """c (+ off floats) [p][ldc] += a[:, :p]^T b"""
        L.check(lib.rr_gemm_tn(L.ptr(a), L.ptr(b), c.data_ptr() + 4 * off, 1, M, p, lda, E, ldc, 0, 0, 0, msplit, 1, L.ptr(ws_tn), st), "rr_gemm_tn")

    # ---- inputs of the narrow maps (zero-padded to 16-byte rows: rr_gemm_tn reads float4)
    loc4 = torch.zeros(M, 4, device=dev)
    loc4[:, :2] = locs.reshape(M, 2)
    xs = {}
    for rc, Dm in (("row", D), ("col", D.transpose(1, 2))):
        x = torch.zeros(M, Kp, device=dev)
        x[:, :K] = Dm.gather(2, sidx).sort(dim=-1).values.reshape(M, K)
        xs[rc] = x
    node, dnode = new(), new()
    smallk(loc4, 4, 2, _P + ".init_embed", node)
    for si, (rc, dout) in enumerate((("row", d_row), ("col", d_col))):
        q = f"{_P}.gating_network_{rc}.gating_fc"
        f, t = gp[rc]["f"], gp[rc]["t"]
        dist, ddist, hA, hB = new(), new(), new(), new()
        smallk(xs[rc], Kp, K, f"{_P}.{rc}_embed", dist)
        b0 = P[q + ".0.bias"].detach()
        lin(f[0], node, hA, bias=b0[:E]); lin(f[1], dist, hA, acc=1)
        lin(f[2], node, hB, bias=b0[E:]); lin(f[3], dist, hB, acc=1)
        io = L.GateBwdIO()
        io.hA, io.hB, io.w2, io.b2 = L.ptr(hA), L.ptr(hB), L.ptr(P[q + ".2.weight"].detach()), L.ptr(P[q + ".2.bias"].detach())
        dout = dout.contiguous()
        io.node, io.dist, io.dout = L.ptr(node), L.ptr(dist), L.ptr(dout)
        io.dnode, io.ddist = L.ptr(dnode), L.ptr(ddist)
        io.dw2, io.db2 = L.ptr(G.buf(q + ".2.weight")), L.ptr(G.buf(q + ".2.bias"))
        io.M, io.acc_node = M, si
        L.check(lib.rr_gate_bwd(io, st), "rr_gate_bwd")
        # gating_fc.0: weight blocks dW[o][i] = dh_o^T cat_i, bias = column sums of dh, input gradient dcat_i = sum_o dh_o W[o][i]
        gW0, gb0 = G.buf(q + ".0.weight"), G.buf(q + ".0.bias")
        for o, dh in enumerate((hA, hB)):
            tn(dh, E, E, node, gW0, 2 * E, o * E * 2 * E)
            tn(dh, E, E, dist, gW0, 2 * E, o * E * 2 * E + E)
        lin(t[0], hA, dnode, acc=1, colsum=gb0[:E]); lin(t[2], hB, dnode, acc=1, colsum=gb0[E:])
        lin(t[1], hA, ddist, acc=1); lin(t[3], hB, ddist, acc=1)
        # {row,col}_embed
        gT = torch.zeros(K, E, device=dev)
        tn(xs[rc], K, Kp, ddist, gT, E)
        G.buf(f"{_P}.{rc}_embed.weight").add_(gT.t())
        if f"{_P}.{rc}_embed.bias" in P:
            G.buf(f"{_P}.{rc}_embed.bias").add_(ddist.sum(0))
    gT = torch.zeros(2, E, device=dev)
    tn(loc4, 2, 4, dnode, gT, E)
    G.buf(_P + ".init_embed.weight").add_(gT.t())
    if (_P + ".init_embed.bias") in P:
        G.buf(_P + ".init_embed.bias").add_(dnode.sum(0))
