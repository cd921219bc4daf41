"""Host side of csrc/rr_bign.hip: encoder, decoder cache, decoder.forward and the selection for instances with 104 .. 1 024 nodes
(ATSP, RCVRP, RCVRPTW / the multi-task variants; gating NAB without or with the duration matrix, instance norm).  The on-chip kernels (rr_enc_layer, rr_rollout) hold one instance's
activations in registers / LDS and stop at 103 nodes; here the same operators (rrnco/models/nn/attn_freenet.py:417-441,
rrnco/models/decoder.py:151-329) run as row-parallel kernels over HBM / L2-resident tensors and the decode loop runs step by step,
as the reference's own policy loop does (rrnco/models/policy.py:210-228)."""
from __future__ import annotations

import torch

from .. import _lib as L
from .. import packing

E = 128
MAX_N_ONCHIP, MAX_N_BIG = 103, 1024      # (rows of up to 208 keys stay in registers, longer ones are streamed: csrc/rr_bign.hip)


def _np(n: int) -> int:
    return (n + 15) // 16 * 16


def supported(env_name: str, packed: dict, normalization: str) -> bool:
    return (env_name in ("atsp", "rcvrp", "rcvrptw") and normalization == "instance" and packed.get("nab_kind", "gating") == "gating"
            and (len(packed["nabdur"]) > 0) == (env_name == "rcvrptw"))


def _ffn_packs(packed):
    """bf16-split packs of the 2 * num_layers FFNs for k_mlp_rows<0> (packing.pack_mlp_train_batched), cached with the pack."""
    if "ffn_train" not in packed:
        sd, nl = packed["sd_ref"], packed["num_layers"]
        names = [f"encoder.net.layers.{l}.{rc}_encoding_block.feed_forward.ops.ffn" for l in range(nl) for rc in ("row", "col")]
        with torch.no_grad():
            packed["ffn_train"] = packing.pack_mlp_train_batched([sd[n + ".W1.weight"] for n in names], [sd[n + ".W1.bias"] for n in names],
                                                                 [sd[n + ".W2.weight"] for n in names], [sd[n + ".W2.bias"] for n in names])
    return packed["ffn_train"]


@torch.no_grad()
def encode(encoder, td, packed):
    """RRNetEncoder.forward (rrnco/models/encoder.py:80-112) for N > 103: init embedding in torch ops (a few [B*N, 128] products),
    every AttnFree_Block as: 2 norms, 3 Linears, NAB per edge, column softmax, AFT mixing, Linear, 2 norms, FFN, norm."""
    from . import grad_replay as GR
    lib, st = L.lib(), L.stream()
    D = td["distance_matrix"].float().contiguous()
    locs = td["locs"].float().contiguous()
    Bp, N = D.shape[0], D.shape[-1]
    M, NP, dev = Bp * N, _np(N), D.device
    sidx = td.get("sample_idx", None)
    if sidx is None:
        from .encoder import ATSPInitEmbedding, draw_sample_indices
        sidx = draw_sample_indices(encoder.init_embedding, D, "val")
    P = packed["sd_ref"]
    vtw = encoder.env_name == "rcvrptw"
    if encoder.env_name == "atsp":
        row, col = GR._init_embedding(P, locs, D, sidx)
    elif vtw:
        extra = torch.cat([td["time_windows"].float(), td["service_time"].float()[..., None]], -1)
        row, col = GR._init_embedding_vrp(P, locs, td["demand_linehaul"].float()[:, 1:], D, sidx, extra, "init_embed")
    else:
        row, col = GR._init_embedding_vrp(P, locs, td["demand"].float(), D, sidx, None, "demand_init")
    T = td["duration_matrix"].float().contiguous() if vtw else None
    bias2 = torch.empty(Bp, 2, N * N, device=D.device) if vtw else None
    row, col = row.contiguous(), col.contiguous()
    theta = torch.empty(Bp, N, N, device=dev)
    L.check(lib.rr_edge_angles(L.ptr(locs), L.ptr(theta), Bp, N, st), "rr_edge_angles")
    ffn = _ffn_packs(packed)
    new = lambda: torch.empty(Bp, N, E, device=dev)                                        # noqa: E731
    bias = torch.empty(Bp, N, N, device=dev)
    ekT, kvT = torch.empty(Bp, E, NP, device=dev), torch.empty(Bp, E, NP, device=dev)

    def lin(wp, b, x, out):
        L.check(lib.rr_linear_rows(wp, b, L.ptr(x), L.ptr(out), M, 0, None, st), "rr_linear_rows")

    def norm(x, res, g, b, out):
        L.check(lib.rr_inorm_fwd(L.ptr(x), L.ptr(res), g, b, L.ptr(out), Bp, N, st), "rr_inorm_fwd")

    for l, pair in enumerate(packed["blocks"]):
        outs = []
        for si, w in enumerate(pair):
            x, y = (row, col) if si == 0 else (col, row)
            r, c, q, k, v, yy, o = new(), new(), new(), new(), new(), new(), new()
            norm(x, None, w.n1g, w.n1b, r)
            norm(y, None, w.n2g, w.n2b, c)
            lin(w.wq, w.bq, r, q); lin(w.wk, w.bk, c, k); lin(w.wv, w.bv, c, v)
            if vtw:       # NAB with the duration matrix (attn_freenet.py:226-237): both sides of the layer by one rr_nab_dur launch
                if si == 0:
                    nr, nc = packed["nabdur"][l]
                    L.check(lib.rr_nab_dur(nr, nc, L.ptr(D), L.ptr(T), L.ptr(locs), L.ptr(bias2), Bp, N, st), "rr_nab_dur")
                bias.view(Bp, N * N).copy_(bias2[:, si])
            else:
                L.check(lib.rr_nab_pwl_fwd(w.nab, L.ptr(D), L.ptr(theta), L.ptr(bias), Bp, N, si, st), "rr_nab_pwl_fwd")
            L.check(lib.rr_colsoftmax_exp(L.ptr(k), L.ptr(v), L.ptr(ekT), L.ptr(kvT), None, Bp, N, NP, st), "rr_colsoftmax_exp")
            L.check(lib.rr_aft_mix_big(L.ptr(bias), L.ptr(q), L.ptr(ekT), L.ptr(kvT), L.ptr(yy), None, None, None, Bp, N, NP, st), "rr_aft_mix_big")
            lin(w.wp, w.bp, yy, o)                                     # project o multi_head_combine, folded (packing.py)
            o3 = q                                                     # reuse
            norm(o, None, w.n3g, w.n3b, o3)
            x1 = k
            norm(r, o3, w.f1g, w.f1b, x1)                              # ffn.norm1(r + norm3(.)) (:355, 436)
            F = v
            L.check(lib.rr_mlp_rows(ffn[2 * l + si]["fwd"], 0, L.ptr(x1), None, L.ptr(F), None, 1, M, M, st), "rr_mlp_rows")
            out = c
            norm(F, None, w.f2g, w.f2b, out)
            outs.append(out)
        row, col = outs
    return row, col


@torch.no_grad()
def encode_bn_train(encoder, td, packed, P, saves, momentum=0.1):
    """RRNetEncoder.forward for normalization='batch' with module.train() (attn_freenet.py:82-83, 102-103: BatchNorm1d over the
    flattened B * N rows, batch statistics, running statistics moved once by `momentum`) on kernels: the block as the row-parallel
    composition of `encode` above with rr_bnorm_fwd in place of the instance norm, the init embedding on rr_init_embed, N <= 103.
    `P` = parameters and BatchNorm buffers by name (grad_replay.params_and_buffers: the kernels' packs carry the running statistics
    folded for eval mode; training needs the raw gamma / beta).  `saves` (a list) receives per layer what the hand-written block
    backward reads (models/enc_backward.py; csrc/rr_enc_w.inc: EncSave) and finally {"theta"}: the gradients run on kernels too."""
    lib, st = L.lib(), L.stream()
    D = td["distance_matrix"].float().contiguous()
    locs = td["locs"].float().contiguous()
    Bp, N = D.shape[0], D.shape[-1]
    M, NP, dev = Bp * N, _np(N), D.device
    assert N <= MAX_N_ONCHIP, "the block backward kernels hold an instance's N x N weights on chip"
    sidx = td.get("sample_idx", None)
    if sidx is None:
        from .encoder import ATSPInitEmbedding, draw_sample_indices
        sidx = draw_sample_indices(encoder.init_embedding, D, "val")
    sidx = sidx.contiguous()
    vtw = encoder.env_name == "rcvrptw"
    row, col = torch.empty(Bp, N, E, device=dev), torch.empty(Bp, N, E, device=dev)
    if encoder.env_name == "atsp":
        L.check(lib.rr_init_embed(packed["init"], 0, L.ptr(D), L.ptr(locs), L.ptr(sidx), None, L.ptr(row), L.ptr(col), Bp, N,
                                  sidx.shape[-1], st), "rr_init_embed")
    else:
        vfeat = encoder.init_embedding.node_features(td).contiguous()
        L.check(lib.rr_init_embed(packed["init"], 1, L.ptr(D), L.ptr(locs), L.ptr(sidx), L.ptr(vfeat), L.ptr(row), L.ptr(col), Bp, N,
                                  sidx.shape[-1], st), "rr_init_embed")
    T = td["duration_matrix"].float().contiguous() if vtw else None
    bias2 = torch.empty(Bp, 2, N * N, device=dev) if vtw else None
    theta = torch.empty(Bp, N, N, device=dev)
    L.check(lib.rr_edge_angles(L.ptr(locs), L.ptr(theta), Bp, N, st), "rr_edge_angles")
    ffn = _ffn_packs(packed)
    new = lambda: torch.empty(Bp, N, E, device=dev)                                        # noqa: E731
    bias = torch.empty(Bp, N, N, device=dev)
    ekT, kvT = torch.empty(Bp, E, NP, device=dev), torch.empty(Bp, E, NP, device=dev)
    ws = torch.empty(256, dtype=torch.float64, device=dev)

    def lin(wp, b, x, out):
        L.check(lib.rr_linear_rows(wp, b, L.ptr(x), L.ptr(out), M, 0, None, st), "rr_linear_rows")

    def norm(x, res, pname, out, sum_out=None):
        n = pname + ".normalizer."
        L.check(lib.rr_bnorm_fwd(L.ptr(x), L.ptr(res), L.ptr(P[n + "weight"].detach()), L.ptr(P[n + "bias"].detach()), L.ptr(out), L.ptr(sum_out),
                                 L.ptr(ws), L.ptr(P[n + "running_mean"]), L.ptr(P[n + "running_var"]), float(momentum), M, st), "rr_bnorm_fwd")

    for l, pair in enumerate(packed["blocks"]):
        outs, sv = [], {"row_in": row, "col_in": col}
        for si, (w, side) in enumerate(zip(pair, ("row", "col"))):
            b = f"encoder.net.layers.{l}.{side}_encoding_block"
            x, y = (row, col) if si == 0 else (col, row)
            r, c, q, k, v, ek, yy, o = new(), new(), new(), new(), new(), new(), new(), new()
            num, den, u1, x1 = new(), new(), new(), new()
            eaT = torch.zeros(Bp, 112, 112, device=dev)
            norm(x, None, b + ".norm1", r)
            norm(y, None, b + ".norm2", c)
            lin(w.wq, w.bq, r, q); lin(w.wk, w.bk, c, k); lin(w.wv, w.bv, c, v)
            if vtw:
                if si == 0:
                    nr, nc = packed["nabdur"][l]
                    L.check(lib.rr_nab_dur(nr, nc, L.ptr(D), L.ptr(T), L.ptr(locs), L.ptr(bias2), Bp, N, st), "rr_nab_dur")
                bias.view(Bp, N * N).copy_(bias2[:, si])
            else:
                L.check(lib.rr_nab_pwl_fwd(w.nab, L.ptr(D), L.ptr(theta), L.ptr(bias), Bp, N, si, st), "rr_nab_pwl_fwd")
            L.check(lib.rr_colsoftmax_exp(L.ptr(k), L.ptr(v), L.ptr(ekT), L.ptr(kvT), L.ptr(ek), Bp, N, NP, st), "rr_colsoftmax_exp")
            L.check(lib.rr_aft_mix_big(L.ptr(bias), L.ptr(q), L.ptr(ekT), L.ptr(kvT), L.ptr(yy), L.ptr(num), L.ptr(den), L.ptr(eaT),
                                       Bp, N, NP, st), "rr_aft_mix_big")
            lin(w.wp, w.bp, yy, o)                                     # project o multi_head_combine, folded (packing.py)
            o3 = k                                                     # reuse (K is dead: ek / kvT carry it)
            norm(o, None, b + ".norm3", o3)
            norm(r, o3, b + ".feed_forward.ops.norm1", x1, sum_out=u1)  # ffn.norm1(r + norm3(.)) (:355, 436); u1 = its input
            F = new()
            L.check(lib.rr_mlp_rows(ffn[2 * l + si]["fwd"], 0, L.ptr(x1), None, L.ptr(F), None, 1, M, M, st), "rr_mlp_rows")
            out = new()
            norm(F, None, b + ".feed_forward.ops.norm2", out)
            outs.append(out)
            sv[side] = {"r": r, "c": c, "q": q, "ek": ek, "v": v, "num": num, "den": den, "y": yy, "o": o, "u1": u1, "x1": x1, "eaT": eaT}
        saves.append(sv)
        row, col = outs
    saves.append({"theta": theta})
    return row, col


@torch.no_grad()
def precompute_cache(decoder, row, col, packed):
    """decoder.py:214-232 + the step-context tables: five Linear maps on k_linear_rows; V also transposed and padded."""
    from .decoder import PrecomputedCache
    lib, st = L.lib(), L.stream()
    Bp, N, _ = row.shape
    M, NP = Bp * N, _np(N)
    cw = packed["cache"]
    K, V, Lk, cb = (torch.empty_like(row) for _ in range(4))
    ca = torch.empty_like(row) if decoder.env_name == "atsp" else None
    for wp, x, out in ((cw.wk, col, K), (cw.wv, col, V), (cw.wl, col, Lk), (cw.wcb, row, cb)) + (((cw.wca, row, ca),) if ca is not None else ()):
        L.check(lib.rr_linear_rows(wp, None, L.ptr(x), L.ptr(out), M, 0, None, st), "rr_linear_rows")
    Vt = torch.zeros(Bp, E, NP, device=row.device)
    Vt[:, :, :N] = V.transpose(1, 2)
    return PrecomputedCache(row, 0, K, Vt, Lk, ca, cb)


@torch.no_grad()
def decoder_forward(decoder, td, cache, packed):
    """decoder.py:151-206 for all rollouts -> (logits [R,N], mask [R,N])."""
    D = td["distance_matrix"].float().contiguous()
    Bp, N = D.shape[0], D.shape[-1]
    mask = td["action_mask"]
    R = mask.shape[0]
    S = max(R // Bp, 1)
    dev = D.device
    dw = packed["dec"]
    m8 = mask.to(torch.uint8).contiguous()
    cur = td["current_node"].reshape(-1).contiguous()
    logits = torch.empty(R, N, device=dev)
    io = L.DecBigIO()
    atsp = decoder.env_name == "atsp"
    keep = [m8, cur]
    ctxA, ctxB, first = cache.ctx_a, cache.ctx_b, None
    if atsp:
        if td.meta.get("i", 1) == 0:          # first step without multistart: the W_placeholder context (TSPContext), same for every rollout
            q0 = torch.as_tensor((packed["sd_ref"]["decoder.context_embedding.project_context.weight"].detach().float()
                                  @ packed["sd_ref"]["decoder.context_embedding.W_placeholder"].detach().float()))
            ctxB, ctxA = q0.view(1, 1, E).expand(Bp, N, E).contiguous(), None
            keep.append(ctxB)
        else:
            first = td["first_node"].reshape(-1).contiguous()
            keep.append(first)
    scal, nscal, Dur = None, 0, None
    if decoder.env_name == "rcvrp":           # VRPContext: vehicle_capacity - used_capacity
        rem = (td["vehicle_capacity"].reshape(-1) - td["used_capacity"].reshape(-1)).float()
        scal = torch.zeros(R, 4, device=dev); scal[:, 0] = rem
        nscal = 1
        keep.append(scal)
    elif decoder.env_name == "rcvrptw":       # MTVRPContextEmbedding (env_embeddings/context.py:51-70)
        b_of_r = torch.arange(R, device=dev) % Bp
        cap = td["vehicle_capacity"].reshape(-1).float()
        cap = cap if cap.shape[0] == R else cap[b_of_r]
        ul, ub = td["used_capacity_linehaul"].reshape(-1).float(), td["used_capacity_backhaul"].reshape(-1).float()
        ub = ub if ub.shape[0] == R else ub[b_of_r]
        lim = td["distance_limit"].reshape(-1).float()[b_of_r]
        rd = torch.nan_to_num(lim - td["current_route_length"].reshape(-1).float(), posinf=10.0)
        scal = torch.stack([cap - torch.where(ub == 0, ul, ub), td["current_time"].reshape(-1).float(),
                            td["open_route"].reshape(-1).float()[b_of_r], rd], 1).contiguous()
        nscal, Dur = 4, td["duration_matrix"].float().contiguous()
        keep += [scal, Dur]
    io.K, io.Vt, io.L, io.ctxA, io.ctxB = L.ptr(cache.glimpse_key), L.ptr(cache.glimpse_val_t), L.ptr(cache.logit_key), L.ptr(ctxA), L.ptr(ctxB)
    io.D, io.Dur, io.cur, io.first, io.scal, io.wstate = L.ptr(D), L.ptr(Dur), L.ptr(cur), L.ptr(first), L.ptr(scal), (dw.wstate if not atsp else None)
    io.mask, io.w1, io.w2, io.b1, io.b2, io.logits = L.ptr(m8), dw.w1, dw.w2, dw.b1, dw.b2, L.ptr(logits)
    io.Bp, io.N, io.NP, io.S, io.nscal = Bp, N, cache.glimpse_val_t.shape[-1], S, nscal
    io.alpha, io.beta = dw.alpha, dw.beta
    L.check(L.lib().rr_dec_fwd_big(io, L.stream()), "rr_dec_fwd_big")
    return logits, mask
