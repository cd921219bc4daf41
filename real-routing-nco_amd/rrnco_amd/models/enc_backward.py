"""Hand-written HIP backward of the encoder side of the REINFORCE step — host side of csrc/rr_train_enc.hip.

Given d loss / d (glimpse keys, values, logit keys, step-context tables) from the decoder backward (models/dec_backward.py)
this walks rrnco/models/decoder.py:214-232 (the five Linear maps of the embeddings) and the 2 x num_layers AttnFree_Blocks
(rrnco/models/nn/attn_freenet.py:417-441, 444-488) backwards on kernels, using what the training forward stored
(_lib.EncSave).  What stays in torch: the chain rule through the host-side folds (project o multi_head_combine, the NAB's
second MLP layers: tiny tensors) and the init embedding (atsp.py:69-121 / rcvrp.py:88-150: a few [Bp*N, 128] ops).
Covers instance norm + gating NAB without duration (ATSP, RCVRP: RRNetEncoder.supports_hip_backward)."""
from __future__ import annotations

import os

import torch

from .. import _lib as L
from .. import packing

E = 128
_NORMS = (("n1", "norm1"), ("n2", "norm2"), ("n3", "norm3"), ("f1", "feed_forward.ops.norm1"), ("f2", "feed_forward.ops.norm2"))


def train_packs(policy) -> dict:
    """Operand packs of the backward (transposed weights as fp32 MFMA A operands, bf16 split packs of the FFNs), rebuilt when
    the policy's weights change (keyed like policy.packed)."""
    dev = policy.param_index()["params"][0].device
    policy.packed(dev)
    key = policy._pack_cache[0]
    cached = getattr(policy, "_enc_train_pack", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    sd = policy.param_index()["P"]                # (used under no_grad below: no ~470 detached views per step)
    nl = 1 + max(int(n.split(".")[3]) for n in sd if n.startswith("encoder.net.layers."))
    keep, blocks = [], []
    with torch.no_grad():
        # all blocks' matrices packed together (a handful of launches per shape, not per matrix: this runs every step)
        names = [f"encoder.net.layers.{l}.{side}_encoding_block" for l in range(nl) for side in ("row", "col")]
        stk = lambda k: torch.stack([sd[f"{b}.{k}"].float() for b in names])                  # noqa: E731
        Wpc = packing.small_gemm(stk("multi_head_combine.weight").double(), stk("attn_free.project.weight").double()).float()
        T = packing.pack_a(torch.stack([stk("attn_free.to_q.weight"), stk("attn_free.to_k.weight"), stk("attn_free.to_v.weight"), Wpc])
                           .transpose(-1, -2).contiguous())                                     # [4][nb] packs of the TRANSPOSED matrices
        mlps = packing.pack_mlp_train_batched([sd[f"{b}.feed_forward.ops.ffn.W1.weight"] for b in names], [sd[f"{b}.feed_forward.ops.ffn.W1.bias"] for b in names],
                                              [sd[f"{b}.feed_forward.ops.ffn.W2.weight"] for b in names], [sd[f"{b}.feed_forward.ops.ffn.W2.bias"] for b in names])
        keep.append(T)
        for l in range(nl):
            pair = {}
            for si, side in enumerate(("row", "col")):
                bi = 2 * l + si
                pair[side] = {"wqT": T[0, bi], "wkT": T[1, bi], "wvT": T[2, bi], "wpcT": T[3, bi], "mlp": mlps[bi]}
            blocks.append(pair)
        Wn = sd["decoder.project_node_embeddings.weight"].float()
        Wc = sd["decoder.context_embedding.project_context.weight"].float()
        mats = [Wn[:E], Wn[E:2 * E], Wn[2 * E:], Wc[:, :E]] + ([Wc[:, E:2 * E]] if policy.env_name == "atsp" else [])
        C = packing.pack_a(torch.stack(mats).transpose(-1, -2).contiguous())
        keep.append(C)
        cache = {"wkT": C[0], "wvT": C[1], "wlT": C[2], "wc0T": C[3]}
        if policy.env_name == "atsp":
            cache["wc1T"] = C[4]
    out = {"blocks": blocks, "cache": cache, "keep": keep, "num_layers": nl}
    if policy.env_name in ("atsp", "rcvrp", "rcvrptw") and "encoder.init_embedding.gating_network_row.gating_fc.0.weight" in sd:
        from .init_backward import gate_packs
        with torch.no_grad():
            out["init_gate"] = gate_packs(sd, vrp=policy.env_name != "atsp")
    policy._enc_train_pack = (key, out)
    return out


def _nab_tab(P, p, alpha):
    """The folded table of csrc/rr_train.hip as a torch expression of the module parameters (autograd carries d tab back)."""
    from .grad_replay import _nab_table
    rows, ks, bg, bo = _nab_table(P, p)
    z = torch.zeros((), device=alpha.device, dtype=alpha.dtype)
    return torch.cat([torch.cat(rows), torch.stack([ks[0], ks[1], ks[2], ks[3], bg, bo, alpha.reshape(()), z])]).float()


def _nab_tabs_batched(P, prefixes, alphas):
    """_nab_tab for several blocks in one torch expression ([nb, 8*128+8]): stacked parameters and two batched contractions per
    family instead of a dozen tiny ops per block (the step is launch-bound on them); autograd unstacks the gradient."""
    st = lambda k: torch.stack([P[p + k] for p in prefixes])           # noqa: E731
    wo, bo = st(".out_lin.weight")[:, 0], st(".out_lin.bias")[:, 0]     # [nb,E], [nb]
    wg, bg = st(".gate.0.weight")[:, 0], st(".gate.0.bias")[:, 0]       # [nb,2E], [nb]
    rows, ks = [], []
    for f, nm in enumerate(("dist_emb", "angle_emb")):
        wgh = wg[:, f * E:(f + 1) * E]
        W2, b2 = st(f".{nm}.2.weight"), st(f".{nm}.2.bias")             # [nb,E,E], [nb,E]
        rows += [st(f".{nm}.0.weight")[:, :, 0], st(f".{nm}.0.bias"),
                 (W2 * wo[:, :, None]).sum(1), (W2 * wgh[:, :, None]).sum(1)]      # W2^T wo, W2^T wg (elementwise: no BLAS in the step)
        ks += [(wo * b2).sum(1), (wgh * b2).sum(1)]
    alpha = torch.stack([a.reshape(()) for a in alphas])
    scal = torch.stack([ks[0], ks[1], ks[2], ks[3], bg, bo, alpha, torch.zeros_like(alpha)], dim=1)
    return torch.cat(rows + [scal], dim=1).float()


def nab_grad_from_hist(tabs: torch.Tensor, hist: torch.Tensor) -> torch.Tensor:
    """d loss / d (folded table) from the per-segment moments of csrc/rr_train.hip:k_nab_hist_bwd, for a batch of blocks.
    tabs [nb, 8*128+8] (rows a_d, b_d, co_d, cg_d, a_a, b_a, co_a, cg_a + scalars, models/grad_replay._nab_table);
    hist [nb, 2*129*4 + 1].  Unit k of a family is active (a_k x + b_k > 0) on the segments after its breakpoint
    t_k = -b_k / a_k when a_k > 0, on those up to it when a_k < 0: prefix sums over the 129 segments, float64."""
    nb = tabs.shape[0]
    R = tabs[:, :8 * E].double().view(nb, 2, 4, E)                    # family, (a, b, co, cg), unit
    a, b, co, cg = R[:, :, 0], R[:, :, 1], R[:, :, 2], R[:, :, 3]
    H = hist[:, :2 * 129 * 4].double().view(nb, 2, 129, 4)
    t = torch.where(a != 0, -b / torch.where(a != 0, a, torch.ones_like(a)), torch.full_like(a, float("inf")))
    rank = torch.argsort(torch.argsort(t, dim=-1, stable=True), dim=-1, stable=True)         # position among the sorted breakpoints
    C = H.cumsum(dim=2)
    T = C[:, :, -1:, :]                                                  # [nb,2,1,4]
    Cr = C.gather(2, rank[..., None].expand(-1, -1, -1, 4))              # C[rho_k]: segments 0..rho_k
    pos, neg = (a > 0)[..., None], (a < 0)[..., None]
    A = torch.where(pos, T - Cr, torch.where(neg, Cr, torch.where((b > 0)[..., None], T.expand_as(Cr), torch.zeros_like(Cr))))
    A0, A1, A2, A3 = A.unbind(-1)
    g = torch.stack([co * A1 + cg * A3, co * A0 + cg * A2, a * A1 + b * A0, a * A3 + b * A2], dim=2)      # d a, d b, d co, d cg
    T0, T2 = T[:, :, 0, 0], T[:, :, 0, 2]
    z = torch.zeros(nb, dtype=torch.float64, device=tabs.device)
    scal = torch.stack([T0[:, 0], T2[:, 0], T0[:, 1], T2[:, 1], T2[:, 0], T0[:, 0] + T0[:, 1], hist[:, -1].double(), z], dim=1)
    return torch.cat([g.reshape(nb, 8 * E), scal], dim=1).float()


NAB_TAB_PARAMS = (".dist_emb.0.weight", ".dist_emb.0.bias", ".dist_emb.2.weight", ".dist_emb.2.bias",
                  ".angle_emb.0.weight", ".angle_emb.0.bias", ".angle_emb.2.weight", ".angle_emb.2.bias",
                  ".out_lin.weight", ".out_lin.bias", ".gate.0.weight", ".gate.0.bias")


def _nab_tab_table(policy, P, G, order, dev):
    """Device table of csrc/rr_train.hip:rr_nab_tab_bwd for the blocks in processing order: per block the addresses of the 13
    DistAngleFusion parameters (+ alpha) and the offsets of their gradient accumulators in G.flat.  Cached on the policy: parameter
    storage and the layout of the gradient buffer do not change between steps."""
    names = [[b + ".angle_distance_fusion" + s for s in NAB_TAB_PARAMS] + [b + ".alpha"] for b in order]
    key = tuple(P[n].data_ptr() for row in names for n in row) + tuple(G.offsets[n] for row in names for n in row)
    c = getattr(policy, "_nab_tab_tbl", None)
    if c is None or c[0] != key:
        for row in names:
            for n in row:
                assert P[n].dtype == torch.float32 and P[n].is_contiguous(), n
        rows = [[P[n].data_ptr() for n in row] + [G.offsets[n] for n in row] for row in names]
        c = (key, torch.tensor(rows, dtype=torch.int64).to(dev), [n for row in names for n in row])
        policy._nab_tab_tbl = c
    return c[1], c[2]


class _Grads:
    """Gradient buffers of the parameters the kernels write (zero-filled: float atomics add into them)."""

    def __init__(self, P):
        self.P, self.g = P, {}
        sizes = [(p.numel() + 3) // 4 * 4 for p in P.values()]      # 16-byte aligned views (float4 stores / atomics on rows)
        self.flat = torch.zeros(sum(sizes), dtype=torch.float32, device=next(iter(P.values())).device)   # one memset
        self.chunks = dict(zip(P.keys(), self.flat.split(sizes)))   # one op for all the views
        self.offsets, o = {}, 0                                     # (floats into `flat`: kernels that take the buffer as a whole, rr_nab_tab_bwd)
        for n, sz in zip(P.keys(), sizes):
            self.offsets[n] = o
            o += sz

    def buf(self, name):
        if name not in self.g:
            p = self.P[name]
            c = self.chunks[name]
            self.g[name] = (c if c.numel() == p.numel() else c[:p.numel()]).view(p.shape)
        return self.g[name]

    def flush(self):
        for n, g in self.g.items():
            p = self.P[n]
            g = g.to(p.dtype)
            p.grad = g if p.grad is None else p.grad + g


def encoder_backward(policy, capture, dec, D, locs, sample_idx, td):
    """capture: what policy._forward_impl kept ("enc": per-layer saves, "emb": final embeddings); dec: decoder_backward's
    result.  Accumulates the gradients of every encoder-side parameter (and project_node_embeddings / project_context)."""
    from . import grad_replay as GR
    lib, st = L.lib(), L.stream()
    P = policy.param_index()["P"]
    packs = train_packs(policy)
    saves = capture["enc"]
    theta, layers = saves[-1]["theta"], saves[:-1]
    row_emb, col_emb = capture["emb"]
    Bp, N = D.shape[0], D.shape[-1]
    M = Bp * N
    dev = D.device
    atsp = policy.env_name == "atsp"
    vtw = policy.env_name == "rcvrptw"
    Dur = td["duration_matrix"].float().contiguous() if vtw else None
    dur_todo = []             # rcvrptw: (block name, side index, d loss / d bias) — differentiated after the kernels' gradients are flushed
    G = _Grads(P)
    new = lambda: torch.empty(Bp, N, E, device=dev)                                         # noqa: E731
    MS = 256                                                                                 # row splits of the weight-gradient products
    from .dec_backward import wgrad_workspace
    ws_wg = wgrad_workspace(dev)
    ws_tn = torch.empty(MS * E * E, device=dev) if ws_wg is not None else None                                              # their partials (rr_gemm_tn reduces them in a fixed order)

    def lin(wp, x, out, acc=0, colsum=None):
        L.check(lib.rr_linear_rows(L.ptr(wp), None, L.ptr(x), L.ptr(out), M, acc, L.ptr(colsum), st), "rr_linear_rows")

    def wgrad(dy, x, gbuf, off=0, ldc=E):
        """gbuf (+ off floats) [128][ldc] += dy^T x"""
        L.check(lib.rr_gemm_tn(L.ptr(dy), L.ptr(x), gbuf.data_ptr() + 4 * off, 1, M, E, E, E, ldc, 0, 0, 0, MS, 1, L.ptr(ws_tn), st), "rr_gemm_tn")

    bn_train = GR.uses_batch_statistics(policy)      # normalization='batch', module.train(): batch statistics over all Bp * N rows
    ws_bn = torch.empty(512, dtype=torch.float64, device=dev) if bn_train else None

    def inorm(x, dy1, dy2, pname, dx, acc=0):
        if bn_train:
            L.check(lib.rr_bnorm_bwd(L.ptr(x), L.ptr(dy1), L.ptr(dy2), L.ptr(P[pname + ".normalizer.weight"].detach()), L.ptr(dx),
                                     L.ptr(G.buf(pname + ".normalizer.weight")), L.ptr(G.buf(pname + ".normalizer.bias")), L.ptr(ws_bn),
                                     M, acc, st), "rr_bnorm_bwd")
            return
        L.check(lib.rr_inorm_bwd(L.ptr(x), L.ptr(dy1), L.ptr(dy2), L.ptr(P[pname + ".normalizer.weight"].detach()), L.ptr(dx),
                                 L.ptr(G.buf(pname + ".normalizer.weight")), L.ptr(G.buf(pname + ".normalizer.bias")), Bp, N, acc, st),
                "rr_inorm_bwd")

    trace = os.environ.get("RR_NAN_TRACE", "0") == "1"      # diagnostic: name the first kernel output of the backward that is not finite

    def chk(where, **tensors):
        if trace:
            for n_, t_ in tensors.items():
                if t_ is not None and not bool(torch.isfinite(t_).all()):
                    bad = (~torch.isfinite(t_)).nonzero()
                    raise FloatingPointError(f"encoder_backward: {n_} is not finite after {where} ({bad.shape[0]} entries, first at {bad[0].tolist()})")

    with torch.no_grad():
        chk("decoder_backward", **{k_: v_ for k_, v_ in dec.items() if torch.is_tensor(v_) and v_.is_floating_point()})
        # ---- decoder.py:214-232 backwards: embeddings -> K, V, L (col) and the step-context tables (row)
        cp = packs["cache"]
        d_col, d_row = new(), new()
        gWn = G.buf("decoder.project_node_embeddings.weight")
        gWc = G.buf("decoder.context_embedding.project_context.weight")
        ldc_c = gWc.shape[1]
        lin(cp["wkT"], dec["dK"], d_col); lin(cp["wvT"], dec["dV"], d_col, 1); lin(cp["wlT"], dec["dL"], d_col, 1)
        wgrad(dec["dK"], col_emb, gWn, 0); wgrad(dec["dV"], col_emb, gWn, E * E); wgrad(dec["dL"], col_emb, gWn, 2 * E * E)
        if atsp:
            lin(cp["wc0T"], dec["dctxA"], d_row); lin(cp["wc1T"], dec["dctxB"], d_row, 1)
            wgrad(dec["dctxA"], row_emb, gWc, 0, ldc_c); wgrad(dec["dctxB"], row_emb, gWc, E, ldc_c)
        else:
            lin(cp["wc0T"], dec["dctxB"], d_row)
            wgrad(dec["dctxB"], row_emb, gWc, 0, ldc_c)
            gWc[:, E:E + dec["dwstate"].shape[0]] += dec["dwstate"].t()
        Dt = D.transpose(1, 2).contiguous()
        small = []            # (torch expression of parameters, its gradient): chained through autograd at the end
        packed = policy.packed(dev)
        nab_tabs, nab_hists = [], torch.zeros(2 * packs["num_layers"], 2 * 129 * 4 + 1, device=dev)
        # the folded project + combine Linear of every block (Wpc = Wc Wp, bpc = Wc bp + bc) as ONE expression of the parameters, its
        # gradients in one buffer each: the kernels fill the per-block views, autograd unstacks at the end (a dozen launches, not 12 x 8)
        order_all = [f"encoder.net.layers.{l}.{side}_encoding_block" for l in reversed(range(packs["num_layers"])) for side in ("row", "col")]
        dWpc_all, dbpc_all, bidx = torch.zeros(len(order_all), E, E, device=dev), torch.zeros(len(order_all), E, device=dev), 0
        tabs_all = None       # the folded NAB tables of all blocks in processing order as ONE expression (RR_NAB_TAB_PERBLOCK=1: one per block)
        # round 5: the moments -> parameter gradients on ONE kernel launch (csrc/rr_train.hip:rr_nab_tab_bwd); RR_NAB_TAB_TORCH=1 keeps the
        # torch route (nab_grad_from_hist + autograd through the fold) for A/B and for the tests that compare the two
        nab_on_kernel = not vtw and os.environ.get("RR_NAB_TAB_TORCH", "0") != "1"
        if not vtw and not nab_on_kernel and os.environ.get("RR_NAB_TAB_PERBLOCK", "0") != "1":
            order = [f"encoder.net.layers.{l}.{side}_encoding_block" for l in reversed(range(packs["num_layers"])) for side in ("row", "col")]
            with torch.enable_grad():
                tabs_all = _nab_tabs_batched(P, [b + ".angle_distance_fusion" for b in order], [P[b + ".alpha"] for b in order])
        # ---- the blocks, last layer first
        for l in reversed(range(packs["num_layers"])):
            sv = layers[l]
            n_row, n_col = new(), new()                                       # d loss / d (this layer's row / col input)
            for si, side in enumerate(("row", "col")):
                b = f"encoder.net.layers.{l}.{side}_encoding_block"
                pk, S = packs["blocks"][l][side], sv[side]
                dout = d_row if side == "row" else d_col
                x_in, y_in = (sv["row_in"], sv["col_in"]) if side == "row" else (sv["col_in"], sv["row_in"])
                dx_out, dy_out = (n_row, n_col) if side == "row" else (n_col, n_row)
                mlp = pk["mlp"]
                # x2 = ffn.norm2(x1 + FFN(x1)) (:356): recompute the norm's input, then norm and FFN backwards
                F = new()
                L.check(lib.rr_mlp_rows(mlp["fwd"], 0, L.ptr(S["x1"]), None, L.ptr(F), None, 1, M, M, st), "rr_mlp_rows")
                dF = new()
                chk(b + " recomputed F", F=F, dout=dout)
                inorm(F, dout, None, b + ".feed_forward.ops.norm2", dF)
                chk(b + " ffn.norm2 backward", dF=dF)
                dx1 = F                                                                   # reuse
                half = getattr(policy, "precision", "32") == "16-mixed"      # opt-in: one bf16 piece per operand in the FFN's backward products
                L.check(lib.rr_mlp_rows(mlp["bwd"], 3 if half else 1, L.ptr(S["x1"]), L.ptr(dF), L.ptr(dx1), None, 1, M, M, st), "rr_mlp_rows")
                f = b + ".feed_forward.ops.ffn"
                L.check((lib.rr_mlp_wgrad16 if half else lib.rr_mlp_wgrad)(mlp["wgrad"], L.ptr(S["x1"]), L.ptr(dF), L.ptr(G.buf(f + ".W1.weight")), L.ptr(G.buf(f + ".W1.bias")),
                                         L.ptr(G.buf(f + ".W2.weight")), L.ptr(G.buf(f + ".W2.bias")), None, 1, M, M, L.ptr(ws_wg), st), "rr_mlp_wgrad")
                chk(b + " FFN backward", dx1=dx1)
                # x1 = ffn.norm1(r + norm3(o)) (:355, 436)
                dU1 = dF                                                                  # reuse
                inorm(S["u1"], dx1, None, b + ".feed_forward.ops.norm1", dU1)
                chk(b + " ffn.norm1 backward", dU1=dU1)
                dO = dx1
                inorm(S["o"], dU1, None, b + ".norm3", dO)
                chk(b + " norm3 backward", dO=dO)
                # o = combine(project(y)) (:325, 435): one folded Linear Wpc = Wc Wp
                dbpc, dWpc = dbpc_all[bidx], dWpc_all[bidx]                               # (zero-filled views of one buffer each)
                bidx += 1
                dY = new()
                lin(pk["wpcT"], dO, dY, 0, dbpc)
                wgrad(dO, S["y"], dWpc)
                # AFTFull (:309-324)
                dq, dk, dv = new(), new(), new()
                dbias = torch.empty(Bp, N, N, device=dev)
                io = L.AftBwdIO()
                io.dy, io.q, io.ek, io.v, io.num, io.den, io.eaT = (L.ptr(dY), L.ptr(S["q"]), L.ptr(S["ek"]), L.ptr(S["v"]), L.ptr(S["num"]),
                                                                    L.ptr(S["den"]), L.ptr(S["eaT"]))
                io.dq, io.dk, io.dv, io.dbias, io.N = L.ptr(dq), L.ptr(dk), L.ptr(dv), L.ptr(dbias), N
                L.check(lib.rr_aft_bwd(io, Bp, st), "rr_aft_bwd")
                chk(b + " rr_aft_bwd", dY=dY, dq=dq, dk=dk, dv=dv, dbias=dbias, **{"saved " + k_: S[k_] for k_ in ("q", "ek", "v", "num", "den", "eaT")})
                if vtw:       # alpha * NAB with duration (:226-237, 265-286): d bias is all the kernels contribute
                    dur_todo.append((b, si, dbias))
                else:
                    # alpha * NAB (:427-429): the folded-table backward of csrc/rr_train.hip, chained to the module parameters by autograd
                    if tabs_all is None and not nab_on_kernel:
                        with torch.enable_grad():
                            tab = _nab_tab(P, b + ".angle_distance_fusion", P[b + ".alpha"])
                    else:
                        tab = None
                    xd = D if side == "row" else Dt
                    hist = nab_hists[len(nab_tabs)]
                    L.check(lib.rr_nab_hist_bwd(packed["blocks"][l][si].nab, L.ptr(xd), L.ptr(theta), L.ptr(dbias), L.ptr(hist),
                                                dbias.numel(), st), "rr_nab_hist_bwd")
                    nab_tabs.append(tab)
                # q = to_q(r), k = to_k(c), v = to_v(c) (:313-315)
                dr = dY                                                                   # reuse
                lin(pk["wqT"], dq, dr, 0, G.buf(b + ".attn_free.to_q.bias"))
                wgrad(dq, S["r"], G.buf(b + ".attn_free.to_q.weight"))
                dc = dO
                lin(pk["wkT"], dk, dc, 0, G.buf(b + ".attn_free.to_k.bias"))
                lin(pk["wvT"], dv, dc, 1, G.buf(b + ".attn_free.to_v.bias"))
                wgrad(dk, S["c"], G.buf(b + ".attn_free.to_k.weight"))
                wgrad(dv, S["c"], G.buf(b + ".attn_free.to_v.weight"))
                # r = norm1(x) (:421; also the residual into ffn.norm1), c = norm2(y) (:422)
                chk(b + " projections backward", dr=dr, dc=dc)
                inorm(x_in, dU1, dr, b + ".norm1", dx_out, acc=si)           # the col block adds to what the row block wrote
                inorm(y_in, dc, None, b + ".norm2", dy_out, acc=si)
                chk(b + " norm1 / norm2 backward", dx_out=dx_out, dy_out=dy_out)
            d_row, d_col = n_row, n_col
        # chain rule through the fold Wpc = Wc Wp, bpc = Wc bp + bc (attn_freenet.py:325, 435), all blocks at once, on rr_small_gemm:
        # dWc = dWpc Wp^T + dbpc (x) bp,  dWp = Wc^T dWpc,  dbp = Wc^T dbpc,  dbc = dbpc
        stk_ = lambda k: torch.stack([P[b + k].detach().float() for b in order_all])                  # noqa: E731
        Wc_all, Wp_all, bp_all = stk_(".multi_head_combine.weight"), stk_(".attn_free.project.weight"), stk_(".attn_free.project.bias")
        dWc = packing.small_gemm(dWpc_all, Wp_all, transB=True) + dbpc_all[:, :, None] * bp_all[:, None, :]
        dWp = packing.small_gemm(Wc_all, dWpc_all, transA=True)
        dbp = (Wc_all * dbpc_all[:, :, None]).sum(1)
        for i, b in enumerate(order_all):
            G.buf(b + ".multi_head_combine.weight").add_(dWc[i]); G.buf(b + ".attn_free.project.weight").add_(dWp[i])
            G.buf(b + ".attn_free.project.bias").add_(dbp[i]); G.buf(b + ".multi_head_combine.bias").add_(dbpc_all[i])
        if nab_tabs and nab_on_kernel:
            tbl, tbl_names = _nab_tab_table(policy, P, G, order_all, dev)
            L.check(lib.rr_nab_tab_bwd(L.ptr(tbl), L.ptr(nab_hists), L.ptr(G.flat), len(nab_tabs), st), "rr_nab_tab_bwd")
            for n in tbl_names:
                G.buf(n)          # (registers the view: flush hands it to the parameter)
        elif nab_tabs:        # the NAB moments of all blocks -> d (folded tables), one batched prefix-sum pass
            if tabs_all is None:
                gtabs = nab_grad_from_hist(torch.stack([t.detach() for t in nab_tabs]), nab_hists)
                small += [(t, gtabs[i]) for i, t in enumerate(nab_tabs)]
            else:
                small.append((tabs_all, nab_grad_from_hist(tabs_all.detach(), nab_hists)))
        # ---- init embedding (atsp.py:69-121, rcvrp.py:88-150) on kernels too (models/init_backward.py); RR_INIT_BWD_TORCH=1: torch autograd below
        from . import init_backward as IB
        init_on_kernels = "init_gate" in packs and IB.supported(policy, sample_idx) and os.environ.get("RR_INIT_BWD_TORCH", "0") != "1"
        if init_on_kernels:
            feats = None
            if policy.env_name == "rcvrp":
                dm = td["demand"].float()
                feats = torch.cat([torch.zeros_like(dm[:, :1]), dm], dim=1)[..., None]
            elif vtw:
                feats = torch.cat([td["demand_linehaul"].float()[..., None], td["time_windows"].float(), td["service_time"].float()[..., None]], -1)
                feats[:, 0, 0] = 0.0                                  # (the depot's demand entry: grad_replay._init_embedding_vrp prepends a zero)
            IB.init_embedding_backward(policy.env_name, P, G, packs["init_gate"], locs, D, sample_idx, d_row, d_col, ws_tn, MS, feats)
    G.flush()
    # ---- chain rule through the folds (tiny tensors) and, for the VRPs, the init embedding (autograd on [Bp*N,128] tensors)
    vrp = policy.env_name == "rcvrp"
    with torch.enable_grad():
        # duration NAB of every block: csrc/rr_train_nabdur.hip on the kernels' d bias; the angles are the same for row and col
        # blocks, cost / duration transposed (:480-486)
        Tt = Dur.transpose(1, 2).contiguous() if vtw else None
        for b, si, dbias in dur_todo:
            cost_, dur_ = (D, Dur) if si == 0 else (Dt, Tt)
            GR.nab_duration_backward_hip(P, b + ".neural_adaptive_bias", cost_, theta, dur_, P[b + ".alpha"], dbias)
        if init_on_kernels:
            pass
        elif vtw:
            extra = torch.cat([td["time_windows"].float(), td["service_time"].float()[..., None]], -1)
            row0, col0 = GR._init_embedding_vrp(P, locs, td["demand_linehaul"].float()[:, 1:], D, sample_idx, extra, "init_embed")
        elif vrp:
            row0, col0 = GR._init_embedding_vrp(P, locs, td["demand"].float(), D, sample_idx, None, "demand_init")
        else:
            row0, col0 = GR._init_embedding(P, locs, D, sample_idx)
        if init_on_kernels:
            if small:
                torch.autograd.backward([t for t, _ in small], [g for _, g in small])
        else:
            torch.autograd.backward([t for t, _ in small] + [row0, col0], [g for _, g in small] + [d_row, d_col])
