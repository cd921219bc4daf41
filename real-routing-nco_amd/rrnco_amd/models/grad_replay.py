"""Gradient path of the REINFORCE step (BASELINE configs[4]) — round-1 form.

The reference differentiates straight through its sampling forward (rrnco/models/rl.py:96-128 under Lightning autograd).
Here the forward — encoder, sampling rollout, reward, loss and d loss / d log-likelihood — runs on the HIP kernels without a
graph; the parameter gradient is then obtained by REPLAYING the sampled tours teacher-forced through a differentiable
formulation of the same policy and calling autograd on it: torch ops on the ROCm device (hipBLAS GEMMs, MIOpen norms) for
everything except the Neural Adaptive Bias, whose forward and backward are hand-written HIP (csrc/rr_train.hip) because
differentiating it through torch ops was 73 % of the step.  The remaining torch-op backward is what later rounds replace.

Two things keep the replay cheap:
  * with the actions known, every decode step's context (first node, current node, visited set) is known up front, so the
    N-1 sequential pointer steps become ONE batched evaluation over (instance, start, step);
  * the Neural Adaptive Bias uses the same algebraic fold as the inference kernel (wo.u and wg.u are linear in the hidden
    vector: attn_freenet.py:242-289), so no E x E contraction and no [b,N,N,E] tensor is materialised; the kernels return
    d loss / d (folded table) and autograd carries it through the fold to the module parameters.
Instances are independent (instance norm is per instance), so the encoder runs in instance chunks under activation
checkpointing and the decoder chunks back-propagate into detached copies of the cache; one backward through the encoder
graph finishes the job.  ATSP (the config-5 problem), RCVRP and RCVRPTW (vrptw preset; its duration NAB runs on torch
GEMMs in the module's unfolded form).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .. import _lib as L
from torch.utils.checkpoint import checkpoint

E, HEADS = 128, 8


def _lin(P, name, x):
    return F.linear(x, P[name + ".weight"], P.get(name + ".bias"))


def _inorm(P, p, x):
    """Normalization attn_freenet.py:101-111: 'instance' (:104-105); 'layer' (no parameters, unbiased variance over nodes x
    features); 'rms' (RMSNorm :13-26, weight only) — told apart by which parameters the module has."""
    if (p + ".normalizer.running_mean") in P:
        # 'batch' in TRAIN mode (:82-83, 102-103): BatchNorm1d over the flattened B*N rows with batch statistics.  `P` then carries
        # the module's buffers too (params_and_buffers) and "__bn_momentum__": 0.1 for the training forward (updates the running
        # statistics, like the reference's forward), 0 for the backward's recomputation (same batch statistics, no second update).
        mom = float(P.get("__bn_momentum__", 0.0))
        y = F.batch_norm(x.reshape(-1, x.shape[-1]), P[p + ".normalizer.running_mean"], P[p + ".normalizer.running_var"],
                         P[p + ".normalizer.weight"], P[p + ".normalizer.bias"], bool(P.get("__bn_train__", True)), mom, 1e-5)
        return y.view_as(x)      # (eval mode: the running statistics, as the kernels' folded affine maps)
    if (p + ".normalizer.weight") not in P:
        return (x - x.mean((1, 2)).view(-1, 1, 1)) / torch.sqrt(x.var((1, 2)).view(-1, 1, 1) + 1e-05)
    if (p + ".normalizer.bias") not in P:
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * P[p + ".normalizer.weight"]
    # InstanceNorm1d over the node axis, written out: MIOpen's batch-norm backward, which F.instance_norm lands on, takes
    # 250 us per call on a [512, 100, 128] tensor — several times what these few elementwise / reduction kernels cost
    mu = x.mean(1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(1, keepdim=True)                                      # biased, eps 1e-5 (nn.InstanceNorm1d)
    return xc * torch.rsqrt(var + 1e-5) * P[p + ".normalizer.weight"] + P[p + ".normalizer.bias"]


def _nab_table(P, p):
    """Folded table of csrc/rr_train.hip (rows a_d, b_d, co_d, cg_d, a_a, b_a, co_a, cg_a + scalars), built with torch
    ops from the module parameters so that autograd carries d loss / d table back to W2, wo, wg, ... (tiny tensors)."""
    wo, bo = P[p + ".out_lin.weight"][0], P[p + ".out_lin.bias"][0]
    wg, bg = P[p + ".gate.0.weight"][0], P[p + ".gate.0.bias"][0]
    rows, ks = [], []
    for nm, wgh in (("dist_emb", wg[:E]), ("angle_emb", wg[E:])):
        W2, b2 = P[f"{p}.{nm}.2.weight"], P[f"{p}.{nm}.2.bias"]
        rows += [P[f"{p}.{nm}.0.weight"][:, 0], P[f"{p}.{nm}.0.bias"], W2.t() @ wo, W2.t() @ wgh]
        ks += [wo @ b2, wgh @ b2]
    return rows, ks, bg, bo


class _NabGatingHip(torch.autograd.Function):
    """alpha * NAB(x_d, x_a) per edge on the HIP kernels of csrc/rr_train.hip; differentiable w.r.t. the folded table."""

    @staticmethod
    def forward(ctx, tab, xd, xa):
        from .. import _lib as L
        tab, xd, xa = tab.contiguous(), xd.contiguous(), xa.contiguous()
        out = torch.empty_like(xd)
        L.check(L.lib().rr_nab_train_fwd(L.ptr(tab), L.ptr(xd), L.ptr(xa), L.ptr(out), xd.numel(), L.stream()), "rr_nab_train_fwd")
        ctx.save_for_backward(tab, xd, xa)
        return out

    @staticmethod
    def backward(ctx, gout):
        from .. import _lib as L
        tab, xd, xa = ctx.saved_tensors
        g = torch.zeros_like(tab)
        L.check(L.lib().rr_nab_train_bwd(L.ptr(tab), L.ptr(xd), L.ptr(xa), L.ptr(gout.contiguous()), L.ptr(g), xd.numel(),
                                         L.stream()), "rr_nab_train_bwd")
        return g, None, None


def _nab_folded(P, p, cost, theta, alpha):
    """alpha * DistAngleFusion (gating, no duration) attn_freenet.py:242-289, 427-429, with the second MLP layers folded
    into out_lin / gate.  On the device: HIP forward / backward kernels (no [b,N,N,E] tensor); on CPU tensors (the unit
    tests of the gradient math) the same formula in torch ops."""
    rows, ks, bg, bo = _nab_table(P, p)
    if cost.is_cuda:
        z = torch.zeros((), device=cost.device, dtype=cost.dtype)
        tab = torch.cat([torch.cat(rows), torch.stack([ks[0], ks[1], ks[2], ks[3], bg, bo, alpha.reshape(()), z])]).float()
        return _NabGatingHip.apply(tab, cost.float(), theta.float())
    outs = []
    for f, x in enumerate((cost, theta)):
        a, b, co, cg = rows[4 * f:4 * f + 4]
        h = F.relu(x.unsqueeze(-1) * a + b)                                  # [b,N,N,E]
        outs.append((h @ co + ks[2 * f], h @ cg + ks[2 * f + 1]))
    (od, gd), (oa, ga) = outs
    g = torch.sigmoid(gd + ga + bg)
    return (g * od + (1 - g) * oa + bo) * alpha


class _NabDurationFolded(torch.autograd.Function):
    """DistAngleFusion with the duration matrix in its folded form (the arithmetic of csrc/rr_encoder.hip:k_nab_dur), with a
    hand-written backward that recomputes instead of storing: per edge, h_f = relu(a_f x_f + b_f) (3 x 128), z = Mcat h + cg,
    gate = softmax((Wg2 silu(z) + bg2) / tau), bias = sum_f gate_f (co_f . h_f + ko_f) + bo, out = alpha * bias.  Through the
    unfolded module autograd kept a dozen [edges, 128..384] tensors per block alive and ran ~60 elementwise kernels over them;
    here the three contractions are single GEMMs over all edges ([M,384] x [384,128] and its two transposes) and nothing but
    the three input scalars per edge is saved.  The folded parameters are built with torch ops from the module's, so autograd
    carries their gradients back (tiny tensors)."""

    @staticmethod
    def _forward_parts(x3, a, b, Mcat, cg, co, ko, Wg2, bg2, inv_tau):
        xr = x3.repeat_interleave(E, dim=1)                                      # [M, 3E]: the family's scalar under each hidden unit
        H = torch.relu(xr * a + b)
        Z = torch.addmm(cg, H, Mcat.t())
        sg = torch.sigmoid(Z)
        zs = Z * sg
        lraw = torch.addmm(bg2, zs, Wg2.t())
        g = torch.softmax(lraw * inv_tau, dim=-1)
        po = (H * co).view(-1, 3, E).sum(-1) + ko
        return xr, H, Z, sg, zs, lraw, g, po

    @staticmethod
    def forward(ctx, x3, a, b, Mcat, cg, co, ko, Wg2, bg2, inv_tau, bo, alpha):
        _, _, _, _, _, _, g, po = _NabDurationFolded._forward_parts(x3, a, b, Mcat, cg, co, ko, Wg2, bg2, inv_tau)
        ctx.save_for_backward(x3, a, b, Mcat, cg, co, ko, Wg2, bg2, inv_tau, bo, alpha)
        return ((g * po).sum(-1) + bo) * alpha

    @staticmethod
    def backward(ctx, gout):
        x3, a, b, Mcat, cg, co, ko, Wg2, bg2, inv_tau, bo, alpha = ctx.saved_tensors
        xr, H, Z, sg, zs, lraw, g, po = _NabDurationFolded._forward_parts(x3, a, b, Mcat, cg, co, ko, Wg2, bg2, inv_tau)
        bias = (g * po).sum(-1) + bo
        dalpha = (bias * gout).sum().reshape(alpha.shape)
        dbias = gout * alpha
        dbo = dbias.sum().reshape(bo.shape)
        dpo = dbias[:, None] * g
        dg = dbias[:, None] * po
        dl = g * (dg - (g * dg).sum(-1, keepdim=True))                           # softmax backward, w.r.t. lraw * inv_tau
        dinv_tau = (dl * lraw).sum().reshape(inv_tau.shape)
        dlraw = dl * inv_tau
        dWg2 = dlraw.t() @ zs
        dbg2 = dlraw.sum(0)
        dZ = (dlraw @ Wg2) * (sg * (1 + Z * (1 - sg)))                           # silu'(z) = s (1 + z (1 - s))
        dcg = dZ.sum(0)
        dMcat = dZ.t() @ H
        dpor = dpo.repeat_interleave(E, dim=1)
        dko = dpo.sum(0)
        dco = (H * dpor).sum(0)
        dpre = (dZ @ Mcat + dpor * co) * (H > 0)
        da = (dpre * xr).sum(0)
        db = dpre.sum(0)
        return None, da, db, dMcat, dcg, dco, dko, dWg2, dbg2, dinv_tau, dbo, dalpha


class _SmallMM(torch.autograd.Function):
    """A @ B for the folds' small matrices on csrc/rr_train_enc.hip:k_small_gemm (packing.small_gemm), differentiable: no BLAS library in
    the training step (CPU tensors — the unit tests of the gradient math — take torch.matmul inside small_gemm)."""

    @staticmethod
    def forward(ctx, A, B):
        from ..packing import small_gemm
        A, B = A.contiguous(), B.contiguous()
        ctx.save_for_backward(A, B)
        return small_gemm(A, B)

    @staticmethod
    def backward(ctx, g):
        from ..packing import small_gemm
        A, B = ctx.saved_tensors
        g = g.contiguous()
        return small_gemm(g, B, transB=True), small_gemm(A, g, transA=True)


def _nab_duration_params(P, p, alpha):
    """The folded parameters of DistAngleFusion(use_duration_matrix=True) as torch expressions of the module's (packing.py's fold:
    M_f = Wg0_f W2_f, cg = sum_f Wg0_f b2_f + bg0, co_f = W2_f^T wo, ko_f = wo . b2_f): (a [384], b [384], Mcat [128,384],
    cg [128], co [384], ko [3], Wg2 [3,128], bg2 [3], inv_tau, bo, alpha)."""
    Wg0, bg0 = P[p + ".gate.0.weight"], P[p + ".gate.0.bias"]
    wo, bo = P[p + ".out_lin.weight"][0], P[p + ".out_lin.bias"][0]
    a_, b_, Ms, cos, kos = [], [], [], [], []
    cg = bg0
    for f, nm in enumerate(("dist_emb", "angle_emb", "dur_emb")):
        W2, b2 = P[f"{p}.{nm}.2.weight"], P[f"{p}.{nm}.2.bias"]
        Wg0f = Wg0[:, f * E:(f + 1) * E]
        a_.append(P[f"{p}.{nm}.0.weight"][:, 0]); b_.append(P[f"{p}.{nm}.0.bias"])
        Ms.append(_SmallMM.apply(Wg0f, W2)); cg = cg + (Wg0f * b2[None, :]).sum(1)          # Wg0_f W2, Wg0_f b2 (elementwise: no BLAS)
        cos.append((W2 * wo[:, None]).sum(0)); kos.append((wo * b2).sum())                    # W2^T wo, wo . b2
    return (torch.cat(a_), torch.cat(b_), torch.cat(Ms, dim=1), cg, torch.cat(cos), torch.stack(kos),
            P[p + ".gate.2.weight"], P[p + ".gate.2.bias"], torch.exp(-P[p + ".gate_temperature"]), bo, alpha.reshape(()))


def _nab_duration(P, p, cost, theta, dur, alpha):
    """alpha * DistAngleFusion(use_duration_matrix=True) attn_freenet.py:226-237, 265-286 through _NabDurationFolded."""
    x3 = torch.stack([cost.reshape(-1), theta.reshape(-1), dur.reshape(-1)], dim=1)
    out = _NabDurationFolded.apply(x3, *_nab_duration_params(P, p, alpha))
    return out.view(cost.shape)


def nab_duration_backward_hip(P, p, cost, theta, dur, alpha, gout):
    """d loss / d (parameters of the duration NAB of block `p`) from d loss / d bias `gout` [Bp,N,N] on the kernels of
    csrc/rr_train_nabdur.hip (fp32 MFMA; no [edges, 384] tensor ever exists), chained to the module parameters by autograd
    through the fold.  `cost`, `theta`, `dur` [Bp,N,N] contiguous (the col block passes the transposed matrices)."""
    from .. import _lib as L
    from ..packing import pack_a, pack_bf16x2
    with torch.enable_grad():
        fp = _nab_duration_params(P, p, alpha)
    a, b, Mcat, cg, co, ko, Wg2, bg2, inv_tau, bo, al = [t.detach().float().contiguous() for t in fp]
    dev = cost.device
    w = L.NabDurBwdW()
    scal = torch.cat([bg2.reshape(3), ko.reshape(3), inv_tau.reshape(1), bo.reshape(1), al.reshape(1)]).contiguous()
    mc, mct = pack_a(Mcat), pack_a(Mcat.t().contiguous())
    ms, mst = pack_bf16x2(Mcat), pack_bf16x2(Mcat.t().contiguous())      # bf16 pipe, two-piece operands (RR_NABDUR_F32=1: fp32 MFMA)
    w.a, w.b, w.co, w.cg, w.wg2, w.scal, w.mcat, w.mcatT = (L.ptr(a), L.ptr(b), L.ptr(co), L.ptr(cg), L.ptr(Wg2), L.ptr(scal),
                                                           L.ptr(mc), L.ptr(mct))
    w.mcat_s, w.mcatT_s = L.ptr(ms), L.ptr(mst)
    M = cost.numel()
    grads = torch.zeros(1680, device=dev)
    dmcat = torch.zeros(128, 384, device=dev)
    dzf = torch.empty(((M + 31) // 32) * 32 * 128, device=dev)
    L.check(L.lib().rr_nabdur_bwd(w, L.ptr(cost), L.ptr(theta), L.ptr(dur), L.ptr(gout.contiguous()), L.ptr(dzf), L.ptr(grads),
                                  L.ptr(dmcat), M, L.stream()), "rr_nabdur_bwd")
    g = grads
    gl = [g[0:384], g[384:768], dmcat, g[1152:1280], g[768:1152], g[1667:1670], g[1280:1664].view(3, 128), g[1664:1667],
          g[1670].reshape(fp[8].shape), g[1671].reshape(fp[9].shape), g[1672].reshape(fp[10].shape)]
    torch.autograd.backward(list(fp), [x.to(t.dtype).reshape(t.shape) for x, t in zip(gl, fp)])


def _block(P, p, x, y, cost, theta, dur=None):
    """AttnFree_Block.forward attn_freenet.py:417-441 (AFTFull :309-327, TransformerFFN :330-357)."""
    r = _inorm(P, p + ".norm1", x)
    c = _inorm(P, p + ".norm2", y)
    if dur is None:
        bias = _nab_folded(P, p + ".angle_distance_fusion", cost, theta, P[p + ".alpha"])
    else:
        bias = _nab_duration(P, p + ".neural_adaptive_bias", cost, theta, dur, P[p + ".alpha"])
    q, k, v = _lin(P, p + ".attn_free.to_q", r), _lin(P, p + ".attn_free.to_k", c), _lin(P, p + ".attn_free.to_v", c)
    ea = torch.exp(torch.softmax(bias, dim=-1))
    ek = torch.exp(torch.softmax(k, dim=1))
    mixed = (ea @ (ek * v)) / (ea @ ek)
    out = _lin(P, p + ".attn_free.project", torch.sigmoid(q) * mixed)
    out = _inorm(P, p + ".norm3", _lin(P, p + ".multi_head_combine", out))
    f = p + ".feed_forward.ops"
    x1 = _inorm(P, f + ".norm1", r + out)
    return _inorm(P, f + ".norm2", x1 + _lin(P, f + ".ffn.W2", F.relu(_lin(P, f + ".ffn.W1", x1))))


def _init_embedding(P, locs, D, sidx):
    """ATSPInitEmbedding atsp.py:69-91 + ContextualGating :108-121."""
    p = "encoder.init_embedding"
    node = _lin(P, p + ".init_embed", locs)
    rowd = D.gather(2, sidx).sort(dim=-1).values
    cold = D.transpose(1, 2).gather(2, sidx).sort(dim=-1).values
    out = []
    for rc, dist in (("row", _lin(P, p + ".row_embed", rowd)), ("col", _lin(P, p + ".col_embed", cold))):
        q = f"{p}.gating_network_{rc}.gating_fc"
        g = torch.sigmoid(_lin(P, q + ".2", F.relu(_lin(P, q + ".0", torch.cat([node, dist], -1)))))
        out.append(g * node + (1 - g) * dist)
    return out[0], out[1]


def _init_embedding_vrp(P, locs, demand, D, sidx, extra=None, attr="demand_init"):
    """RVRPInitEmbedding._embed_with_distance rcvrp.py:88-102 (CoordinateExpert :105-124, DistanceExpert :127-150).
    locs [b,N+1,2] with the depot first, demand [b,N]."""
    p = "encoder.init_embedding"
    depot, cities = locs[:, :1, :], locs[:, 1:, :]
    c = cities - depot
    ang = torch.atan2(c[..., 1:], c[..., :1])
    node = torch.cat([_lin(P, p + ".coord_expert.init_embed_depot", depot),
                      _lin(P, p + ".coord_expert.init_embed", torch.cat([cities, ang], dim=-1))], dim=-2)
    rowd = D.gather(2, sidx).sort(dim=-1).values
    cold = D.transpose(1, 2).gather(2, sidx).sort(dim=-1).values
    feats = torch.cat([torch.zeros_like(demand[:, :1]), demand], dim=1)[..., None]
    if extra is not None:                # rcvrptw.py:51-56: (demand, tw_start, tw_end, service), attribute layer `init_embed`
        feats = torch.cat([feats, extra], -1)
    de = _lin(P, p + "." + attr, feats)
    out = []
    for rc, dist in (("row", _lin(P, p + ".distance_expert.row_embed", rowd)), ("col", _lin(P, p + ".distance_expert.col_embed", cold))):
        q = f"{p}.gating_network_{rc}.gating_fc"
        g = torch.sigmoid(_lin(P, q + ".2", F.relu(_lin(P, q + ".0", torch.cat([node, dist], -1)))))
        out.append(_lin(P, f"{p}.combine_{rc}_embed", torch.cat([g * node + (1 - g) * dist, de], -1)))
    return out[0], out[1]


def encode(P, locs, D, sidx, num_layers, use_checkpoint=True, demand=None, extra=None, dur=None):
    """RRNetEncoder.forward encoder.py:80-112 -> row_emb, col_emb [b,N,E].  atsp; rcvrp when `demand` is given; rcvrptw
    when also `extra` (time windows, service time) and `dur` (duration matrix) are."""
    if demand is None:
        row, col = _init_embedding(P, locs, D, sidx)
    else:
        row, col = _init_embedding_vrp(P, locs, demand, D, sidx, extra, "demand_init" if extra is None else "init_embed")
    d = locs.unsqueeze(2) - locs.unsqueeze(1)
    theta = torch.atan2(d[..., 1], d[..., 0])                                # attn_freenet.py:254-262
    Dt = D.transpose(1, 2)
    Tt = None if dur is None else dur.transpose(1, 2)
    for l in range(num_layers):
        p = f"encoder.net.layers.{l}"
        if use_checkpoint:
            r = checkpoint(_block, P, p + ".row_encoding_block", row, col, D, theta, dur, use_reentrant=False)
            c = checkpoint(_block, P, p + ".col_encoding_block", col, row, Dt, theta, Tt, use_reentrant=False)
        else:
            r = _block(P, p + ".row_encoding_block", row, col, D, theta, dur)
            c = _block(P, p + ".col_encoding_block", col, row, Dt, theta, Tt)  # attn_freenet.py:480-486: D^T, Dur^T, same angles
        row, col = r, c
    return row, col


def decode_log_likelihood(P, row_emb, col_emb, D, actions, tanh_clipping=10.0, temperature=1.0):
    """Teacher-forced RRNetDecoder + process_logits + get_log_likelihood for ATSP under multistart
    (decoder.py:151-329, decoding.py:311-361, policy.py:240-242), all steps at once.
    row_emb, col_emb [b,N,E]; D [b,N,N] (normalised); actions [b,S,N] (actions[...,0] = start node) -> ll [b,S]."""
    b, S, N = actions.shape
    T = N - 1
    k, v, lk = F.linear(col_emb, P["decoder.project_node_embeddings.weight"]).chunk(3, dim=-1)      # decoder.py:214-232
    Wc = P["decoder.context_embedding.project_context.weight"]                                        # TSPContext: Linear(2E,E)
    ctx_first, ctx_cur = F.linear(row_emb, Wc[:, :E]), F.linear(row_emb, Wc[:, E:])                   # [b,N,E]
    first = actions[..., 0]
    prev = actions[..., :T]                                       # current node when step t+1's action is chosen
    target = actions[..., 1:]
    idx = lambda t, i: t.gather(1, i.reshape(b, -1, 1).expand(-1, -1, t.size(-1)))                    # noqa: E731
    q = idx(ctx_first, first).unsqueeze(2) + idx(ctx_cur, prev).view(b, S, T, E)                      # [b,S,T,E]
    onehot = F.one_hot(prev, N).to(torch.int32)
    visited = onehot.cumsum(dim=2) > 0                            # nodes visited before each decision (incl. current)
    mask = ~visited                                               # atsp/env.py:80-105 action_mask
    q = q.reshape(b, S * T, E)
    m = mask.reshape(b, 1, S * T, N)
    heads = lambda t: t.unflatten(-1, (HEADS, -1)).transpose(1, 2)                                    # noqa: E731
    h = F.scaled_dot_product_attention(heads(q), heads(k), heads(v), attn_mask=m)                     # decoder.py:308-323
    g = h.transpose(1, 2).flatten(-2) + q                                                             # :294
    g = g + F.linear(F.relu(F.linear(g, P["decoder.pointer.ffn.lins.0.weight"], P["decoder.pointer.ffn.lins.0.bias"])),
                     P["decoder.pointer.ffn.lins.1.weight"], P["decoder.pointer.ffn.lins.1.bias"])    # :296
    logits = torch.bmm(g, lk.transpose(1, 2)) / math.sqrt(E)                                          # :300-302
    bias = P["decoder.alpha"] * idx(D, prev)                                                          # :187-190
    logits = torch.log(torch.exp(logits - bias) + 1e-6)                                               # :191-198
    if tanh_clipping > 0:
        logits = torch.tanh(logits) * tanh_clipping
    logits = logits.masked_fill(~mask.reshape(b, S * T, N), float("-inf")) / temperature
    logp = F.log_softmax(logits, dim=-1).gather(-1, target.reshape(b, S * T, 1)).view(b, S, T)
    return logp.sum(-1)


@torch.no_grad()
def rcvrp_replay_states(demand, actions, cap=1.0):
    """RCVRPEnv._step / get_action_mask (rcvrp/env.py:90-122, 183-195) replayed along given routes, in the reference's own
    operation order (the capacity test compares float sums).  demand [b,N], actions [b,S,T] -> remaining capacity [b,S,T-1]
    and action masks [b,S,T-1,N+1] seen when actions[..., 1:] were chosen."""
    b, S, T = actions.shape
    N1 = demand.shape[1] + 1
    dem0 = torch.cat([torch.zeros_like(demand[:, :1]), demand], 1)             # demand by node id, 0 at the depot
    dem_s = dem0[:, None, :].expand(b, S, N1)
    used = torch.zeros(b, S, device=demand.device)
    visited = torch.zeros(b, S, N1, dtype=torch.bool, device=demand.device)
    rem, masks = [], []
    for t in range(T - 1):
        a = actions[..., t]
        used = (used + dem_s.gather(2, a[..., None])[..., 0]) * (a != 0).float()
        visited = visited.scatter(2, a[..., None], True)
        exceeds = dem_s[..., 1:] + used[..., None] > cap
        mask_loc = visited[..., 1:] | exceeds
        mask_depot = (a == 0) & (~mask_loc).any(-1)
        masks.append(~torch.cat([mask_depot[..., None], mask_loc], -1))
        rem.append(cap - used)
    return torch.stack(rem, 2), torch.stack(masks, 2)


def rcvrp_replay_states_hip(demand, actions, cap=1.0):
    """The same replay on the env's own step kernel (rr_rcvrp_step): one launch per decode step for all b * S routes."""
    b, S, T = actions.shape
    N = demand.shape[1]
    dev = demand.device
    R = b * S
    lib = L.lib()
    dem = demand.contiguous()
    acts = actions.permute(2, 1, 0).reshape(T, R).contiguous()                    # step-major, r = s * b + instance
    vcap = torch.full((R,), float(cap), device=dev)
    used = torch.zeros(R, device=dev)
    vis = torch.zeros(R, N + 1, dtype=torch.uint8, device=dev)
    cur = torch.empty(R, dtype=torch.int64, device=dev)
    done = torch.empty(R, dtype=torch.uint8, device=dev)
    masks = torch.empty(T - 1, R, N + 1, dtype=torch.uint8, device=dev)
    rems = torch.empty(T - 1, R, device=dev)
    for k in range(T - 1):
        L.check(lib.rr_rcvrp_step(L.ptr(acts[k]), L.ptr(dem), L.ptr(vcap), L.ptr(used), L.ptr(vis), L.ptr(masks[k]), L.ptr(cur),
                                  L.ptr(done), R, b, N, L.stream()), "rr_rcvrp_step")
        rems[k] = cap - used
    return rems.view(T - 1, S, b).permute(2, 1, 0), masks.view(T - 1, S, b, N + 1).permute(2, 1, 0, 3).bool()


def decode_log_likelihood_rcvrp(P, row_emb, col_emb, D, demand, actions, tanh_clipping=10.0, temperature=1.0, states=None):
    """Teacher-forced decoder for RCVRP (rl4co VRPContext: Linear(E+1,E)([emb[cur]; capacity - used]); routes of different
    lengths are padded with depot visits, whose log-probability is 0 once everything is served).  actions [b,S,T]."""
    b, S, T = actions.shape
    N1 = row_emb.shape[1]
    Td = T - 1
    rem, mask = states if states is not None else rcvrp_replay_states(demand, actions)
    k, v, lk = F.linear(col_emb, P["decoder.project_node_embeddings.weight"]).chunk(3, dim=-1)
    Wc = P["decoder.context_embedding.project_context.weight"]                  # [E, E+1]
    ctx_cur = F.linear(row_emb, Wc[:, :E])
    prev, target = actions[..., :Td], actions[..., 1:]
    idx = lambda t, i: t.gather(1, i.reshape(b, -1, 1).expand(-1, -1, t.size(-1)))                    # noqa: E731
    q = idx(ctx_cur, prev).view(b, S, Td, E) + rem[..., None] * Wc[:, E]
    q = q.reshape(b, S * Td, E)
    heads = lambda t: t.unflatten(-1, (HEADS, -1)).transpose(1, 2)                                    # noqa: E731
    h = F.scaled_dot_product_attention(heads(q), heads(k), heads(v), attn_mask=mask.reshape(b, 1, S * Td, N1))
    g = h.transpose(1, 2).flatten(-2) + q
    g = g + F.linear(F.relu(F.linear(g, P["decoder.pointer.ffn.lins.0.weight"], P["decoder.pointer.ffn.lins.0.bias"])),
                     P["decoder.pointer.ffn.lins.1.weight"], P["decoder.pointer.ffn.lins.1.bias"])
    logits = torch.bmm(g, lk.transpose(1, 2)) / math.sqrt(E)
    logits = torch.log(torch.exp(logits - P["decoder.alpha"] * idx(D, prev)) + 1e-6)
    if tanh_clipping > 0:
        logits = torch.tanh(logits) * tanh_clipping
    logits = logits.masked_fill(~mask.reshape(b, S * Td, N1), float("-inf")) / temperature
    logp = F.log_softmax(logits, dim=-1).gather(-1, target.reshape(b, S * Td, 1)).view(b, S, Td)
    return logp.sum(-1)


@torch.no_grad()
def rcvrptw_replay_states(D, Dur, demand_l, tw, service, actions, cap=1.0, variant=None):
    """RMTVRPEnv._step / get_action_mask (rmtvrp/env.py:155-215, 343-428), replayed along given routes in the env's operation
    order.  demand_l / service [b,N+1] (depot first), tw [b,N+1,2], actions [b,S,T]; `variant` = None (vrptw preset) or a dict
    with demand_backhaul [b,N+1], open_route [b] bool, distance_limit [b], backhaul_class [b] (the multi-task terms: backhaul
    loads, open routes, distance limits, class 1 / 2).  -> per decision (when actions[..., 1:] were chosen): available load,
    current time, open-route flag, remaining distance (the four MTVRP context scalars, context.py:51-70) [b,S,T-1] each, and
    the masks [b,S,T-1,N+1]."""
    b, S, T = actions.shape
    N1 = D.shape[-1]
    dev = D.device
    bi = torch.arange(b, device=dev)[:, None].expand(b, S)
    ex = lambda v: v[:, None].expand(b, S, *v.shape[1:])                                             # noqa: E731
    dl, sv, early, late = ex(demand_l), ex(service), ex(tw[..., 0]), ex(tw[..., 1])
    dur_j0, dist_j0 = ex(Dur[:, :, 0]), ex(D[:, :, 0])
    if variant is not None:
        db = ex(variant["demand_backhaul"].float())
        closed = ex((~variant["open_route"].reshape(b).bool()).float())               # [b,S]
        limit = ex(variant["distance_limit"].reshape(b).float())
        bclass = ex(variant["backhaul_class"].reshape(b).long())
    else:
        db = torch.zeros_like(dl)
        closed = torch.ones(b, S, device=dev)
        limit = torch.full((b, S), float("inf"), device=dev)
        bclass = torch.ones(b, S, dtype=torch.long, device=dev)
    t = torch.zeros(b, S, device=dev); used = torch.zeros(b, S, device=dev); ub = torch.zeros(b, S, device=dev)
    rlen = torch.zeros(b, S, device=dev)
    prev = torch.zeros(b, S, dtype=torch.long, device=dev)
    visited = torch.zeros(b, S, N1, dtype=torch.bool, device=dev)
    rems, times, rds, masks = [], [], [], []
    for k in range(T - 1):
        a = actions[..., k]
        nz = (a != 0).float()
        pick = lambda v: v.gather(2, a[..., None])[..., 0]                                            # noqa: E731
        t = nz * (torch.maximum(t + Dur[bi, prev, a], pick(early)) + pick(sv))                        # :170-172
        rlen = nz * (rlen + D[bi, prev, a])                                                          # :175-177
        used = nz * (used + pick(dl))                                                                # :189-191
        dba = pick(db)
        ub = nz * (ub + dba)                                                                         # :192-194
        visited = visited.scatter(2, a[..., None], True)
        arrival = t[..., None] + Dur[bi, a]                                                           # [b,S,N1]
        reach = arrival < late                                                                       # :361
        back = ((torch.maximum(arrival, early) + sv + dur_j0) * closed[..., None]) < late[..., 0:1]   # :364-366
        far = (rlen[..., None] + D[bi, a] + dist_j0 * closed[..., None]) > limit[..., None]           # :369-372
        ex_l = dl + used[..., None] > cap
        ex_b = db + ub[..., None] > cap                                                              # :375-380
        missing = ((dl * ~visited).sum(-1) > 0)[..., None]                                           # :384-386
        carrying = (dba > 0)[..., None]                                                              # :388-396
        ok1 = (missing & ~ex_l & ~carrying & (dl > 0)) | (~ex_b & (db > 0))                           # :397-402
        ok2 = ~ex_l & ~ex_b & ~(dl > cap - ub[..., None])                                            # :407-412
        ok = torch.where((bclass == 1)[..., None], ok1, torch.where((bclass == 2)[..., None], ok2, torch.zeros_like(ok1)))
        can = reach & back & ok & ~far & ~visited                                                    # :420-426
        can[..., 0] = ~((a == 0) & (can[..., 1:].sum(-1) > 0))                                        # :429
        masks.append(can); rems.append(cap - torch.where(ub == 0, used, ub)); times.append(t)        # context.py:55-60
        rds.append(torch.where(torch.isfinite(limit), limit - rlen, torch.full_like(rlen, 10.0)))     # nan_to_num(posinf=10)
        prev = a
    opn = (1.0 - closed)[..., None].expand(b, S, T - 1)
    return torch.stack(rems, 2), torch.stack(times, 2), opn, torch.stack(rds, 2), torch.stack(masks, 2)


def rcvrptw_replay_states_hip(D, Dur, demand_l, tw, service, actions, cap=1.0, variant=None):
    """The same replay on the env's own step kernel (rr_rmtvrp_step, the product's masks bit for bit): one launch per decode
    step for all b * S routes instead of ~45 small torch launches — the training step of RCVRPTW was launch-bound on them."""
    b, S, T = actions.shape
    N1 = D.shape[-1]
    dev = D.device
    R = b * S
    lib = L.lib()
    Dc, Tc = D.contiguous(), Dur.contiguous()
    d0, t0 = Dc[:, :, 0].contiguous(), Tc[:, :, 0].contiguous()
    dl, twc, sv = demand_l.contiguous(), tw.contiguous(), service.contiguous()
    acts = actions.permute(2, 1, 0).reshape(T, R).contiguous()                    # step-major, r = s * b + instance
    vcap = torch.full((R,), float(cap), device=dev)
    cur = torch.zeros(R, dtype=torch.int64, device=dev)
    ctime, rlen, used = torch.zeros(R, device=dev), torch.zeros(R, device=dev), torch.zeros(R, device=dev)
    vis = torch.zeros(R, N1, dtype=torch.uint8, device=dev)
    done = torch.empty(R, dtype=torch.uint8, device=dev)
    masks = torch.empty(T - 1, R, N1, dtype=torch.uint8, device=dev)
    rems, times, rds = torch.empty(T - 1, R, device=dev), torch.empty(T - 1, R, device=dev), torch.empty(T - 1, R, device=dev)
    extra, ub, lim, keep = None, None, None, None
    if variant is not None:
        ub = torch.zeros(R, device=dev)
        lim = variant["distance_limit"].reshape(b).float().contiguous()
        keep = (variant["demand_backhaul"].float().contiguous(), variant["open_route"].reshape(b).to(torch.uint8).contiguous(), lim,
                variant["backhaul_class"].reshape(b).to(torch.int32).contiguous())
        extra = L.MtvrpExtra()
        extra.demand_b, extra.used_b = L.ptr(keep[0]), L.ptr(ub)
        extra.open_route, extra.dist_limit, extra.bclass = L.ptr(keep[1]), L.ptr(keep[2]), L.ptr(keep[3])
    for k in range(T - 1):
        L.check(lib.rr_rmtvrp_step(L.ptr(acts[k]), L.ptr(Dc), L.ptr(Tc), L.ptr(d0), L.ptr(t0), L.ptr(dl), L.ptr(twc), L.ptr(sv),
                                   L.ptr(vcap), L.ptr(cur), L.ptr(ctime), L.ptr(rlen), L.ptr(used), L.ptr(vis), L.ptr(masks[k]),
                                   L.ptr(done), R, b, N1, extra, L.stream()), "rr_rmtvrp_step")
        times[k] = ctime
        if variant is None:
            rems[k] = cap - used
        else:
            rems[k] = cap - torch.where(ub == 0, used, ub)
            rds[k] = rlen
    view = lambda t: t.view(T - 1, S, b).permute(2, 1, 0)                          # noqa: E731  -> [b, S, T-1]
    if variant is None:
        opn, rd = torch.zeros(b, S, T - 1, device=dev), torch.full((b, S, T - 1), 10.0, device=dev)
    else:
        limr = lim.repeat(S)[None, :]                                              # [1, R]: r = s * b + instance
        rd = view(torch.where(torch.isfinite(limr), limr - rds, torch.full_like(rds, 10.0)))
        opn = keep[1].float()[:, None, None].expand(b, S, T - 1)
    return view(rems), view(times), opn, rd, masks.view(T - 1, S, b, N1).permute(2, 1, 0, 3).bool()


def decode_log_likelihood_rcvrptw(P, row_emb, col_emb, D, Dur, demand_l, tw, service, actions, tanh_clipping=10.0, temperature=1.0,
                                  variant=None, states=None):
    """Teacher-forced decoder for RCVRPTW / RMTVRP: MTVRPContextEmbedding (context.py:34-70: [emb[cur]; available load, current
    time, open route, remaining distance (10 without a limit)]), bias alpha*D[cur] + beta*Dur[cur] (decoder.py:187-190)."""
    b, S, T = actions.shape
    N1 = row_emb.shape[1]
    Td = T - 1
    rem, tm, opn, rd, mask = states if states is not None else rcvrptw_replay_states(D, Dur, demand_l, tw, service, actions, variant=variant)
    k, v, lk = F.linear(col_emb, P["decoder.project_node_embeddings.weight"]).chunk(3, dim=-1)
    Wc = P["decoder.context_embedding.project_context.weight"]                  # [E, E+4]
    ctx_cur = F.linear(row_emb, Wc[:, :E])
    prev, target = actions[..., :Td], actions[..., 1:]
    idx = lambda t, i: t.gather(1, i.reshape(b, -1, 1).expand(-1, -1, t.size(-1)))                    # noqa: E731
    q = (idx(ctx_cur, prev).view(b, S, Td, E) + rem[..., None] * Wc[:, E] + tm[..., None] * Wc[:, E + 1]
         + opn[..., None] * Wc[:, E + 2] + rd[..., None] * Wc[:, E + 3])
    q = q.reshape(b, S * Td, E)
    heads = lambda t: t.unflatten(-1, (HEADS, -1)).transpose(1, 2)                                    # noqa: E731
    h = F.scaled_dot_product_attention(heads(q), heads(k), heads(v), attn_mask=mask.reshape(b, 1, S * Td, N1))
    g = h.transpose(1, 2).flatten(-2) + q
    g = g + F.linear(F.relu(F.linear(g, P["decoder.pointer.ffn.lins.0.weight"], P["decoder.pointer.ffn.lins.0.bias"])),
                     P["decoder.pointer.ffn.lins.1.weight"], P["decoder.pointer.ffn.lins.1.bias"])
    logits = torch.bmm(g, lk.transpose(1, 2)) / math.sqrt(E)
    bias = P["decoder.alpha"] * idx(D, prev) + P["decoder.beta"] * idx(Dur, prev)
    logits = torch.log(torch.exp(logits - bias) + 1e-6)
    if tanh_clipping > 0:
        logits = torch.tanh(logits) * tanh_clipping
    logits = logits.masked_fill(~mask.reshape(b, S * Td, N1), float("-inf")) / temperature
    logp = F.log_softmax(logits, dim=-1).gather(-1, target.reshape(b, S * Td, 1)).view(b, S, Td)
    return logp.sum(-1)


def params_and_buffers(policy, bn_momentum=0.0):
    """named_parameters, plus — for normalization='batch' in train mode — the BatchNorm buffers and the momentum `_inorm` applies."""
    pidx = policy.param_index()
    P = dict(pidx["P"])
    if _has_running_stats(policy):
        P.update({k: v for k, v in pidx["named_buffers"] if ".normalizer." in k})
        P["__bn_momentum__"] = bn_momentum if policy.training else 0.0
        P["__bn_train__"] = bool(policy.training)      # eval mode: gradients of the running-statistics network the kernels ran
    return P


def _has_running_stats(policy) -> bool:
    """normalization='batch' modules carry running statistics; a structural fact, looked up once per policy (state_dict() detaches
    every parameter: ~470 views per call, several calls per training step)."""
    flag = getattr(policy, "_has_bn_buffers", None)
    if flag is None:
        flag = any(k.endswith(".normalizer.running_mean") for k, _ in policy.named_buffers())
        policy._has_bn_buffers = flag
    return flag


def uses_batch_statistics(policy) -> bool:
    return policy.training and _has_running_stats(policy)


def encode_for_policy(policy, td, sample_idx, bn_momentum=0.0):
    """The encoder through torch ops (the model restated for autograd above), for the one mode the fused block kernel cannot
    serve: batch statistics across instances (normalization='batch', module.train()).  -> row_emb, col_emb."""
    P = params_and_buffers(policy, bn_momentum)
    nl = 1 + max(int(n.split(".")[3]) for n in P if n.startswith("encoder.net.layers."))
    vrp, vtw = policy.env_name == "rcvrp", policy.env_name == "rcvrptw"
    demand, extra = (td["demand"].float() if vrp else None), None
    if vtw:
        demand = td["demand_linehaul"].float()[:, 1:]
        extra = torch.cat([td["time_windows"].float(), td["service_time"].float()[..., None]], -1)
    return encode(P, td["locs"].float(), td["distance_matrix"].float(), sample_idx, nl, use_checkpoint=False, demand=demand,
                  extra=extra, dur=td["duration_matrix"].float() if vtw else None)


def _check_replay_supported(policy):
    """(normalization='batch' in train mode is served since round 2: params_and_buffers / _inorm; the replay then runs on the
    WHOLE shard at once — batch statistics do not split into chunks.)"""
    return None


def replay_backward(policy, td, actions, num_starts, grad_ll, sample_idx, enc_chunk=512, dec_chunk=None, tanh_clipping=None,
                    temperature=None, vehicle_capacity=1.0):
    """Accumulate d loss / d theta into policy parameters' .grad, given d loss / d log-likelihood.

    td: the reset state the rollout started from (`locs`, normalised `distance_matrix`); actions [S*B, N] and grad_ll [S*B]
    in the reference's flattening r = s*B + b.  Returns the replayed log-likelihood [S*B] (for checking against the
    rollout's)."""
    if policy.env_name not in ("atsp", "rcvrp", "rcvrptw"):
        raise NotImplementedError(f"gradient replay for env '{policy.env_name}'")
    _check_replay_supported(policy)
    vrp, vtw = policy.env_name == "rcvrp", policy.env_name == "rcvrptw"
    if dec_chunk is None:      # instances per teacher-forced decoder evaluation: measured optimum (tools/bench_train.py --dec-chunk);
        dec_chunk = 64 if vtw else 256      # RCVRPTW routes are ~1.8 N steps long: 4x the rows per instance
    P = params_and_buffers(policy, 0.0)
    nl = 1 + max(int(n.split(".")[3]) for n in P if n.startswith("encoder.net.layers."))
    D, locs = td["distance_matrix"].float(), td["locs"].float()
    if uses_batch_statistics(policy):
        enc_chunk = D.shape[0]          # batch statistics: the whole shard in one piece
    demand = td["demand"].float() if vrp else None
    if vtw:
        Dur, dl_full = td["duration_matrix"].float(), td["demand_linehaul"].float()      # [B,N+1] with the depot zero
        tw, service = td["time_windows"].float(), td["service_time"].float()
        demand = dl_full[:, 1:]
        extra = torch.cat([tw, service[..., None]], -1)
        # multi-task terms, when the instances carry any (same test as RMTVRPEnv._reset): backhauls, open routes, limits
        variant = None
        if "demand_backhaul" in td.keys():
            v = {k: td[k] for k in ("demand_backhaul", "open_route", "distance_limit", "backhaul_class")}
            if bool((v["demand_backhaul"] != 0).any() or v["open_route"].any() or torch.isfinite(v["distance_limit"]).any()
                    or (v["backhaul_class"] != 1).any()):
                variant = v
    B, N = D.shape[0], D.shape[-1]
    S = num_starts
    acts = actions.view(S, B, actions.shape[-1]).transpose(0, 1)  # [B,S,T]
    gll = grad_ll.view(S, B).transpose(0, 1)
    ll_out = torch.empty(B, S, device=D.device)
    tanh_clipping = policy.tanh_clipping if tanh_clipping is None else tanh_clipping     # the values the rollout ran with
    temperature = policy.temperature if temperature is None else temperature
    cap = float(vehicle_capacity)
    vstates = None
    if vtw:      # the env replay is data (no gradient) and ~40 tiny launches per decode step: once for the whole shard, not per chunk
        with torch.no_grad():
            replay = rcvrptw_replay_states_hip if D.is_cuda else rcvrptw_replay_states      # (CPU: the oracle-side unit tests)
            vstates = replay(D, Dur, dl_full, tw, service, acts, cap=cap, variant=variant)
    elif vrp:
        with torch.no_grad():
            vstates = (rcvrp_replay_states_hip if D.is_cuda else rcvrp_replay_states)(demand, acts, cap=cap)
    with torch.enable_grad():
        for lo in range(0, B, enc_chunk):
            hi = min(B, lo + enc_chunk)
            # activation checkpointing only where the block is memory-heavy: the duration NAB's [b,N,N,E] tensors (RCVRPTW);
            # the ATSP / RCVRP blocks keep ~10 GB of activations per 512 instances, which a 288 GB device does not notice
            row, col = encode(P, locs[lo:hi], D[lo:hi], sample_idx[lo:hi], nl, use_checkpoint=False,
                              demand=demand[lo:hi] if (vrp or vtw) else None,
                              extra=extra[lo:hi] if vtw else None, dur=Dur[lo:hi] if vtw else None)
            row_d, col_d = row.detach().requires_grad_(), col.detach().requires_grad_()
            for a in range(lo, hi, dec_chunk):
                z = min(hi, a + dec_chunk)
                if vtw:
                    ll = decode_log_likelihood_rcvrptw(P, row_d[a - lo:z - lo], col_d[a - lo:z - lo], D[a:z], Dur[a:z], dl_full[a:z],
                                                       tw[a:z], service[a:z], acts[a:z], tanh_clipping, temperature,
                                                       states=tuple(u[a:z] for u in vstates))
                elif vrp:
                    ll = decode_log_likelihood_rcvrp(P, row_d[a - lo:z - lo], col_d[a - lo:z - lo], D[a:z], demand[a:z], acts[a:z],
                                                     tanh_clipping, temperature, states=tuple(u[a:z] for u in vstates))
                else:
                    ll = decode_log_likelihood(P, row_d[a - lo:z - lo], col_d[a - lo:z - lo], D[a:z], acts[a:z],
                                               tanh_clipping, temperature)
                ll_out[a:z] = ll.detach()
                ll.backward(gll[a:z])                             # decoder parameters + the detached embeddings
            torch.autograd.backward([row, col], [row_d.grad, col_d.grad])
    return ll_out.transpose(0, 1).reshape(-1)


def replay_backward_hip(policy, td, capture, num_starts, grad_ll, sample_idx, enc_chunk=512):
    """Same contract as replay_backward, with the decoder differentiated by the hand-written kernels of
    csrc/rr_train_dec.hip (models/dec_backward.py) on what the sampling rollout dumped: no teacher-forced re-evaluation of
    the decoder, no torch op over the S*N decoder rows.  The kernels return d loss / d (glimpse keys, values, logit keys,
    step-context tables) per instance; the encoder side (embeddings -> those five Linear maps, decoder.py:214-232) is
    differentiated by autograd from there."""
    from .dec_backward import decoder_backward
    env_name = policy.env_name
    _check_replay_supported(policy)
    atsp, vrp, vtw = env_name == "atsp", env_name == "rcvrp", env_name == "rcvrptw"
    P = params_and_buffers(policy, 0.0)
    nl = 1 + max(int(n.split(".")[3]) for n in P if n.startswith("encoder.net.layers."))
    D, locs = td["distance_matrix"].float().contiguous(), td["locs"].float()
    if uses_batch_statistics(policy):
        enc_chunk = D.shape[0]          # batch statistics: the whole shard in one piece
    Dur = td["duration_matrix"].float().contiguous() if vtw else None
    res = decoder_backward(policy, capture["cache"], capture["dump"], D, Dur, grad_ll)

    def acc(name, g):
        p = P[name]
        g = g.reshape(p.shape).to(p.dtype)
        p.grad = g if p.grad is None else p.grad + g

    acc("decoder.pointer.ffn.lins.0.weight", res["dW1"]); acc("decoder.pointer.ffn.lins.0.bias", res["db1"])
    acc("decoder.pointer.ffn.lins.1.weight", res["dW2"]); acc("decoder.pointer.ffn.lins.1.bias", res["db2"])
    acc("decoder.alpha", res["dalpha"])
    if vtw:
        acc("decoder.beta", res["dbeta"])
    if "enc" in capture:      # encoder side on the hand-written kernels too (models/enc_backward.py): no torch replay of the blocks
        from .enc_backward import encoder_backward
        encoder_backward(policy, capture, res, D, locs, sample_idx, td)
        return res["log_likelihood"]
    demand = td["demand"].float() if vrp else None
    extra = None
    if vtw:
        dl_full = td["demand_linehaul"].float()
        demand = dl_full[:, 1:]
        extra = torch.cat([td["time_windows"].float(), td["service_time"].float()[..., None]], -1)
    B = D.shape[0]
    Wn, Wc = P["decoder.project_node_embeddings.weight"], P["decoder.context_embedding.project_context.weight"]
    with torch.enable_grad():
        for lo in range(0, B, enc_chunk):
            hi = min(B, lo + enc_chunk)
            row, col = encode(P, locs[lo:hi], D[lo:hi], sample_idx[lo:hi], nl, use_checkpoint=False,
                              demand=demand[lo:hi] if (vrp or vtw) else None,
                              extra=extra[lo:hi] if vtw else None, dur=Dur[lo:hi] if vtw else None)
            k, v, lk = F.linear(col, Wn).chunk(3, dim=-1)                      # decoder.py:214-232
            outs, grads = [k, v, lk], [res["dK"][lo:hi], res["dV"][lo:hi], res["dL"][lo:hi]]
            if atsp:
                outs += [F.linear(row, Wc[:, :E]), F.linear(row, Wc[:, E:])]
                grads += [res["dctxA"][lo:hi], res["dctxB"][lo:hi]]
            else:
                outs.append(F.linear(row, Wc[:, :E]))
                grads.append(res["dctxB"][lo:hi])
            torch.autograd.backward(outs, grads)
    if not atsp:       # the state columns of project_context (VRPContext / MTVRPContextEmbedding, context.py:51-70)
        g = torch.zeros_like(Wc)
        g[:, E:E + res["dwstate"].shape[0]] = res["dwstate"].t()
        Wc.grad = g if Wc.grad is None else Wc.grad + g
    return res["log_likelihood"]
