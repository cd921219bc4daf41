"""Weight preparation for the HIP kernels: MFMA A-operand packing and algebraic folding of the NAB.

Nothing here touches per-instance data; it runs once per set of weights (cached on the policy).
"""
from __future__ import annotations

import contextlib
import math

import torch

from . import _lib as L

E = 128


@torch.no_grad()
def weights_fingerprint(module, tensors=None) -> tuple:
    """Per-parameter L2 norms (one multi-tensor launch, one host read) for the pack caches.  Version counters alone are not
    enough: fused optimizers (torch.optim.Adam(fused=True)) update the parameters in place WITHOUT bumping `_version`, and a
    stale pack would silently roll out with the previous weights."""
    if tensors is None:        # (`tensors`: the caller's list of the module's parameters + buffers, when it has walked the tree already)
        tensors = list(module.parameters()) + list(module.buffers())
    ps = [q for q in tensors if q.is_floating_point() and q.numel() > 0]
    if not ps:
        return ()
    norms = torch._foreach_norm(ps)
    if len({(n.dtype, n.device) for n in norms}) > 1:      # mixed precision / devices: bring the scalars together first
        norms = [n.to(device=norms[0].device, dtype=torch.float64) for n in norms]
    return tuple(torch.stack(norms).tolist())


def pack_a(Wm: torch.Tensor) -> torch.Tensor:
    """[M,K] -> [M/16, K/16, 64, 4] float32: lane (i = l&15, g = l>>4) of tile t, k-group kk holds
    W[16t+i][16kk+4g+m], m = 0..3 (csrc/rr_common.h).  M, K zero-padded to multiples of 16."""
    *lead, M, K = Wm.shape                     # leading dims = a batch of matrices packed by the same few launches
    Mp, Kp = (M + 15) // 16 * 16, (K + 15) // 16 * 16
    if (Mp, Kp) != (M, K):
        Wp = torch.zeros(*lead, Mp, Kp, dtype=torch.float32, device=Wm.device)
        Wp[..., :M, :K] = Wm
    else:
        Wp = Wm.float()
    n = len(lead)
    x = Wp.reshape(*lead, Mp // 16, 16, Kp // 16, 4, 4).permute(*range(n), n, n + 2, n + 3, n + 1, n + 4).contiguous()
    return x.view(*lead, Mp // 16, Kp // 16, 64, 4)


def small_gemm(A: torch.Tensor, B: torch.Tensor, transA: bool = False, transB: bool = False) -> torch.Tensor:
    """op(A) @ op(B) for [batch, ., .] (or 2-D) float32 / float64 DEVICE tensors on csrc/rr_train_enc.hip:k_small_gemm — the weight
    folds' products without a BLAS library in the per-step repack; CPU tensors (unit tests of the folds) take torch.matmul."""
    if not A.is_cuda:
        return torch.matmul(A.transpose(-1, -2) if transA else A, B.transpose(-1, -2) if transB else B)
    from . import _lib as L
    squeeze = A.dim() == 2
    A3, B3 = (A[None], B[None]) if squeeze else (A, B)
    A3, B3 = A3.contiguous(), B3.contiguous()
    assert A3.dtype == B3.dtype and A3.dtype in (torch.float32, torch.float64) and A3.shape[0] == B3.shape[0]
    nb = A3.shape[0]
    M, K = (A3.shape[2], A3.shape[1]) if transA else (A3.shape[1], A3.shape[2])
    N, K2 = (B3.shape[1], B3.shape[2]) if transB else (B3.shape[2], B3.shape[1])
    assert K == K2, (A.shape, B.shape, transA, transB)
    Cm = torch.empty(nb, M, N, dtype=A3.dtype, device=A3.device)
    L.check(L.lib().rr_small_gemm(L.ptr(A3), L.ptr(B3), L.ptr(Cm), nb, M, N, K, int(transA), int(transB), int(A3.dtype == torch.float64),
                                  L.stream()), "rr_small_gemm")
    return Cm[0] if squeeze else Cm


import threading as _threading

# > 0 inside force_fp32(): the range guard's second pass (models/policy.py) runs every kernel on the fp32 MFMA.  PER THREAD: with
# parallel.run_on_streams two policies run on two host threads, and a retry in one must not flip the other's kernel selection (and
# pack-cache key) in the middle of its call.
_TLS = _threading.local()


def mlp_split_enabled() -> bool:
    """Encoder GEMMs, decoder cache and the whole rollout on the fp16 matrix pipe with two-piece split fp32 operands: the default
    since round 2 (the whole GPU suite — golden tours, embeddings to 2e-4, gradients — is green with it, and the full-size
    rollouts equal the fp32-MFMA build's, tests/test_gpu_fullsize.py); RR_MLP_SPLIT=0 packs for and runs the all-fp32-MFMA kernels,
    and so does a call whose operands left the fp16 range (force_fp32, the range guard of models/policy.py)."""
    import os
    return getattr(_TLS, "force_fp32", 0) == 0 and os.environ.get("RR_MLP_SPLIT", "1") != "0"


class force_fp32:
    """Context: every kernel of the policy on the fp32 MFMA, for the calling THREAD (the pack cache is keyed by mlp_split_enabled();
    the C side selects by which weight images the pack carries, not by the environment)."""

    def __enter__(self):
        _TLS.force_fp32 = getattr(_TLS, "force_fp32", 0) + 1

    def __exit__(self, *exc):
        _TLS.force_fp32 -= 1
        return False


F16U_WEIGHT_SCALE_LOG2 = 6      # csrc/rr_common.h RR_WS
F16_LIMIT = 65504.0


def pack_a_bf16x3(Wm: torch.Tensor) -> torch.Tensor:
    """[M,K] fp32 -> [M/16, K/32, 3, 64, 8] bf16: A operands of v_mfma_f32_16x16x32_bf16 for the three pieces
    W = hi + mid + lo (each rounded to nearest bf16 of what is left; the sum is exactly W).  Lane (i = l&15, g = l>>4) of
    tile t, k-slice s holds, for e = 0..7, W[16t+i][32s + 4g + e] (e < 4) and W[16t+i][32s + 16 + 4g + e - 4] (e >= 4): the
    order in which a lane owns the values of two consecutive C-layout tiles (csrc/rr_rollout_w.inc, SPLIT)."""
    *lead, M, K = Wm.shape                                   # leading dims: a batch of matrices
    assert M % 16 == 0 and K % 32 == 0
    n = len(lead)
    W = Wm.detach().float()                                  # stays on the weights' device: this runs on every repack
    hi = W.to(torch.bfloat16)
    r1 = W - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    x = torch.stack((hi, mid, lo), dim=n).reshape(*lead, 3, M // 16, 16, K // 32, 2, 4, 4)      # piece, t, i, s, half, g, e4
    x = x.permute(*range(n), n + 1, n + 3, n, n + 5, n + 2, n + 4, n + 6)                       # t, s, piece, g, i, half, e4
    return x.reshape(*lead, M // 16, K // 32, 3, 64, 8).contiguous()                            # [t][s][piece][lane][8]



def pack_a_f16x2(Wm: torch.Tensor) -> torch.Tensor:
    """[M,K] fp32 -> [M/16, K/32, 2, 64, 8] fp16: A operands of v_mfma_f32_16x16x32_f16 for the two pieces of
    W = hi + 2^-11 lo' (hi = fp16(W), lo' = fp16(2^11 (W - hi)); csrc/rr_common.h), in the lane / k order of pack_a_bf16x3."""
    *lead, M, K = Wm.shape
    assert M % 16 == 0 and K % 32 == 0
    n = len(lead)
    W = Wm.detach().float()
    hi = W.to(torch.float16)
    lo = ((W - hi.float()) * 2048.0).to(torch.float16)
    x = torch.stack((hi, lo), dim=n).reshape(*lead, 2, M // 16, 16, K // 32, 2, 4, 4)           # piece, t, i, s, half, g, e4
    x = x.permute(*range(n), n + 1, n + 3, n, n + 5, n + 2, n + 4, n + 6)                       # t, s, piece, g, i, half, e4
    return x.reshape(*lead, M // 16, K // 32, 2, 64, 8).contiguous()                            # [t][s][piece][lane][8]


def pack_a_f16u(Wm: torch.Tensor, scale_log2: int = F16U_WEIGHT_SCALE_LOG2) -> torch.Tensor:
    """[M,K] fp32 -> [M/16, K/32, 2, 64, 8] fp16: A operands of v_mfma_f32_16x16x32_f16 for the two pieces of the SECOND form
    (csrc/rr_common.h): W~ = 2^scale_log2 W (exact), hi = fp16(W~), lo = fp16(W~ - hi) — no scale between the pieces, so all
    three partial products of a split product share one accumulator; lane / k order of pack_a_bf16x3.  The rollout's pointer MLP."""
    *lead, M, K = Wm.shape
    assert M % 16 == 0 and K % 32 == 0
    n = len(lead)
    W = Wm.detach().float() * float(2 ** scale_log2)
    hi = W.to(torch.float16)
    lo = (W - hi.float()).to(torch.float16)
    x = torch.stack((hi, lo), dim=n).reshape(*lead, 2, M // 16, 16, K // 32, 2, 4, 4)           # piece, t, i, s, half, g, e4
    x = x.permute(*range(n), n + 1, n + 3, n, n + 5, n + 2, n + 4, n + 6)                       # t, s, piece, g, i, half, e4
    return x.reshape(*lead, M // 16, K // 32, 2, 64, 8).contiguous()                            # [t][s][piece][lane][8]


def f16_range_status(tensors, scale_log2: int = 0) -> torch.Tensor:
    """int32[1] on the tensors' device: 2 if any value of any tensor is non-finite or leaves the fp16 range after the scale
    (|2^s x| >= 65504), else 0 — bit 1 of the range guard's status word (models/policy.py); no host synchronisation."""
    lim = F16_LIMIT / float(2 ** scale_log2)
    bad = None
    for t in tensors:
        b = ~(t.detach().abs().max() < lim)          # NaN compares false: flagged
        bad = b if bad is None else (bad | b)
    return (bad.to(torch.int32) * 2).reshape(1)


def f16x2_image(t: torch.Tensor) -> torch.Tensor:
    """fp32 tensor (numel % 4 == 0) -> same shape, every group of four values replaced by its four `hi` and four `lo'` fp16
    halves (x = hi + 2^-11 lo', csrc/rr_common.h) — what csrc/rr_decode.hip:k_pack_f16x2 does on the device; applied to a pack_a
    fragment array it yields the [hi | lo'] weight fragments of csrc/rr_gemm_f16.h."""
    x = t.detach().float().contiguous()
    g = x.view(-1, 4)
    hi = g.to(torch.float16)
    lo = ((g - hi.float()) * 2048.0).to(torch.float16)
    return torch.cat((hi, lo), dim=1).contiguous().view(torch.float32).view(x.shape)


def pack_bf16x2(Wm: torch.Tensor, k_major: bool = False) -> torch.Tensor:
    """[M,K] fp32 -> two-piece bf16 split (hi = bf16(W), lo = bf16(W - hi)) as A operands of v_mfma_f32_16x16x32_bf16 in the
    permuted k order of pack_a_bf16x3 (a lane's eight values = its four of two consecutive C-layout tiles): [M/16][K/32][2][64][8],
    or k-slice-major [K/32][M/16][2][64][8] (the fragments one hidden pair of csrc/rr_train_dec.hip:k_mlp_rows needs are then
    contiguous)."""
    *lead, M, K = Wm.shape                                   # leading dims: a batch of matrices
    assert M % 16 == 0 and K % 32 == 0
    n = len(lead)
    W = Wm.detach().float()
    hi = W.to(torch.bfloat16)
    lo = (W - hi.float()).to(torch.bfloat16)
    x = torch.stack((hi, lo), dim=n).reshape(*lead, 2, M // 16, 16, K // 32, 2, 4, 4)           # piece, t, i, s, half, g, e4
    if k_major:
        x = x.permute(*range(n), n + 3, n + 1, n, n + 5, n + 2, n + 4, n + 6)                   # s, t, piece, g, i, half, e4
        return x.reshape(*lead, K // 32, M // 16, 2, 64, 8).contiguous()
    x = x.permute(*range(n), n + 1, n + 3, n, n + 5, n + 2, n + 4, n + 6)                       # t, s, piece, g, i, half, e4
    return x.reshape(*lead, M // 16, K // 32, 2, 64, 8).contiguous()


def pack_bf16x2_nat(Wm: torch.Tensor) -> torch.Tensor:
    """The same split with the NATURAL k order (lane (i = l&15, g = l>>4), element e <-> W[16t+i][32s + 8g + e]):
    [M/16][K/32][2][64][8] — the B operand of products whose A operand is read row-wise from memory (k_mlp_wgrad)."""
    *lead, M, K = Wm.shape
    assert M % 16 == 0 and K % 32 == 0
    n = len(lead)
    W = Wm.detach().float()
    hi = W.to(torch.bfloat16)
    lo = (W - hi.float()).to(torch.bfloat16)
    x = torch.stack((hi, lo), dim=n).reshape(*lead, 2, M // 16, 16, K // 32, 4, 8)              # piece, t, i, s, g, e
    x = x.permute(*range(n), n + 1, n + 3, n, n + 4, n + 2, n + 5)                               # t, s, piece, g, i, e
    return x.reshape(*lead, M // 16, K // 32, 2, 64, 8).contiguous()


def pack_mlp_train_batched(W1s, b1s, W2s, b2s) -> list:
    """Operand packs of the training-side 128 -> 512 -> 128 MLP kernels (pointer MLP decoder.py:272-277, TransformerFFN
    attn_freenet.py:330-357) for a LIST of MLPs at once (the packs of all of them cost the launches of one): forward, input
    gradient, weight gradient.  Built on the weights' device.  -> one dict of kernel descriptors per MLP."""
    W1 = torch.stack([w.detach().float() for w in W1s])              # [n,512,128]
    W2 = torch.stack([w.detach().float() for w in W2s])              # [n,128,512]
    W1t, W2t = W1.transpose(1, 2).contiguous(), W2.transpose(1, 2).contiguous()
    keep = {"wa1": pack_bf16x2(W1), "wa2": pack_bf16x2(W2t), "wb_fwd": pack_bf16x2(W2, k_major=True),
            "wb_bwd": pack_bf16x2(W1t, k_major=True), "w1n": pack_bf16x2_nat(W1), "w2tn": pack_bf16x2_nat(W2t),
            "b1": [b.detach().float().contiguous() for b in b1s], "b2": [b.detach().float().contiguous() for b in b2s]}
    out = []
    for i in range(W1.shape[0]):
        fw, bw, wg = L.MlpRowsW(), L.MlpRowsW(), L.MlpWgradW()
        b1p, b2p = keep["b1"][i].data_ptr(), keep["b2"][i].data_ptr()
        fw.wa1, fw.wa2, fw.wb, fw.b1, fw.b2 = keep["wa1"][i].data_ptr(), None, keep["wb_fwd"][i].data_ptr(), b1p, b2p
        bw.wa1, bw.wa2, bw.wb, bw.b1, bw.b2 = keep["wa1"][i].data_ptr(), keep["wa2"][i].data_ptr(), keep["wb_bwd"][i].data_ptr(), b1p, None
        wg.w1n, wg.w2tn, wg.b1 = keep["w1n"][i].data_ptr(), keep["w2tn"][i].data_ptr(), b1p
        out.append({"keep": keep, "fwd": fw, "bwd": bw, "wgrad": wg})
    return out


def pack_mlp_train(W1: torch.Tensor, b1: torch.Tensor, W2: torch.Tensor, b2: torch.Tensor) -> dict:
    return pack_mlp_train_batched([W1], [b1], [W2], [b2])[0]


@contextlib.contextmanager
def _few_threads():
    """The folds are a few hundred small float64 host ops.  On a many-core host (128 intra-op threads on the MI355X boxes)
    the per-op fork/join of the BLAS / intra-op pools costs 15x the arithmetic (397 -> 26 ms per repack of the RCVRPTW
    policy), and the repack runs after every optimizer step: run them single-threaded."""
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        try:
            from threadpoolctl import threadpool_limits
        except ImportError:          # optional: only numpy's BLAS pool is left unlimited without it
            yield
        else:
            with threadpool_limits(limits=1):
                yield
    finally:
        torch.set_num_threads(n)


def fold_nab(sd, p: str, alpha: torch.Tensor) -> torch.Tensor:
    """DistAngleFusion without duration (attn_freenet.py:242-289) -> [8*E + 8] coefficients.
    wo.(W2 h + b2) = (W2^T wo).h + wo.b2, same for the two gate halves; folded in float64."""
    d = lambda k: sd[p + k].detach().double().cpu()  # noqa: E731
    wo, bo = d(".out_lin.weight")[0], d(".out_lin.bias")[0]
    wg, bg = d(".gate.0.weight")[0], d(".gate.0.bias")[0]
    rows, ks = [], []
    for nm, wgh in (("dist_emb", wg[:E]), ("angle_emb", wg[E:])):
        W2, b2 = d(f".{nm}.2.weight"), d(f".{nm}.2.bias")
        rows += [d(f".{nm}.0.weight")[:, 0], d(f".{nm}.0.bias"), W2.t() @ wo, W2.t() @ wgh]
        ks += [wo @ b2, wgh @ b2]
    tail = torch.stack(ks + [bg, bo, alpha.detach().double().cpu().reshape(()), torch.zeros((), dtype=torch.float64)])
    return torch.cat([torch.cat(rows), tail]).float()


def fold_nab_pwl(sd, p: str, alpha: torch.Tensor) -> torch.Tensor:
    """Exact piecewise-linear form of the folded gating NAB (see fold_nab).  Each of the four scalar functions
        f(x) = sum_k c_k relu(a_k x + b_k) + const       (x = distance or angle; c = W2^T wo or W2^T wg)
    is piecewise linear in x with breakpoints t_k = -b_k / a_k.  Per sorted segment m we store the slope S_m and the
    function value F_m at the segment's anchor breakpoint (both accumulated in float64), and the kernel evaluates
    f(x) = F_m + S_m (x - t_anchor) after a binary search: ~45 VALU ops per edge instead of 1 024, with no cancellation
    (x - t_anchor is small).  Layout: t_d[128] | t_a[128] | seg_d[129][4] | seg_a[129][4] | (bg, bo, alpha, 0...)[8],
    seg = (S_out, F_out, S_gate, F_gate); unused breakpoints are +inf."""
    import numpy as np
    d = lambda k: sd[p + k].detach().double().cpu().numpy()  # noqa: E731
    wo, bo = d(".out_lin.weight")[0], float(d(".out_lin.bias")[0])
    wg, bg = d(".gate.0.weight")[0], float(d(".gate.0.bias")[0])
    ts, segs = [], []
    for nm, wgh in (("dist_emb", wg[:E]), ("angle_emb", wg[E:])):
        a, b = d(f".{nm}.0.weight")[:, 0], d(f".{nm}.0.bias")
        W2, b2 = d(f".{nm}.2.weight"), d(f".{nm}.2.bias")
        co, cg, ko, kg = W2.T @ wo, W2.T @ wgh, float(wo @ b2), float(wgh @ b2)
        nz = a != 0
        t = np.sort(-b[nz] / a[nz])
        M = len(t)
        # all M + 1 segments at once (this runs after every optimizer step): probe point inside the segment -> active units ->
        # slope; anchor breakpoint -> function value.  Same float64 sums as a per-segment loop.
        seg = np.zeros((129, 4))
        if M == 0:
            xm, anchor = np.zeros(1), np.zeros(1)
        else:
            xm = np.concatenate(([t[0] - 1.0], 0.5 * (t[:-1] + t[1:]), [t[M - 1] + 1.0]))
            anchor = np.concatenate(([t[0]], t))                                   # segment m > 0 is anchored at t[m-1]
        act = (a[None, :] * xm[:, None] + b[None, :]) > 0                          # [M+1, 128]
        h = np.maximum(a[None, :] * anchor[:, None] + b[None, :], 0.0)
        seg[:M + 1, 0] = (act * (co * a)[None, :]).sum(1)
        seg[:M + 1, 1] = (h * co[None, :]).sum(1) + ko
        seg[:M + 1, 2] = (act * (cg * a)[None, :]).sum(1)
        seg[:M + 1, 3] = (h * cg[None, :]).sum(1) + kg
        seg[M + 1:] = seg[M]
        tt = np.full(128, np.inf)
        tt[:M] = t
        ts.append(tt); segs.append(seg.reshape(-1))
    tail = np.zeros(8)
    tail[0], tail[1], tail[2] = bg, bo, float(alpha.detach().double().cpu().reshape(()))
    tab = np.concatenate(ts + segs + [tail]).astype(np.float32)
    return torch.from_numpy(np.concatenate([tab, nab_grid_cells(tab)]))


def fold_nab_pwl_batched(sd, prefixes, alphas) -> torch.Tensor:
    """fold_nab_pwl + nab_grid_cells for several blocks at once, in torch float64 ON THE WEIGHTS' DEVICE: the training step
    repacks after every optimizer step, and the numpy fold costs a dozen blocking device-to-host copies per block (the host
    then cannot run ahead of the GPU).  Same tables (the float64 sums may differ from numpy's in the last bit before the
    float32 cast).  -> [len(prefixes), NAB_TAB2_FLOATS] float32."""
    dev = sd[prefixes[0] + ".out_lin.weight"].device
    nb = len(prefixes)
    st = lambda k: torch.stack([sd[p + k].detach() for p in prefixes]).double()           # noqa: E731  (one conversion per stack, not per block)
    wo, bo = st(".out_lin.weight")[:, 0], st(".out_lin.bias")[:, 0]                        # [nb,E], [nb]
    wg, bg = st(".gate.0.weight")[:, 0], st(".gate.0.bias")[:, 0]                          # [nb,2E], [nb]
    ts, segs, cells = [], [], []
    m_idx = torch.arange(129, device=dev)
    for f, nm in enumerate(("dist_emb", "angle_emb")):
        wgh = wg[:, f * E:(f + 1) * E]
        a, b = st(f".{nm}.0.weight")[:, :, 0], st(f".{nm}.0.bias")                         # [nb,E]
        W2, b2 = st(f".{nm}.2.weight"), st(f".{nm}.2.bias")                                # [nb,E,E], [nb,E]
        co, cg = (W2 * wo[:, :, None]).sum(1), (W2 * wgh[:, :, None]).sum(1)               # W2^T wo, W2^T wg (elementwise: no BLAS in a repack)
        ko, kg = (wo * b2).sum(1), (wgh * b2).sum(1)
        nz = a != 0
        t = torch.where(nz, -b / torch.where(nz, a, torch.ones_like(a)), torch.full_like(a, float("inf"))).sort(dim=1).values
        M = nz.sum(1)                                                                      # breakpoints per block
        Mi = M.clamp(min=1)
        g = lambda idx: t.gather(1, idx.clamp(0, E - 1))                                   # noqa: E731
        mm = torch.minimum(m_idx[None].expand(nb, -1), M[:, None])                         # segments beyond M repeat segment M
        lo_, hi_ = g(mm - 1), g(torch.minimum(mm, (Mi - 1)[:, None]))
        first, last = g(torch.zeros_like(mm)), g((Mi - 1)[:, None].expand(-1, 129))
        xm = torch.where(mm == 0, first - 1.0, torch.where(mm == M[:, None], last + 1.0, 0.5 * (lo_ + hi_)))
        anchor = torch.where(mm == 0, first, lo_)
        none = (M == 0)[:, None]
        xm, anchor = torch.where(none, torch.zeros_like(xm), xm), torch.where(none, torch.zeros_like(anchor), anchor)
        act = (a[:, None, :] * xm[:, :, None] + b[:, None, :]) > 0                         # [nb,129,E]
        h = torch.clamp(a[:, None, :] * anchor[:, :, None] + b[:, None, :], min=0.0)
        seg = torch.stack([(act * (co * a)[:, None, :]).sum(2), (h * co[:, None, :]).sum(2) + ko[:, None],
                           (act * (cg * a)[:, None, :]).sum(2), (h * cg[:, None, :]).sum(2) + kg[:, None]], dim=2)      # [nb,129,4]
        ts.append(t); segs.append(seg.reshape(nb, -1))
        # grid-start bounds (nab_grid_cells) on the float32 breakpoints the kernel sees
        t32 = t.float().double()
        lo, hi = NAB_RANGES[f]
        wdt = (hi - lo) / NAB_G
        ar = torch.arange(NAB_G, device=dev, dtype=torch.float64)
        edges = lo + wdt * ar - 1e-2 * wdt - 1e-6
        uppers = lo + wdt * (ar + 1) + 1e-2 * wdt + 1e-6
        t32c = t32.contiguous()
        start = torch.searchsorted(t32c, edges[None].expand(nb, -1).contiguous(), right=True)
        end = torch.searchsorted(t32c, uppers[None].expand(nb, -1).contiguous(), right=True)
        need = (end != start) | (start >= 128)                                             # bit 7 (nab_grid_cells): the scan is needed
        need[:, 0] = True; need[:, -1] = True
        cells.append((start.clamp(max=127) | (need.long() << 7)).to(torch.uint8))
    alpha = torch.stack([x.detach().double().reshape(()) for x in alphas])
    tail = torch.zeros(nb, 8, dtype=torch.float64, device=dev)
    tail[:, 0], tail[:, 1], tail[:, 2] = bg, bo, alpha
    tab = torch.cat(ts + segs + [tail], dim=1).float()
    cellw = torch.cat(cells, dim=1).contiguous().view(torch.float32)                       # 4 bounds per word
    return torch.cat([tab, cellw], dim=1).contiguous()


NAB_G = 1024            # csrc/rr_encoder.hip: NAB_G
NAB_RANGES = ((0.0, 1.0), (-math.pi, math.pi))   # min-max-normalised distance, atan2 angle


def nab_grid_cells(tab):
    """Start-of-scan tables for csrc/rr_encoder.hip:nab_edge4_grid: per family, for each of the NAB_G uniform cells of
    the input range, a lower bound (7 bits) of "number of float32 breakpoints <= x" valid for every x the kernel maps to
    that cell, and in bit 7 whether a scan from that bound is needed at all.  The kernel computes the cell in float32, so the bound is taken a little below the cell's lower edge.
    Inputs below the range (negative raw cell index) start their scan at 0 in the kernel; inputs above it land in the
    last cell and scan on.  Returned packed 4 bytes per float32."""
    import numpy as np
    out = []
    for f, (lo, hi) in enumerate(NAB_RANGES):
        t = tab[128 * f:128 * (f + 1)].astype(np.float64)
        t = t[np.isfinite(t)]
        w = (hi - lo) / NAB_G
        edges = lo + w * np.arange(NAB_G) - 1e-2 * w - 1e-6
        start = np.searchsorted(t, edges, side="right")          # breakpoints <= (edge - margin)
        # bit 7: the scan is needed — a breakpoint may lie among the inputs of this cell (between its lower edge minus the margin
        # and its upper edge plus the margin), or the cell also receives out-of-range inputs (first / last); without it the bound
        # IS the segment and the kernel reads no breakpoint at all.  The bound itself in 7 bits (128 -> 127 + scan).
        end = np.searchsorted(t, lo + w * (np.arange(NAB_G) + 1) + 1e-2 * w + 1e-6, side="right")
        need = (end != start) | (start >= 128)
        need[0] = need[-1] = True
        out.append((np.minimum(start, 127) | (need.astype(np.int64) << 7)).astype(np.uint8))
    return np.concatenate(out).view(np.float32)


def eval_nab_pwl(tab: torch.Tensor, dmat: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    """float32 emulation of csrc/rr_encoder.hip:nab_edge_pwl (used by the CPU tests to validate the tables)."""
    def fam(x, t, seg):
        seg = seg.view(129, 4)
        m = torch.searchsorted(t.contiguous(), x.contiguous(), right=True)          # number of breakpoints <= x
        anchor = t[(m - 1).clamp_min(0)]
        anchor = torch.where(torch.isfinite(anchor), anchor, torch.zeros_like(anchor))
        s = seg[m]
        dx = x - anchor
        return s[..., 1] + s[..., 0] * dx, s[..., 3] + s[..., 2] * dx
    od, gd = fam(dmat, tab[0:128], tab[256:256 + 516])
    oa, ga = fam(theta, tab[128:256], tab[256 + 516:256 + 1032])
    bg, bo, al = tab[1288], tab[1289], tab[1290]
    g = torch.sigmoid(gd + ga + bg)
    return (g * od + (1 - g) * oa + bo) * al


def fold_nab_dur(sd, p: str, alpha: torch.Tensor, ar) -> "L.NabDurW":
    """DistAngleFusion with duration (attn_freenet.py:226-237, 265-286): fold every second MLP layer into the gate's
    first Linear (M_x = Wg0_x W2_x) and into out_lin (co_x = W2_x^T wo), in float64; see csrc/rr_encoder.hip:k_nab_dur."""
    names = [".out_lin.weight", ".out_lin.bias", ".gate.0.weight", ".gate.0.bias", ".gate.2.weight", ".gate.2.bias",
             ".gate_temperature"] + [f".{nm}.{l}.{wb}" for nm in ("dist_emb", "angle_emb", "dur_emb") for l in (0, 2)
                                     for wb in ("weight", "bias")]
    # ONE device-to-host copy for the module (this runs after every optimizer step; per-tensor .cpu() calls each synchronise)
    flat = torch.cat([sd[p + k].detach().reshape(-1).double() for k in names]).cpu()
    host, off = {}, 0
    for k in names:
        n = sd[p + k].numel()
        host[k] = flat[off:off + n].view(sd[p + k].shape)
        off += n
    d = host.__getitem__
    wo, bo = d(".out_lin.weight")[0], d(".out_lin.bias")[0]
    Wg0, bg0 = d(".gate.0.weight"), d(".gate.0.bias")
    Ms, cos, kos, a_, b_ = [], [], [], [], []
    cg = bg0.clone()
    for i, nm in enumerate(("dist_emb", "angle_emb", "dur_emb")):
        W2, b2 = d(f".{nm}.2.weight"), d(f".{nm}.2.bias")
        Wgx = Wg0[:, i * E:(i + 1) * E]
        Ms.append(Wgx @ W2)
        cg += Wgx @ b2
        cos.append(W2.t() @ wo)
        kos.append(float(wo @ b2))
        a_.append(d(f".{nm}.0.weight")[:, 0]); b_.append(d(f".{nm}.0.bias"))
    Mcat = torch.zeros(9 * 16, 3 * E, dtype=torch.float64)
    Mcat[:E] = torch.cat(Ms, dim=1)
    for i in range(3):
        Mcat[E + i, i * E:(i + 1) * E] = cos[i]
    w = L.NabDurW()
    w.mp = ar.put(pack_a(Mcat.float()))
    w.ab = ar.put(torch.cat(a_ + b_).float())
    w.cg = ar.put(cg.float())
    w.wg2 = ar.put(d(".gate.2.weight").float().contiguous())
    bg2 = d(".gate.2.bias")
    for i in range(3):
        w.bg2[i] = float(bg2[i]); w.ko[i] = kos[i]
    w.inv_tau = float(torch.exp(-d(".gate_temperature")))
    w.bo = float(bo)
    w.alpha = float(alpha.detach().double().cpu().reshape(()))
    w.pwl = ar.put(fold_nab_dur_pwl([m.numpy() for m in Ms], cg.numpy(), [c.numpy() for c in cos], kos,
                                    [x.numpy() for x in a_], [x.numpy() for x in b_]))
    return w


NABD_TS, NABD_SEG = 132, 129     # csrc/rr_encoder.hip: sentinel-terminated breakpoint list / segments per family
NABD_RANGES = ((0.0, 1.0), (-math.pi, math.pi), (0.0, 1.0))   # distance, angle, duration (both matrices min-max normalised)


def fold_nab_dur_pwl(Ms, cg, cos, kos, a_, b_) -> torch.Tensor:
    """Vector-valued piecewise-linear form of the gate pre-activation of DistAngleFusion with duration.

    z = cg + sum_x M_x relu(a_x x + b_x) is, for each input family x in (d, theta, t), a function R -> R^128 that is linear
    between the 128 breakpoints -b_k / a_k of that family.  Per family and segment m we store the value F_m at the
    segment's anchor breakpoint and the slope S_m (float64 accumulation); the kernel (k_nab_dur_pwl) evaluates
    z = sum_x F_x[m_x] + S_x[m_x] (x - anchor) with six 512-byte row reads per edge instead of a 128 x 384 contraction:
    110 kflop of MFMA work per edge become ~0.8 kflop of fma.  The three out_lin projections co_x . h_x are scalar PWL
    functions of the same segments.  Layout (float32 words):
      ts      [3][132]   sorted breakpoints, +inf padded (entry 128.. = +inf: scan sentinel)
      anchor  [3][132]   anchor of segment m (0 where the family has no breakpoint to anchor on)
      osc     [3][129][2] (+2 pad)  (F_o, S_o) of co_x . h_x + ko_x
      cells   [3][1024] u8 grid-start bounds (nab_grid_cells semantics), packed 4 per word
      rows    [3][129][2][128]  F then S of the 128 gate units; the constant cg is folded into family 0's F.
      sliced  [4][3][129][16][2][2] the same rows per 32-unit slice as unit pairs (F_u, F_u+1, S_u, S_u+1) (k_nab_dur_lds copies a slice to LDS)."""
    import numpy as np
    ts = np.full((3, NABD_TS), np.inf)
    anc = np.zeros((3, NABD_TS))
    osc = np.zeros((3 * NABD_SEG * 2 + 2,))          # (F_o, S_o) pairs, padded to a multiple of 4 words
    rows = np.zeros((3, NABD_SEG, 2, E))
    cells = []
    for f in range(3):
        a, b, M, co = a_[f].astype(np.float64), b_[f].astype(np.float64), Ms[f].astype(np.float64), cos[f].astype(np.float64)
        nz = a != 0
        t = np.sort(-b[nz] / a[nz])
        nb = len(t)
        ts[f, :nb] = t
        # all 129 segments at once (this runs after every optimizer step).  Segment m: probe point inside it -> active units ->
        # slope; anchor breakpoint -> value.  Segments beyond the family's nb breakpoints repeat the last one.
        mm = np.minimum(np.arange(NABD_SEG), nb)
        if nb == 0:
            xm, an = np.zeros(NABD_SEG), np.zeros(NABD_SEG)
        else:
            lo_ = t[np.clip(mm - 1, 0, nb - 1)]
            hi_ = t[np.clip(mm, 0, nb - 1)]
            xm = np.where(mm == 0, t[0] - 1.0, np.where(mm == nb, t[nb - 1] + 1.0, 0.5 * (lo_ + hi_)))
            an = np.where(mm == 0, t[0], lo_)
        act = (a[None, :] * xm[:, None] + b[None, :]) > 0                          # [129, 128]
        h = np.maximum(a[None, :] * an[:, None] + b[None, :], 0.0)
        rows[f, :, 0] = h @ M.T + (cg if f == 0 else 0.0)
        rows[f, :, 1] = (act * a[None, :]) @ M.T
        oscv = osc[:3 * NABD_SEG * 2].reshape(3, NABD_SEG, 2)                      # a view: (F_o, S_o) pairs
        oscv[f, :, 0] = h @ co + kos[f]
        oscv[f, :, 1] = (act * (co * a)[None, :]).sum(1)
        anc[f, :NABD_SEG] = an
        # grid start bounds, same construction as nab_grid_cells (float32 breakpoints as the kernel sees them)
        t32 = ts[f, :128].astype(np.float32).astype(np.float64)
        t32 = t32[np.isfinite(t32)]
        lo, hi = NABD_RANGES[f]
        wdt = (hi - lo) / NAB_G
        edges = lo + wdt * np.arange(NAB_G) - 1e-2 * wdt - 1e-6
        cells.append(np.minimum(np.searchsorted(t32, edges, side="right"), 128).astype(np.uint8))
    head = np.concatenate([ts.reshape(-1), anc.reshape(-1), osc.reshape(-1)]).astype(np.float32)
    cellw = np.concatenate(cells).view(np.float32)
    # the same rows again as [4 unit slices][3][129][16 unit pairs](F, F, S, S) for the LDS-resident kernel (k_nab_dur_lds)
    sliced = rows.reshape(3, NABD_SEG, 2, 4, 16, 2).transpose(3, 0, 1, 4, 2, 5)
    return torch.from_numpy(np.concatenate([head, cellw, rows.reshape(-1).astype(np.float32),
                                            np.ascontiguousarray(sliced).reshape(-1).astype(np.float32)]))


def fold_init_gate(W0, b0, Wd, bd, Wn, bn, Wdep=None, bdep=None):
    """The gate's first layer (gating_fc.0, Linear(2E,2E): atsp.py:108-121 / rcvrp.py:88-103) folded through the two embeddings it
    reads — both are linear in their inputs, so
        W0 [node_emb | dist_emb] + b0 = (W0[:, :E] Wn) feat + (W0[:, E:] Wd) sorted + (W0[:, :E] bn + W0[:, E:] bd + b0).
    -> gf [32, 2E] = (W0[:, E:] Wd)^T zero-padded behind the SS samples; gn [2E, 4] = (the <= 3 coefficients of the node features,
    the constant); gd = gn for the VRP depot's own Linear(2,E) (None for ATSP).  Products in float64 (small_gemm), results fp32:
    csrc/rr_encoder.hip:k_init_embed<., ., true> runs the layer as K = 32 on the fp16 matrix pipe + 3 fmas per hidden unit."""
    E2 = W0.shape[0]
    W0 = W0.detach().double()
    W0a, W0b = W0[:, :E].contiguous(), W0[:, E:].contiguous()
    SS = Wd.shape[1]
    assert SS <= 32 and W0.shape == (E2, 2 * E)
    F = small_gemm(W0b, Wd.detach().double().contiguous())                        # [2E, SS]
    gf = torch.zeros(32, E2, dtype=torch.float64, device=W0.device)
    gf[:SS] = F.t()
    c = small_gemm(W0b, bd.detach().double().reshape(-1, 1).contiguous())[:, 0] + b0.detach().double()

    def node(Wx, bx):
        A = small_gemm(W0a, Wx.detach().double().contiguous())                    # [2E, 2 | 3]
        g = torch.zeros(E2, 4, dtype=torch.float64, device=W0.device)
        g[:, :A.shape[1]] = A
        g[:, 3] = c + small_gemm(W0a, bx.detach().double().reshape(-1, 1).contiguous())[:, 0]
        return g.float()
    return gf.float(), node(Wn, bn), (node(Wdep, bdep) if Wdep is not None else None)


class _Arena:
    """Keeps every packed tensor alive and hands out raw device pointers."""

    def __init__(self, device):
        self.device = device
        self.keep = []

    def put(self, t: torch.Tensor):
        t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        self.keep.append(t)
        return t.data_ptr()

    def put_raw(self, t: torch.Tensor):
        t = t.detach().to(device=self.device).contiguous()
        self.keep.append(t)
        return t.data_ptr()


def pack_policy(sd: dict, env_name: str, device) -> dict:
    with _few_threads(), torch.no_grad():          # (sd may hold the Parameters themselves: no graph is wanted here)
        return _pack_policy(sd, env_name, device)


def _pack_policy(sd: dict, env_name: str, device) -> dict:
    """state_dict (reference names) -> ctypes structs for the kernels."""
    ar = _Arena(device)
    split = mlp_split_enabled()          # the bf16 split copies are only built when the opt-in switch is on at pack time
    nl = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("encoder.net.layers."))
    blocks, nabdur, nabsimple, pwl_todo = [], {}, {}, []
    nabname = "angle_distance_fusion" if env_name in ("atsp", "rcvrp") else "neural_adaptive_bias"
    q0 = "encoder.net.layers.0.row_encoding_block.neural_adaptive_bias"
    nab_kind = "naive" if (q0 + ".mlp.0.weight") in sd else "heuristic" if (q0 + ".alpha") in sd else "gating"
    # The weight matrices of all 2 * nl blocks are packed together, one short launch sequence per shape instead of one per
    # matrix: the training step repacks after every optimizer step and was bound by these thousands of small launches.
    names = [f"encoder.net.layers.{l}.{rc}_encoding_block" for l in range(nl) for rc in ("row", "col")]
    dv = lambda t: t.detach().to(device)                                                    # noqa: E731
    stk = lambda k: torch.stack([dv(sd[f"{b}.{k}"]).float() for b in names])               # noqa: E731
    sq = pack_a(torch.stack([stk("attn_free.to_q.weight"), stk("attn_free.to_k.weight"), stk("attn_free.to_v.weight")]))    # [3][nb]...
    W1s, W2s = stk("feed_forward.ops.ffn.W1.weight"), stk("feed_forward.ops.ffn.W2.weight")
    p1, p2 = pack_a(W1s), pack_a(W2s)
    # AFTFull.project (attn_freenet.py:325) feeds multi_head_combine (:435) directly: one Linear, folded in float64
    Wc64, Wp64 = stk("multi_head_combine.weight").double(), stk("attn_free.project.weight").double()
    Wpc32 = small_gemm(Wc64, Wp64).float()
    ppc = pack_a(Wpc32)
    bpc = ((Wc64 * stk("attn_free.project.bias").double()[:, None, :]).sum(2) + stk("multi_head_combine.bias").double()).float().contiguous()
    # mean over the nodes of K = to_k(norm2(y)): norm2's output has mean n2.beta per feature whatever the instance (InstanceNorm1d; the
    # running-statistics / RMS / layer forms never take the kernels that read this), so mean(K) = Wk n2.beta + bk — the shift of the
    # node softmax in csrc/rr_enc_split.inc (any shift gives the same softmax; this one needs no reduction)
    n2b = torch.stack([dv(sd[f"{b}.norm2.normalizer.bias"]).double() if f"{b}.norm2.normalizer.bias" in sd else torch.zeros(E, dtype=torch.float64, device=device)
                       for b in names])
    muk = ((stk("attn_free.to_k.weight").double() * n2b[:, None, :]).sum(2) + stk("attn_free.to_k.bias").double()).float().contiguous()
    ar.keep += [sq, p1, p2, ppc, bpc, muk]
    if split:      # FFN weights again as 3-way bf16 splits for the bf16-pipe FFN
        p1s, p2s = pack_a_f16u(W1s), pack_a_f16u(W2s)          # second-form images (x 2^6, csrc/rr_common.h): encoder FFN
        sqs = pack_a_f16x2(torch.stack([stk("attn_free.to_q.weight"), stk("attn_free.to_k.weight"), stk("attn_free.to_v.weight"),
                                        Wpc32]))                                               # [4][nb][8][4][2][64][8]
        ar.keep += [p1s, p2s, sqs]
    for bi, b in enumerate(names):
        l = bi // 2
        if bi % 2 == 0:
            pair = []
        if True:
            w = L.EncBlockW()
            for f, k in (("n1", "norm1"), ("n2", "norm2"), ("n3", "norm3"),
                         ("f1", "feed_forward.ops.norm1"), ("f2", "feed_forward.ops.norm2")):
                kg, kb = f"{b}.{k}.normalizer.weight", f"{b}.{k}.normalizer.bias"
                if f"{b}.{k}.normalizer.running_mean" in sd:      # BatchNorm1d (eval): fold the running statistics (float64)
                    gam, bet = sd[kg].detach().double(), sd[kb].detach().double()
                    rm, rv = sd[f"{b}.{k}.normalizer.running_mean"].double(), sd[f"{b}.{k}.normalizer.running_var"].double()
                    gam = gam / torch.sqrt(rv + 1e-5)
                    bet = bet - rm * gam
                    setattr(w, f + "g", ar.put(gam.float())); setattr(w, f + "b", ar.put(bet.float()))
                else:
                    # "layer" has no parameters, RMSNorm a weight only (attn_freenet.py:13-26, 92-93): identity where absent;
                    # an existing float32 parameter on the device is used in place (no copy)
                    setattr(w, f + "g", ar.put(sd[kg] if kg in sd else torch.ones(E)))
                    setattr(w, f + "b", ar.put(sd[kb] if kb in sd else torch.zeros(E)))
            w.wq, w.wk, w.wv = sq[0, bi].data_ptr(), sq[1, bi].data_ptr(), sq[2, bi].data_ptr()
            w.w1, w.w2 = p1[bi].data_ptr(), p2[bi].data_ptr()
            for f, k in (("q", "attn_free.to_q"), ("k", "attn_free.to_k"), ("v", "attn_free.to_v"),
                         ("1", "feed_forward.ops.ffn.W1"), ("2", "feed_forward.ops.ffn.W2")):
                setattr(w, "b" + f, ar.put(sd[f"{b}.{k}.bias"]))
            w.wp, w.bp = ppc[bi].data_ptr(), bpc[bi].data_ptr()
            w.muk = muk[bi].data_ptr()
            w.wc, w.bc = None, None
            if split:
                w.w1s, w.w2s = p1s[bi].data_ptr(), p2s[bi].data_ptr()
                w.wqs, w.wks, w.wvs, w.wps = (sqs[i, bi].data_ptr() for i in range(4))
            if nab_kind != "gating":          # ablation modules: bias computed by rr_nab_simple, fed as bias_pre
                w.nab = None
                q, sw = f"{b}.neural_adaptive_bias", L.NabSimpleW()
                sw.alpha = float(sd[f"{b}.alpha"])
                if nab_kind == "naive":
                    if sd[q + ".mlp.0.weight"].shape[1] != 3:
                        raise NotImplementedError("NaiveNeuralAdaptiveBias needs the duration matrix (the reference raises "
                                                  "without it, attn_freenet.py:193-195): rcvrptw only")
                    sw.w0, sw.b0 = ar.put(sd[q + ".mlp.0.weight"]), ar.put(sd[q + ".mlp.0.bias"])
                    sw.w2, sw.b2 = ar.put(sd[q + ".mlp.2.weight"].reshape(-1)), float(sd[q + ".mlp.2.bias"])
                else:
                    sw.dw = float(sd[q + ".distance_weight"]) if (q + ".distance_weight") in sd else 1.0
                    sw.tw = float(sd[q + ".duration_weight"]) if (q + ".duration_weight") in sd else 0.0
                nabsimple.setdefault(l, []).append(sw)
            elif nabname == "angle_distance_fusion":
                w.nab = None
                pwl_todo.append((w, f"{b}.{nabname}", sd[f"{b}.alpha"]))
            else:
                w.nab = None
                nabdur.setdefault(l, []).append(fold_nab_dur(sd, f"{b}.{nabname}", sd[f"{b}.alpha"], ar))
            pair.append(w)
        if bi % 2 == 1:
            blocks.append(tuple(pair))
    if pwl_todo:      # the gating NABs of all blocks in one batched fold on the weights' device (no host round trips)
        tabs = fold_nab_pwl_batched({k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in sd.items()
                                     if ".angle_distance_fusion." in k}, [p for _, p, _ in pwl_todo], [a.to(device) for _, _, a in pwl_todo])
        ar.keep.append(tabs)
        for i, (w, _, _) in enumerate(pwl_todo):
            w.nab = tabs[i].data_ptr()

    out = {"arena": ar, "sd_ref": sd, "blocks": blocks, "num_layers": nl, "nabdur": [tuple(nabdur[l]) for l in sorted(nabdur)],
           "nab_kind": nab_kind, "nabsimple": [tuple(nabsimple[l]) for l in sorted(nabsimple)]}
    p = "encoder.init_embedding"
    gate_w = []            # weights with fp16 images beside the pointer MLP's: they join the range guard's bit 1
    if env_name == "atsp":
        iw = L.InitW()
        # ATSPInitEmbedding's non-default branches (atsp.py:29-35): without use_coords there is no init_embed, without use_coords AND
        # use_dist there are no gates; out["init_mode"]: 0 the published configuration, 1 coordinates only, 2 distances only (unsorted)
        has_coords, has_gate = (p + ".init_embed.weight") in sd, (p + ".gating_network_row.gating_fc.0.weight") in sd
        out["init_mode"] = 0 if (has_coords and has_gate) else (1 if has_coords else 2)
        if has_coords:
            iw.wi, iw.bi = ar.put(sd[p + ".init_embed.weight"]), ar.put(sd[p + ".init_embed.bias"])
        if (p + ".row_embed.bias") not in sd:
            raise NotImplementedError("ATSPInitEmbedding without linear_bias")
        iw.wr, iw.br = ar.put(sd[p + ".row_embed.weight"].t().contiguous()), ar.put(sd[p + ".row_embed.bias"])   # [SS,E]: coalesced over features
        iw.wcl, iw.bcl = ar.put(sd[p + ".col_embed.weight"].t().contiguous()), ar.put(sd[p + ".col_embed.bias"])
        for rc, s in ((("row", "r"), ("col", "c")) if has_gate else ()):
            q = f"{p}.gating_network_{rc}.gating_fc"
            setattr(iw, "g0" + s, ar.put(pack_a(sd[q + ".0.weight"].detach().float())))
            if split and has_coords:       # the gate's first layer folded through both embeddings, on the fp16 pipe (fold_init_gate)
                gf, gn, _ = fold_init_gate(dv(sd[q + ".0.weight"]), dv(sd[q + ".0.bias"]), dv(sd[f"{p}.{rc}_embed.weight"]),
                                           dv(sd[f"{p}.{rc}_embed.bias"]), dv(sd[p + ".init_embed.weight"]), dv(sd[p + ".init_embed.bias"]))
                setattr(iw, "gf" + s, ar.put(gf)); setattr(iw, "gn" + s, ar.put(gn))
                gate_w += [gf, dv(sd[f"{p}.{rc}_embed.weight"])]      # (2^6 x these are split into fp16 pieces inside the kernel)
            setattr(iw, "g0" + s + "b", ar.put(sd[q + ".0.bias"]))
            setattr(iw, "g2" + s, ar.put(sd[q + ".2.weight"].reshape(-1)))
            setattr(iw, "g2" + s + "b", float(sd[q + ".2.bias"].reshape(-1)[0]))
        iw.nfeat = 0
        out["init"] = iw
        out["sample_size"] = sd[p + ".row_embed.weight"].shape[1]
    else:   # RVRPInitEmbedding / RVRPTWInitEmbedding (rcvrp.py:29-46); distance_expert.*_combine_embed are unused (:127-150)
        iw = L.InitW()
        iw.wdep, iw.bdep = ar.put(sd[p + ".coord_expert.init_embed_depot.weight"]), ar.put(sd[p + ".coord_expert.init_embed_depot.bias"])
        iw.wi, iw.bi = ar.put(sd[p + ".coord_expert.init_embed.weight"]), ar.put(sd[p + ".coord_expert.init_embed.bias"])
        iw.wr, iw.br = ar.put(sd[p + ".distance_expert.row_embed.weight"].t().contiguous()), ar.put(sd[p + ".distance_expert.row_embed.bias"])
        iw.wcl, iw.bcl = ar.put(sd[p + ".distance_expert.col_embed.weight"].t().contiguous()), ar.put(sd[p + ".distance_expert.col_embed.bias"])
        dm = ".demand_init" if (p + ".demand_init.weight") in sd else ".init_embed"   # rcvrptw.py:44 names it init_embed
        iw.wdm, iw.bdm = ar.put(sd[p + dm + ".weight"]), ar.put(sd[p + dm + ".bias"])
        iw.nfeat = sd[p + dm + ".weight"].shape[1]
        for rc, s in (("row", "r"), ("col", "c")):
            q = f"{p}.gating_network_{rc}.gating_fc"
            setattr(iw, "g0" + s, ar.put(pack_a(sd[q + ".0.weight"].detach().float())))
            if split:       # the gate's first layer folded through both embeddings, on the fp16 pipe (fold_init_gate)
                ce = p + ".coord_expert"
                gf, gn, gd = fold_init_gate(dv(sd[q + ".0.weight"]), dv(sd[q + ".0.bias"]), dv(sd[f"{p}.distance_expert.{rc}_embed.weight"]),
                                            dv(sd[f"{p}.distance_expert.{rc}_embed.bias"]), dv(sd[ce + ".init_embed.weight"]),
                                            dv(sd[ce + ".init_embed.bias"]), dv(sd[ce + ".init_embed_depot.weight"]), dv(sd[ce + ".init_embed_depot.bias"]))
                setattr(iw, "gf" + s, ar.put(gf)); setattr(iw, "gn" + s, ar.put(gn)); setattr(iw, "gd" + s, ar.put(gd))
                gate_w += [gf, dv(sd[f"{p}.distance_expert.{rc}_embed.weight"])]
            setattr(iw, "g0" + s + "b", ar.put(sd[q + ".0.bias"]))
            setattr(iw, "g2" + s, ar.put(sd[q + ".2.weight"].reshape(-1)))
            setattr(iw, "g2" + s + "b", float(sd[q + ".2.bias"].reshape(-1)[0]))
            setattr(iw, "cm" + s, ar.put(pack_a(sd[f"{p}.combine_{rc}_embed.weight"].detach().float())))
            setattr(iw, "cm" + s + "b", ar.put(sd[f"{p}.combine_{rc}_embed.bias"]))
        out["init"] = iw
        out["sample_size"] = sd[p + ".distance_expert.row_embed.weight"].shape[1]

    wn = sd["decoder.project_node_embeddings.weight"].detach().float()   # [3E,E] -> K,V,L chunks
    wctx = sd["decoder.context_embedding.project_context.weight"].detach().float()
    cw = L.CacheW()
    cw.wk, cw.wv, cw.wl = ar.put(pack_a(wn[:E])), ar.put(pack_a(wn[E:2 * E])), ar.put(pack_a(wn[2 * E:]))
    if split:      # the same five packs as [hi | lo'] fragments: decoder cache on the fp16 pipe (csrc/rr_gemm_f16.h)
        cw.wks, cw.wvs, cw.wls = (ar.put(f16x2_image(pack_a(wn[i * E:(i + 1) * E]))) for i in range(3))
    dw = L.DecW()
    if env_name == "atsp":
        cw.wca, cw.wcb = ar.put(pack_a(wctx[:, :E])), ar.put(pack_a(wctx[:, E:2 * E]))
        if split:
            cw.wcas, cw.wcbs = ar.put(f16x2_image(pack_a(wctx[:, :E]))), ar.put(f16x2_image(pack_a(wctx[:, E:2 * E])))
        ph = sd["decoder.context_embedding.W_placeholder"].detach().float()
        dw.q0 = ar.put((wctx * ph.to(wctx.device)[None, :]).sum(1))
        dw.wstate = None
    else:
        cw.wca, cw.wcb = None, ar.put(pack_a(wctx[:, :E]))
        if split:
            cw.wcbs = ar.put(f16x2_image(pack_a(wctx[:, :E])))
        dw.q0 = None
        dw.wstate = ar.put(wctx[:, E:].t().contiguous())   # [nstate][E]
    dw.w1 = ar.put(pack_a(sd["decoder.pointer.ffn.lins.0.weight"].detach().float()))
    dw.w2 = ar.put(pack_a(sd["decoder.pointer.ffn.lins.1.weight"].detach().float()))
    # the same matrices as 3-way bf16 splits for the opt-in bf16-pipe MLP (RR_MLP_SPLIT=1); kept as raw 16-bit words
    # (always for the decoder: training rollouts use them, RolloutIO.use_split; two small device-side packs)
    w1d, w2d = sd["decoder.pointer.ffn.lins.0.weight"].to(device), sd["decoder.pointer.ffn.lins.1.weight"].to(device)
    dw.w1s = ar.put_raw(pack_a_f16u(w1d))
    dw.w2s = ar.put_raw(pack_a_f16u(w2d))
    dw.b1, dw.b2 = ar.put(sd["decoder.pointer.ffn.lins.0.bias"]), ar.put(sd["decoder.pointer.ffn.lins.1.bias"])
    dw.b1s = ar.put(sd["decoder.pointer.ffn.lins.0.bias"].detach().float() * float(2 ** F16U_WEIGHT_SCALE_LOG2))
    # range guard, bit 1: a weight image that leaves the fp16 range (models/policy.py repeats such a call on the fp32 kernels)
    out["range_status"] = f16_range_status([w1d, w2d] + gate_w, F16U_WEIGHT_SCALE_LOG2)
    dw.alpha = float(sd["decoder.alpha"].reshape(-1)[0])
    dw.beta = float(sd["decoder.beta"].reshape(-1)[0]) if "decoder.beta" in sd else 0.0
    out["cache"], out["dec"] = cw, dw
    return out
