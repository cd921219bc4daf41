"""Hand-written HIP backward of the decoder for the REINFORCE step (BASELINE configs[4]) — host side of csrc/rr_train_dec.hip.

The reference differentiates its sampling forward with autograd (rrnco/models/rl.py:118-128 through
rrnco/models/decoder.py:151-329 and rrnco/models/decoding.py:311-361).  Here the sampling rollout leaves one row per decoder
evaluation (pointer-MLP input g0, output g, mask words, node decided at, node chosen: policy._fused_rollout(dump=...)), and
five kernels turn d loss / d log-likelihood into
  * the gradients of the decoder's own parameters (pointer.ffn.lins.{0,1}, alpha, beta, the state columns of
    project_context), and
  * d loss / d (glimpse keys, glimpse values, logit keys, step-context tables) per instance — the tensors the encoder side
    produces with five Linear maps of the embeddings (decoder.py:214-232 + the context embedding).
Problem independent: ATSP (first-node table), RCVRP (1 state scalar), RCVRPTW / RMTVRP (4 state scalars, duration bias)."""
from __future__ import annotations

import os

import torch

from .. import _lib as L
from .. import packing

E, LDK = 128, 112


def mlp_train_pack(policy):
    """bf16 split operand packs of the pointer MLP, rebuilt when its four tensors change (same rule as policy.packed)."""
    names = ["decoder.pointer.ffn.lins.0.weight", "decoder.pointer.ffn.lins.0.bias",
             "decoder.pointer.ffn.lins.1.weight", "decoder.pointer.ffn.lins.1.bias"]
    P = policy.param_index()["P"]
    ts = [P[n] for n in names]
    # keyed like enc_backward.train_packs: the policy's own pack key (versions, pointers and — while training — the norm
    # fingerprint policy.packed() has already read this step) instead of a second blocking norm read in front of the backward
    pk = getattr(policy, "_pack_cache", None)
    key = tuple((t._version, t.data_ptr()) for t in ts) + ((pk[0],) if pk is not None else ())
    cached = getattr(policy, "_mlp_train_pack", None)
    if cached is None or cached[0] != key:
        with torch.no_grad():
            cached = (key, packing.pack_mlp_train(*[t.detach() for t in ts]))
        policy._mlp_train_pack = cached
    return cached[1]


_WGRAD_WS = {}


def wgrad_workspace(dev) -> torch.Tensor:
    """64 x (dW1 | dW2) partials of rr_mlp_wgrad's row splits (33.5 MB, one per device, reused by every call on the stream)."""
    if os.environ.get("RR_TRAIN_WS", "1") == "0":          # diagnostic: float atomics instead of partials + reduction
        return None
    # one buffer per (device, stream): two backward passes on different HIP streams must not share partials ("cuda" and "cuda:0"
    # are the same device)
    d = torch.device(dev)
    idx = d.index if d.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    if key not in _WGRAD_WS:
        _WGRAD_WS[key] = torch.empty(64 * 2 * 4 * E * E, device=dev)
    return _WGRAD_WS[key]


@torch.no_grad()
def decoder_backward(policy, cache, dump, D, Dur, grad_ll) -> dict:
    """cache: PrecomputedCache of the forward; dump: what the rollout left (policy._fused_rollout); D / Dur: the normalised
    matrices [Bp,N,N] the rollout used; grad_ll [S*Bp] (r = s*Bp + b).  Returns the gradients listed in the module docstring
    and the replayed log-likelihood [S*Bp]."""
    lib = L.lib()
    st = L.stream()
    dev = D.device
    Bp, N, S, Tst = dump["Bp"], dump["N"], dump["S"], dump["T"]
    T = dump.get("T_used", Tst)
    if T <= 0:
        raise RuntimeError("decoder_backward: the rollout ran no decode step")
    seg = Tst * S
    rows = Bp * seg
    K, Lk = cache.glimpse_key.contiguous(), cache.logit_key.contiguous()
    V = cache.glimpse_val_t[:, :, :N].transpose(1, 2).contiguous()
    Kt = torch.zeros(Bp, E, LDK, device=dev); Kt[:, :, :N] = K.transpose(1, 2)
    Lt = torch.zeros(Bp, E, LDK, device=dev); Lt[:, :, :N] = Lk.transpose(1, 2)
    env_name = policy.env_name
    alpha = float(policy.decoder.alpha.detach().reshape(-1)[0])
    beta = float(policy.decoder.beta.detach().reshape(-1)[0]) if (Dur is not None and hasattr(policy.decoder, "beta")) else 0.0
    gll = grad_ll.contiguous().float()

    # ---- logits: log-probabilities, d logits, d g
    dlg = torch.empty(rows, LDK, device=dev)
    dg = torch.empty(rows, E, device=dev)
    logp = torch.zeros(rows, device=dev)
    dscal = torch.zeros(2, device=dev)
    io = L.DecLogitIO()
    io.g, io.meta, io.L, io.Lt, io.D, io.Dur = L.ptr(dump["g"]), L.ptr(dump["meta"]), L.ptr(Lk), L.ptr(Lt), L.ptr(D), L.ptr(Dur)
    io.gll, io.dlg, io.dg, io.logp, io.dscal = L.ptr(gll), L.ptr(dlg), L.ptr(dg), L.ptr(logp), L.ptr(dscal)
    io.Bp, io.N, io.S, io.T, io.seg_stride = Bp, N, S, T, seg
    io.alpha, io.beta, io.tanh_clip, io.temperature = alpha, beta, float(dump["tanh_clip"]), float(dump["temperature"])
    # the split rollout's own fp16 image of L (the replay then repeats its products); a forward on the fp32 kernels left none
    io.Ls = L.ptr(cache.split[2]) if (cache.split is not None and os.environ.get("RR_LOGIT_BWD_F32", "0") != "1") else None
    L.check(lib.rr_dec_logit_bwd(io, st), "rr_dec_logit_bwd")
    # ---- d logit keys: dL[b] = dlg_b^T g_b (rows on the MFMA k axis)
    dL = torch.empty(Bp, N, E, device=dev)
    L.check(lib.rr_gemm_tn(L.ptr(dlg), L.ptr(dump["g"]), L.ptr(dL), Bp, T * S, N, LDK, E, E, seg * LDK, seg * E, N * E, 1, 0, None, st),
            "rr_gemm_tn")
    del dlg
    # ---- pointer MLP: input gradient and weight gradients
    mp = mlp_train_pack(policy)
    dg0 = torch.empty(rows, E, device=dev)
    # ATSP tours all have the same length: every dumped row is live, no flag to consult
    live = None if env_name == "atsp" else dump["meta"]
    # precision="16-mixed" (opt-in, configs/trainer/default.yaml:8): one bf16 piece per operand in the pointer MLP's backward products
    half = getattr(policy, "precision", "32") == "16-mixed"
    L.check(lib.rr_mlp_rows(mp["bwd"], 3 if half else 1, L.ptr(dump["g0"]), L.ptr(dg), L.ptr(dg0), L.ptr(live), Bp, T * S, seg, st), "rr_mlp_rows")
    dW1, db1 = torch.zeros(4 * E, E, device=dev), torch.zeros(4 * E, device=dev)
    dW2, db2 = torch.zeros(E, 4 * E, device=dev), torch.zeros(E, device=dev)
    L.check((lib.rr_mlp_wgrad16 if half else lib.rr_mlp_wgrad)(mp["wgrad"], L.ptr(dump["g0"]), L.ptr(dg), L.ptr(dW1), L.ptr(db1), L.ptr(dW2), L.ptr(db2),
                                                               L.ptr(live), Bp, T * S, seg, L.ptr(wgrad_workspace(dev)), st), "rr_mlp_wgrad")
    del dg
    # ---- masked multi-head attention: d keys, d values, d query -> the step-context tables
    atsp = env_name == "atsp"
    nscal = 0 if atsp else (1 if env_name == "rcvrp" else 4)
    dK, dV = torch.empty(Bp, N, E, device=dev), torch.empty(Bp, N, E, device=dev)
    dctxB = torch.empty(Bp, N, E, device=dev)
    dctxA = torch.empty(Bp, N, E, device=dev) if atsp else None
    dws = torch.zeros(max(nscal, 1), E, device=dev)
    wstate = None
    if nscal:
        wstate = policy.decoder.context_embedding.project_context.weight.detach()[:, E:E + nscal].t().contiguous().float()
    first = dump["first"].contiguous() if atsp else None
    ia = L.DecAttnIO()
    ia.dg0, ia.meta, ia.scal, ia.first = L.ptr(dg0), L.ptr(dump["meta"]), L.ptr(dump.get("scal")) if nscal else None, L.ptr(first)
    ia.K, ia.V, ia.Kt = L.ptr(K), L.ptr(V), L.ptr(Kt)
    ia.ctxA, ia.ctxB, ia.wstate = L.ptr(cache.ctx_a) if atsp else None, L.ptr(cache.ctx_b), L.ptr(wstate)
    ia.dK, ia.dV, ia.dctxA, ia.dctxB, ia.dwstate = L.ptr(dK), L.ptr(dV), L.ptr(dctxA), L.ptr(dctxB), L.ptr(dws) if nscal else None
    ia.Bp, ia.N, ia.S, ia.T, ia.nscal, ia.seg_stride = Bp, N, S, T, nscal, seg
    L.check(lib.rr_dec_attn_bwd(ia, st), "rr_dec_attn_bwd")
    ll = logp.view(Bp, Tst, S)[:, :T].sum(1).t().reshape(-1)              # r = s*Bp + b
    return {"dK": dK, "dV": dV, "dL": dL, "dctxA": dctxA, "dctxB": dctxB, "dwstate": dws[:nscal] if nscal else None,
            "dW1": dW1, "db1": db1, "dW2": dW2, "db2": db2, "dalpha": dscal[0], "dbeta": dscal[1] if Dur is not None else None,
            "log_likelihood": ll}
