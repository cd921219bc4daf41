"""`-m gpu`: the hand-written backward kernels of the REINFORCE step (csrc/rr_train_dec.hip), each against the same formula
in torch ops (fp32 / float64 autograd), then the whole decoder backward against autograd through a torch restatement of
RRNetDecoder.forward + process_logits (rrnco/models/decoder.py:151-329, decoding.py:311-361) on the tensors the rollout used."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests import helpers as H

pytestmark = pytest.mark.gpu
E = 128


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def _mlp_weights(seed=0):
    g = torch.Generator().manual_seed(seed)
    W1 = (torch.randn(512, E, generator=g) / math.sqrt(E)).cuda()
    b1 = (0.1 * torch.randn(512, generator=g)).cuda()
    W2 = (torch.randn(E, 512, generator=g) / math.sqrt(512)).cuda()
    b2 = (0.1 * torch.randn(E, generator=g)).cuda()
    return W1, b1, W2, b2


def _seg_rows(nseg, seg_rows, stride, width, gen, safe_for=None):
    """Rows laid out in segments.  `safe_for` = (W1, b1): only rows whose hidden pre-activations all stay away from the ReLU
    kink (|W1 x + b1| > 1e-3) are used, so that the comparison with float64 is not decided by which side of zero a
    pre-activation of size 1e-6 lands on (the kernels compute it to ~2^-16 relative; a flip is a legitimate outcome)."""
    need = nseg * seg_rows
    rows = torch.randn(3 * need + 64, width, generator=gen)
    if safe_for is not None:
        W1, b1 = safe_for
        pre = F.linear(rows.double(), W1.double().cpu(), b1.double().cpu())
        rows = rows[pre.abs().min(1).values > 1e-3]
    assert rows.shape[0] >= need
    full = torch.zeros(nseg * stride, width)
    for s in range(nseg):
        full[s * stride:s * stride + seg_rows] = rows[s * seg_rows:(s + 1) * seg_rows]
    idx = torch.cat([torch.arange(s * stride, s * stride + seg_rows) for s in range(nseg)])
    return full.cuda(), idx.cuda()


@pytest.mark.parametrize("nseg,seg_rows,stride", [(1, 1000, 1000), (3, 300, 350), (2, 4100, 4100)])
def test_mlp_rows_forward_and_input_gradient(nseg, seg_rows, stride):
    from rrnco_amd import _lib as L, packing
    W1, b1, W2, b2 = _mlp_weights()
    mp = packing.pack_mlp_train(W1, b1, W2, b2)
    gen = torch.Generator().manual_seed(1)
    X, idx = _seg_rows(nseg, seg_rows, stride, E, gen, safe_for=(W1, b1))
    dY, _ = _seg_rows(nseg, seg_rows, stride, E, gen)
    out = torch.full_like(X, 7.0)
    L.check(L.lib().rr_mlp_rows(mp["fwd"], 0, L.ptr(X), None, L.ptr(out), None, nseg, seg_rows, stride, L.stream()), "fwd")
    Xd = X[idx].double().requires_grad_()
    ref = Xd + F.linear(F.relu(F.linear(Xd, W1.double(), b1.double())), W2.double(), b2.double())
    assert _rel(out[idx], ref.detach()) < 3e-5
    ref.backward(dY[idx].double())
    dX = torch.full_like(X, 7.0)
    L.check(L.lib().rr_mlp_rows(mp["bwd"], 1, L.ptr(X), L.ptr(dY), L.ptr(dX), None, nseg, seg_rows, stride, L.stream()), "bwd")
    assert _rel(dX[idx], Xd.grad) < 3e-5
    if stride > seg_rows:                      # rows between the segments are not touched
        gap = torch.ones(nseg * stride, dtype=torch.bool, device="cuda"); gap[idx] = False
        assert bool((dX[gap] == 7.0).all()) and bool((out[gap] == 7.0).all())


@pytest.mark.parametrize("use_ws", [False, True])
@pytest.mark.parametrize("nseg,seg_rows,stride", [(1, 1000, 1000), (3, 300, 350), (4, 5000, 5000)])
def test_mlp_weight_gradients(nseg, seg_rows, stride, use_ws):
    from rrnco_amd import _lib as L, packing
    W1, b1, W2, b2 = _mlp_weights(3)
    mp = packing.pack_mlp_train(W1, b1, W2, b2)
    gen = torch.Generator().manual_seed(2)
    X, idx = _seg_rows(nseg, seg_rows, stride, E, gen, safe_for=(W1, b1))
    dY, _ = _seg_rows(nseg, seg_rows, stride, E, gen)
    dW1, db1 = torch.zeros(512, E, device="cuda"), torch.zeros(512, device="cuda")
    dW2, db2 = torch.zeros(E, 512, device="cuda"), torch.zeros(E, device="cuda")
    ws = torch.empty(64 * 2 * 512 * E, device="cuda") if use_ws else None       # row splits' partials, reduced in a fixed order
    L.check(L.lib().rr_mlp_wgrad(mp["wgrad"], L.ptr(X), L.ptr(dY), L.ptr(dW1), L.ptr(db1), L.ptr(dW2), L.ptr(db2),
                                 None, nseg, seg_rows, stride, L.ptr(ws), L.stream()), "wgrad")
    P = [t.double().requires_grad_() for t in (W1, b1, W2, b2)]
    y = F.linear(F.relu(F.linear(X[idx].double(), P[0], P[1])), P[2], P[3])
    y.backward(dY[idx].double())
    for got, ref, nm in zip((dW1, db1, dW2, db2), P, ("dW1", "db1", "dW2", "db2")):
        assert _rel(got, ref.grad) < 1e-4, (nm, _rel(got, ref.grad))


@pytest.mark.parametrize("use_ws", [False, True])
@pytest.mark.parametrize("batch,M,P,msplit", [(3, 333, 100, 1), (2, 1000, 101, 1), (1, 5000, 384, 4), (1, 700, 512, 2), (1, 257, 128, 1),
                                              (2, 3000, 128, 16)])
def test_gemm_tn(batch, M, P, msplit, use_ws):
    """msplit > 1: float atomics into C, or (use_ws) per-split partials reduced in a fixed order — then bit-reproducible."""
    from rrnco_amd import _lib as L
    gen = torch.Generator().manual_seed(5)
    lda = 112 if P <= 112 else P
    A = torch.zeros(batch, M, lda); A[:, :, :P] = torch.randn(batch, M, P, generator=gen)
    B = torch.randn(batch, M, E, generator=gen)
    A, B = A.cuda(), B.cuda()
    C = torch.zeros(batch, P, E, device="cuda")
    ws = torch.empty(batch * msplit * P * 128, device="cuda") if use_ws else None
    L.check(L.lib().rr_gemm_tn(L.ptr(A), L.ptr(B), L.ptr(C), batch, M, P, lda, E, E, M * lda, M * E, P * E, msplit, 0, L.ptr(ws), L.stream()), "gemm_tn")
    ref = torch.einsum("bmp,bmq->bpq", A[:, :, :P].double(), B.double())
    assert _rel(C, ref) < 2e-6
    if use_ws and msplit > 1:
        C2 = torch.full_like(C, 7.0)          # (accumulate = 0: the reduction overwrites)
        L.check(L.lib().rr_gemm_tn(L.ptr(A), L.ptr(B), L.ptr(C2), batch, M, P, lda, E, E, M * lda, M * E, P * E, msplit, 0, L.ptr(ws), L.stream()), "gemm_tn")
        assert torch.equal(C, C2)


def _ref_decoder_ll(K, V, Lk, ctxA, ctxB, W1, b1, W2, b2, alpha, D, actions, tanh_clip=10.0, temp=1.0):
    """ATSP teacher-forced decoder on given cache tensors [b,N,E]; actions [b,S,N] -> ll [b,S] (all decode steps at once)."""
    b, S, N = actions.shape
    T = N - 1
    first, prev, target = actions[..., 0], actions[..., :T], actions[..., 1:]
    idx = lambda t, i: t.gather(1, i.reshape(b, -1, 1).expand(-1, -1, t.size(-1)))                    # noqa: E731
    q = idx(ctxA, first).unsqueeze(2) + idx(ctxB, prev).view(b, S, T, E)
    visited = F.one_hot(prev, N).cumsum(dim=2) > 0
    mask = ~visited
    q = q.reshape(b, S * T, E)
    heads = lambda t: t.unflatten(-1, (8, -1)).transpose(1, 2)                                        # noqa: E731
    sc = heads(q) @ heads(K).transpose(-1, -2) / 4.0
    sc = sc.masked_fill(~mask.reshape(b, 1, S * T, N), float("-inf"))
    h = torch.softmax(sc, -1) @ heads(V)
    g0 = h.transpose(1, 2).flatten(-2) + q
    g = g0 + F.linear(F.relu(F.linear(g0, W1, b1)), W2, b2)
    logits = torch.bmm(g, Lk.transpose(1, 2)) / math.sqrt(E)
    logits = torch.log(torch.exp(logits - alpha * idx(D, prev)) + 1e-6)
    logits = (torch.tanh(logits) * tanh_clip).masked_fill(~mask.reshape(b, S * T, N), float("-inf")) / temp
    logp = F.log_softmax(logits, dim=-1).gather(-1, target.reshape(b, S * T, 1)).view(b, S, T)
    return logp.sum(-1)


@pytest.mark.parametrize("name", ["atsp_n20_b4_pomo", "atsp_n100_b2_pomo"])
def test_decoder_backward_matches_autograd_on_the_rollouts_own_cache(name):
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    from rrnco_amd.models import dec_backward
    fx = H.load_fixture(name)
    w = H.atsp_weights(fx)
    pol = H.make_policy(w).train()
    env = ATSPEnv(generator_params=dict(num_loc=fx["N"]), check_solution=True)
    st = H.fixture_state(fx)
    td = TensorDict({k: v.cuda() for k, v in st.items()}, batch_size=[st["locs"].shape[0]])
    td["sample_idx"] = fx["sample_idx"].cuda()
    td = env.reset(td)
    D = td["distance_matrix"].contiguous()
    S, N = fx["S"], fx["N"]
    cap = {}
    with torch.no_grad():
        out = pol(td, env, phase="train", decode_type="multistart_sampling", num_starts=S, seed=5, capture=cap)
    cache, dump = cap["cache"], cap["dump"]
    Bp = D.shape[0]
    gll = torch.randn(S * Bp, generator=torch.Generator().manual_seed(3)).cuda() / (S * Bp)
    res = dec_backward.decoder_backward(pol, cache, dump, D, None, gll)
    # the replayed log-likelihood is the rollout's (same g, logits recomputed in fp32)
    assert torch.allclose(res["log_likelihood"], out["log_likelihood"], rtol=1e-5, atol=2e-4)
    # autograd through the torch restatement on the rollout's own cache tensors (float64)
    P = dict(pol.named_parameters())
    leaves = [cache.glimpse_key, cache.glimpse_val, cache.logit_key, cache.ctx_a, cache.ctx_b,
              P["decoder.pointer.ffn.lins.0.weight"], P["decoder.pointer.ffn.lins.0.bias"],
              P["decoder.pointer.ffn.lins.1.weight"], P["decoder.pointer.ffn.lins.1.bias"], P["decoder.alpha"]]
    leaves = [t.detach().double().contiguous().requires_grad_() for t in leaves]
    acts = out["actions"].view(S, Bp, N).transpose(0, 1)
    ll = _ref_decoder_ll(*leaves, D.double(), acts)
    assert torch.allclose(ll.float().t().reshape(-1), out["log_likelihood"], rtol=1e-5, atol=5e-4)
    ll.backward(gll.double().view(S, Bp).t())
    names = ["dK", "dV", "dL", "dctxA", "dctxB", "dW1", "db1", "dW2", "db2", "dalpha"]
    for nm, leaf in zip(names, leaves):
        got, ref = res[nm].reshape(leaf.grad.shape), leaf.grad
        # dW1 / db1 see the ReLU kink: a hidden pre-activation within ~1e-5 of zero may land on either side in the split-bf16
        # recomputation (one such unit among the 10^4..10^6 of these batches moves them by a few 1e-3)
        tol = 1e-2 if nm in ("dW1", "db1") else 2e-3
        assert _rel(got, ref) < tol, (nm, _rel(got, ref))


@pytest.mark.parametrize("B,N,S", [(3, 5, 5), (2, 17, 17), (2, 33, 20), (1, 18, 3)])
def test_decoder_backward_on_odd_shapes(B, N, S):
    """The row tilings of the backward kernels at their edges: fewer decode steps than a tile (N = 5: only left-over tiles), exactly
    one full tile per rollout (N = 17: T = 16, nothing left over), fewer rollouts than a tile, a ragged last tile."""
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    from rrnco_amd.models import RRNetPolicy, dec_backward
    torch.manual_seed(100 + N)
    pol = RRNetPolicy(env_name="atsp", embed_dim=128, num_heads=8, num_encoder_layers=2, normalization="instance",
                      use_graph_context=False, nab_type="gating", init_embedding_kwargs=dict(sample_size=min(25, N - 1))).cuda().train()
    env = ATSPEnv(generator_params=dict(num_loc=N, device="cuda"), check_solution=True, device="cuda")
    td = env.reset(env.generator(B, generator=torch.Generator(device="cuda").manual_seed(N)))
    D = td["distance_matrix"].contiguous()
    cap = {}
    with torch.no_grad():
        out = pol(td, env, phase="train", decode_type="multistart_sampling", num_starts=S, seed=5, capture=cap)
    cache, dump = cap["cache"], cap["dump"]
    gll = torch.randn(S * B, generator=torch.Generator().manual_seed(3)).cuda() / (S * B)
    res = dec_backward.decoder_backward(pol, cache, dump, D, None, gll)
    assert torch.allclose(res["log_likelihood"], out["log_likelihood"], rtol=1e-5, atol=2e-4)
    P = dict(pol.named_parameters())
    leaves = [cache.glimpse_key, cache.glimpse_val, cache.logit_key, cache.ctx_a, cache.ctx_b,
              P["decoder.pointer.ffn.lins.0.weight"], P["decoder.pointer.ffn.lins.0.bias"],
              P["decoder.pointer.ffn.lins.1.weight"], P["decoder.pointer.ffn.lins.1.bias"], P["decoder.alpha"]]
    leaves = [t.detach().double().contiguous().requires_grad_() for t in leaves]
    acts = out["actions"].view(S, B, N).transpose(0, 1)
    ll = _ref_decoder_ll(*leaves, D.double(), acts)
    ll.backward(gll.double().view(S, B).t())
    for nm, leaf in zip(["dK", "dV", "dL", "dctxA", "dctxB", "dW1", "db1", "dW2", "db2", "dalpha"], leaves):
        got, ref = res[nm].reshape(leaf.grad.shape), leaf.grad
        tol = 1e-2 if nm in ("dW1", "db1") else 2e-3
        assert _rel(got, ref) < tol, (nm, _rel(got, ref))


# ---------------------------------------------------------------- encoder block backward (csrc/rr_train_enc.hip)
@pytest.mark.parametrize("B,N", [(3, 20), (2, 100), (2, 101)])
def test_instance_norm_backward(B, N):
    from rrnco_amd import _lib as L
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(B, N, E, generator=gen).cuda()
    dy1, dy2 = torch.randn(B, N, E, generator=gen).cuda(), torch.randn(B, N, E, generator=gen).cuda()
    gamma = (1 + 0.3 * torch.randn(E, generator=gen)).cuda()
    prev = torch.randn(B, N, E, generator=gen).cuda()
    dx, dg, db = prev.clone(), torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda")
    L.check(L.lib().rr_inorm_bwd(L.ptr(x), L.ptr(dy1), L.ptr(dy2), L.ptr(gamma), L.ptr(dx), L.ptr(dg), L.ptr(db), B, N, 1, L.stream()), "inorm")
    xd, gd = x.double().requires_grad_(), gamma.double().requires_grad_()
    bd = torch.zeros(E, dtype=torch.float64, device="cuda", requires_grad=True)
    mu = xd.mean(1, keepdim=True)
    y = (xd - mu) * torch.rsqrt(((xd - mu) ** 2).mean(1, keepdim=True) + 1e-5) * gd + bd
    y.backward((dy1 + dy2).double())
    assert _rel(dx - prev, xd.grad) < 1e-5 and _rel(dg, gd.grad) < 1e-5 and _rel(db, bd.grad) < 1e-5


@pytest.mark.parametrize("M", [100, 2000, 51200])
def test_linear_rows(M):
    from rrnco_amd import _lib as L, packing
    gen = torch.Generator().manual_seed(8)
    W = (torch.randn(E, E, generator=gen) / math.sqrt(E)).cuda()
    bias = torch.randn(E, generator=gen).cuda()
    x = torch.randn(M, E, generator=gen).cuda()
    out = torch.randn(M, E, generator=gen).cuda()
    out0 = out.clone()
    cs = torch.zeros(E, device="cuda")
    wp = packing.pack_a(W)
    L.check(L.lib().rr_linear_rows(L.ptr(wp), L.ptr(bias), L.ptr(x), L.ptr(out), M, 1, L.ptr(cs), L.stream()), "linear")
    assert _rel(out, out0.double() + F.linear(x.double(), W.double(), bias.double())) < 2e-6
    assert _rel(cs, x.double().sum(0)) < 1e-5


@pytest.mark.parametrize("B,N", [(3, 20), (2, 50), (2, 100), (2, 101)])
def test_aft_mixing_backward(B, N):
    """AFTFull (attn_freenet.py:309-324) backward from the tensors the training forward stores."""
    from rrnco_amd import _lib as L
    gen = torch.Generator().manual_seed(9)
    q, k, v = (torch.randn(B, N, E, generator=gen).cuda() for _ in range(3))
    bias = torch.randn(B, N, N, generator=gen).cuda()
    dy = torch.randn(B, N, E, generator=gen).cuda()
    leaves = [t.double().requires_grad_() for t in (q, k, v, bias)]
    qd, kd, vd, bd = leaves
    ea = torch.exp(torch.softmax(bd, dim=-1))
    ek = torch.exp(torch.softmax(kd, dim=1))
    num, den = ea @ (ek * vd), ea @ ek
    y = torch.sigmoid(qd) * num / den
    y.backward(dy.double())
    eaT = torch.zeros(B, 112, 112, device="cuda")
    eaT[:, :N, :N] = ea.detach().float().transpose(1, 2)
    f32 = lambda t: t.detach().float().contiguous()                                            # noqa: E731
    dq, dk, dv = (torch.empty(B, N, E, device="cuda") for _ in range(3))
    dbias = torch.empty(B, N, N, device="cuda")
    io = L.AftBwdIO()
    keep = [f32(ek), f32(num), f32(den)]
    io.dy, io.q, io.ek, io.v, io.num, io.den, io.eaT = L.ptr(dy), L.ptr(q), L.ptr(keep[0]), L.ptr(v), L.ptr(keep[1]), L.ptr(keep[2]), L.ptr(eaT)
    io.dq, io.dk, io.dv, io.dbias, io.N = L.ptr(dq), L.ptr(dk), L.ptr(dv), L.ptr(dbias), N
    L.check(L.lib().rr_aft_bwd(io, B, L.stream()), "aft_bwd")
    for got, leaf, nm in zip((dq, dk, dv, dbias), leaves, ("dq", "dk", "dv", "dbias")):
        assert _rel(got, leaf.grad) < 2e-4, (nm, _rel(got, leaf.grad))


def test_nab_backward_from_segment_moments_matches_the_per_unit_kernel():
    """csrc/rr_train.hip: k_nab_hist_bwd (O(1) per edge: per-segment moments + host prefix sums) against k_nab_train_bwd
    (O(128) per edge) and, through it, the torch formula (tests/test_gpu_train.py::test_nab_training_kernels_...)."""
    from oracle import restate
    from rrnco_amd import _lib as L, packing
    from rrnco_amd.models import enc_backward as EB
    fx = H.load_fixture("atsp_n100_b2_pomo")
    w = H.atsp_weights(fx)
    P = {k: v.cuda() for k, v in w.items()}
    st0 = restate.atsp_reset(H.fixture_state(fx))
    D = st0["distance_matrix"].cuda().contiguous()
    theta = restate.pairwise_angles(st0["locs"]).cuda().contiguous()
    gout = torch.randn(D.shape, generator=torch.Generator().manual_seed(0)).cuda()
    for blk in ("encoder.net.layers.3.col_encoding_block", "encoder.net.layers.0.row_encoding_block"):
        tab = EB._nab_tab(P, blk + ".angle_distance_fusion", P[blk + ".alpha"]).contiguous()
        ref = torch.zeros_like(tab)
        L.check(L.lib().rr_nab_train_bwd(L.ptr(tab), L.ptr(D), L.ptr(theta), L.ptr(gout), L.ptr(ref), D.numel(), L.stream()), "bwd")
        pwl = packing.fold_nab_pwl(w, blk + ".angle_distance_fusion", w[blk + ".alpha"]).cuda()
        hist = torch.zeros(1, 2 * 129 * 4 + 1, device="cuda")
        L.check(L.lib().rr_nab_hist_bwd(L.ptr(pwl), L.ptr(D), L.ptr(theta), L.ptr(gout), L.ptr(hist), D.numel(), L.stream()), "hist")
        got = EB.nab_grad_from_hist(tab[None], hist)[0]
        for i, nm in enumerate(("a_d", "b_d", "co_d", "cg_d", "a_a", "b_a", "co_a", "cg_a")):
            r_, g_ = ref[128 * i:128 * (i + 1)], got[128 * i:128 * (i + 1)]
            assert float((g_ - r_).abs().max()) < 2e-3 * float(r_.abs().max()) + 1e-4, (blk, nm, float((g_ - r_).abs().max()), float(r_.abs().max()))
        assert torch.allclose(got[1024:1031], ref[1024:1031], rtol=2e-3, atol=1e-3), (got[1024:1032], ref[1024:1032])


@pytest.mark.parametrize("split", [False, True])
def test_duration_nab_backward_kernels_match_float64_autograd(split):
    """csrc/rr_train_nabdur.hip against float64 autograd through the folded duration NAB (models/grad_replay._NabDurationFolded's
    own forward): every folded-parameter gradient, on edges that are not a multiple of the 16-edge tile."""
    from rrnco_amd import _lib as L
    from rrnco_amd.packing import pack_a, pack_bf16x2
    from rrnco_amd.models.grad_replay import _NabDurationFolded as F
    torch.manual_seed(3)
    dev = torch.device("cuda")
    M = 16 * 37 + 5
    a, b = torch.randn(384, device=dev), torch.randn(384, device=dev) * 0.5
    Mcat, cg = torch.randn(128, 384, device=dev) * 0.08, torch.randn(128, device=dev) * 0.1
    co, ko = torch.randn(384, device=dev) * 0.1, torch.randn(3, device=dev) * 0.1
    Wg2, bg2 = torch.randn(3, 128, device=dev) * 0.2, torch.randn(3, device=dev) * 0.1
    inv_tau, bo, alpha = torch.tensor(1.3, device=dev), torch.tensor(0.2, device=dev), torch.tensor(0.7, device=dev)
    x3 = torch.rand(M, 3, device=dev) * torch.tensor([1.0, 6.28, 1.0], device=dev) - torch.tensor([0.0, 3.14, 0.0], device=dev)
    gout = torch.randn(M, device=dev)
    prm = [a, b, Mcat, cg, co, ko, Wg2, bg2, inv_tau, bo, alpha]
    p64 = [t.double().requires_grad_(True) for t in prm]
    _, _, _, _, _, _, g64, po64 = F._forward_parts(x3.double(), *p64[:9])
    out = ((g64 * po64).sum(-1) + p64[9]) * p64[10]
    ref = torch.autograd.grad(out, p64, gout.double())
    w = L.NabDurBwdW()
    scal = torch.cat([bg2, ko, inv_tau.reshape(1), bo.reshape(1), alpha.reshape(1)]).contiguous()
    mc, mct = pack_a(Mcat), pack_a(Mcat.t().contiguous())
    w.a, w.b, w.co, w.cg, w.wg2, w.scal, w.mcat, w.mcatT = (L.ptr(a), L.ptr(b), L.ptr(co), L.ptr(cg), L.ptr(Wg2), L.ptr(scal), L.ptr(mc), L.ptr(mct))
    if split:          # bf16 pipe, two-piece operands
        ms, mst = pack_bf16x2(Mcat), pack_bf16x2(Mcat.t().contiguous())
        w.mcat_s, w.mcatT_s = L.ptr(ms), L.ptr(mst)
    grads, dmcat = torch.zeros(1680, device=dev), torch.zeros(128, 384, device=dev)
    dzf = torch.full((((M + 31) // 32) * 32 * 128,), float("nan"), device=dev)        # the kernels must not depend on its contents
    xs = [x3[:, i].contiguous() for i in range(3)]
    L.check(L.lib().rr_nabdur_bwd(w, L.ptr(xs[0]), L.ptr(xs[1]), L.ptr(xs[2]), L.ptr(gout), L.ptr(dzf), L.ptr(grads), L.ptr(dmcat), M,
                                  L.stream()), "rr_nabdur_bwd")
    g = grads.double()
    got = [g[0:384], g[384:768], dmcat.double(), g[1152:1280], g[768:1152], g[1667:1670], g[1280:1664].view(3, 128), g[1664:1667],
           g[1670], g[1671], g[1672]]
    names = ["a", "b", "Mcat", "cg", "co", "ko", "Wg2", "bg2", "inv_tau", "bo", "alpha"]
    for n, x, r in zip(names, got, ref):
        err = (x.reshape(r.shape) - r).norm() / (r.norm() + 1e-12)
        assert err < (5e-3 if n in ("a", "b") else 2e-4), (n, split, float(err))       # a / b: ReLU-kink flips of single units (see the MLP tests)


def test_linear_smallk_and_gate_backward_match_autograd():
    """csrc/rr_train_enc.hip: k_linear_smallk (the narrow Linear maps of the init embedding) and k_gate_bwd (ContextualGating,
    rrnco/models/env_embeddings/atsp.py:108-121, recomputed and differentiated around its scalar gate) against float64 autograd."""
    from rrnco_amd import _lib as L
    g = torch.Generator().manual_seed(11)
    M, K, Kp = 1003, 25, 28
    X = torch.zeros(M, Kp); X[:, :K] = torch.rand(M, K, generator=g)
    W, b = torch.randn(E, K, generator=g) / 5, torch.randn(E, generator=g)
    out = torch.empty(M, E, device="cuda")
    Xc, Wc, bc = X.cuda(), W.cuda(), b.cuda()
    L.check(L.lib().rr_linear_smallk(L.ptr(Xc), Kp, K, L.ptr(Wc), L.ptr(bc), L.ptr(out), M, L.stream()), "smallk")
    assert _rel(out.cpu(), F.linear(X[:, :K].double(), W.double(), b.double())) < 1e-6
    # gate: h = relu(hA | hB), g = sigmoid(w2 . h + b2), out = g node + (1 - g) dist
    hpre = torch.randn(M, 2 * E, generator=g).double().requires_grad_()
    node, dist = torch.randn(M, E, generator=g).double().requires_grad_(), torch.randn(M, E, generator=g).double().requires_grad_()
    w2, b2 = (torch.randn(2 * E, generator=g) / 8).double().requires_grad_(), torch.randn(1, generator=g).double().requires_grad_()
    dout, dn0 = torch.randn(M, E, generator=g).double(), torch.randn(M, E, generator=g)
    gt = torch.sigmoid(F.relu(hpre) @ w2 + b2)[:, None]
    ((gt * node + (1 - gt) * dist) * dout).sum().backward()
    c = lambda t: t.detach().float().contiguous().cuda()                                              # noqa: E731
    hA, hB = c(hpre[:, :E]), c(hpre[:, E:])
    dnode, ddist = dn0.clone().cuda(), torch.empty(M, E, device="cuda")
    dw2, db2 = torch.zeros(2 * E, device="cuda"), torch.zeros(1, device="cuda")
    keep = [c(w2), c(b2), c(node), c(dist), c(dout)]
    io = L.GateBwdIO()
    io.hA, io.hB, io.w2, io.b2, io.node, io.dist, io.dout = [L.ptr(t) for t in [hA, hB] + keep]
    io.dnode, io.ddist, io.dw2, io.db2, io.M, io.acc_node = L.ptr(dnode), L.ptr(ddist), L.ptr(dw2), L.ptr(db2), M, 1
    L.check(L.lib().rr_gate_bwd(io, L.stream()), "gate")
    assert _rel(torch.cat([hA, hB], 1).cpu(), hpre.grad) < 2e-6
    # (node.grad / dist.grad of the reference include nothing through hpre here: exactly the kernel's g dout / (1 - g) dout)
    assert _rel(dnode.cpu() - dn0, node.grad) < 2e-6 and _rel(ddist.cpu(), dist.grad) < 2e-6
    assert _rel(dw2.cpu(), w2.grad) < 1e-5 and _rel(db2.cpu(), b2.grad) < 1e-5


def _vrp_model(problem):
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import RCVRPEnv, RMTVRPEnv
    from rrnco_amd.models.rl import RRNet
    fx = H.load_fixture("rcvrp_n20_b4_pomo" if problem == "rcvrp" else "rcvrptw_n20_b4_pomo")
    w = H.rcvrp_weights(fx) if problem == "rcvrp" else H.rcvrptw_weights(fx)
    pol = H.make_policy(w, env_name=problem).train()
    env = RCVRPEnv(generator_params=dict(num_loc=fx["N"]), check_solution=True) if problem == "rcvrp" else RMTVRPEnv(generator_params=dict(num_loc=fx["N"]))
    inst = H.rcvrp_instance(fx) if problem == "rcvrp" else H.rcvrptw_instance(fx)
    td_in = TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[fx["B"]])
    td_in["sample_idx"] = fx["sample_idx"].cuda()
    return pol, RRNet(env, policy=pol), td_in


@pytest.mark.parametrize("problem", ["atsp", "rcvrp", "rcvrptw"])
def test_init_embedding_backward_on_kernels_equals_the_autograd_path(monkeypatch, problem):
    """models/init_backward.py (the init embeddings — atsp.py:69-121, rcvrp.py:88-150, rcvrptw.py:44-56 — differentiated on the library's
    kernels) against the torch-autograd formulation it replaced (RR_INIT_BWD_TORCH=1), through a whole training step on the same
    sampled tours."""
    from tests.test_gpu_train import _model
    grads = []
    for flag in ("0", "1"):
        monkeypatch.setenv("RR_INIT_BWD_TORCH", flag)
        if problem == "atsp":
            w, pol, model, st, td_in = _model(H.load_fixture("atsp_n100_b2_pomo"))
        else:
            pol, model, td_in = _vrp_model(problem)
        model.training_step(td_in, seed=11)
        grads.append({n: p.grad.clone() for n, p in pol.named_parameters()})
    names = [n for n in grads[0] if n.startswith("encoder.init_embedding.") and "distance_expert.row_combine" not in n and "distance_expert.col_combine" not in n]
    assert len(names) >= 12
    gnorm = sum(float((g.double() ** 2).sum()) for g in grads[1].values()) ** 0.5
    for n in names:         # (biases in front of an instance norm have analytically zero gradients: rounding noise on both paths)
        assert float(grads[1][n].abs().max()) > 0, n
        err = float((grads[0][n].double() - grads[1][n].double()).norm())
        assert err <= 2e-4 * float(grads[1][n].double().norm()) + 5e-7 * gnorm, (n, err)
    assert sum(float(grads[1][n].double().norm()) > 1e-4 * gnorm for n in names) >= 8          # most of them are real gradients
    for n in grads[0]:
        if n not in names:           # (float atomics in a few kernels: not bit-equal; analytically zero gradients — to_k.bias — are rounding noise)
            err = float((grads[0][n].double() - grads[1][n].double()).norm())
            assert err <= 1e-5 * float(grads[1][n].double().norm()) + 5e-7 * gnorm, (n, err)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_small_gemm_of_the_weight_folds(dtype):
    """csrc/rr_train_enc.hip: k_small_gemm (the folds' 128 x 128 products, so that no BLAS library runs in a repack or a training step)
    with both transposes, batches and sizes that are not multiples of the 16 x 16 tile."""
    from rrnco_amd import packing
    g = torch.Generator().manual_seed(2)
    for (nb, M, N, K) in ((12, 128, 128, 128), (3, 37, 50, 129), (1, 16, 1, 5)):
        for ta in (False, True):
            for tb in (False, True):
                A = torch.randn((nb, K, M) if ta else (nb, M, K), generator=g, dtype=dtype)
                B = torch.randn((nb, N, K) if tb else (nb, K, N), generator=g, dtype=dtype)
                got = packing.small_gemm(A.cuda(), B.cuda(), ta, tb).cpu()
                ref = torch.matmul((A.transpose(1, 2) if ta else A).double(), (B.transpose(1, 2) if tb else B).double())
                assert got.dtype == dtype and _rel(got, ref) < (1e-6 if dtype == torch.float32 else 1e-14)
    A2, B2 = torch.randn(40, 30, generator=g), torch.randn(30, 20, generator=g)
    assert _rel(packing.small_gemm(A2.cuda(), B2.cuda()).cpu(), A2.double() @ B2.double()) < 1e-6
