"""`-m gpu`: the REINFORCE training step of BASELINE configs[4] (ATSP), single rank.

Forward (encoder, sampling rollout, reward, loss, d loss / d ll) on the HIP kernels; parameter gradients by the
teacher-forced replay (rrnco_amd/models/grad_replay.py).  Checked against autograd through the op-for-op CPU oracle fed
the SAME sampled tours (evaluate mode) and the same d loss / d ll."""
import numpy as np
import pytest
import torch

from oracle import restate
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _model(fx):
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    from rrnco_amd.models.rl import RRNet
    w = H.atsp_weights(fx)
    pol = H.make_policy(w).train()
    env = ATSPEnv(generator_params=dict(num_loc=fx["N"]), check_solution=True)
    model = RRNet(env, policy=pol, num_augment=8)
    st = H.fixture_state(fx)
    td_in = TensorDict({k: v.cuda() for k, v in st.items()}, batch_size=[st["locs"].shape[0]])
    td_in["sample_idx"] = fx["sample_idx"].cuda()
    return w, pol, model, st, td_in


@pytest.mark.parametrize("name", ["atsp_n20_b4_pomo", "atsp_n100_b2_pomo"])
def test_training_step_gradients_match_oracle_autograd(name):
    fx = H.load_fixture(name)
    w, pol, model, st, td_in = _model(fx)
    out = model.training_step(td_in, seed=11)
    S, B, N = fx["S"], fx["B"], fx["N"]
    acts = out["actions"].cpu()
    assert acts.shape == (S * B, N) and bool((acts.sort(1).values == torch.arange(N)).all())      # sampled tours are permutations
    # the replay reproduces the rollout's log-likelihood (same tours, different kernels)
    assert torch.allclose(out["replay_log_likelihood"], out["log_likelihood"], rtol=2e-5, atol=2e-3)
    # loss / advantage as rl.py:123-128 + REINFORCE shared baseline (routefinder/model.py:189-195)
    r = out["normalized_reward"].cpu().view(S, B)
    adv = (r - r.mean(0, keepdim=True)).reshape(-1)
    assert torch.allclose(out["advantage"].cpu(), adv, atol=1e-5)
    terms = adv.double() * out["log_likelihood"].cpu().double()
    # (fp32 sum of S*B terms of size |advantage x LL| — LL is -200 at N = 100 on random weights — with heavy cancellation)
    assert abs(float(out["loss"]) + float(terms.mean())) < 4 * len(terms) ** 0.5 * 6e-8 * float(terms.abs().max()) + 1e-5
    gll = out["grad_log_likelihood"].cpu()
    assert torch.allclose(gll, -adv / (S * B), atol=1e-7)
    # oracle: autograd through the sequential CPU restatement on the sampled tours
    wg = {k: v.clone().requires_grad_() for k, v in w.items()}
    ref = restate.atsp_policy(wg, restate.atsp_reset(st), fx["sample_idx"], S, decode="evaluate", actions=acts[:, 1:])
    assert torch.allclose(ref["log_likelihood"], out["log_likelihood"].cpu(), rtol=2e-5, atol=2e-3)
    (ref["log_likelihood"] * gll).sum().backward()
    refs = {n: wg[n].grad for n, _ in pol.named_parameters()}
    gnorm = sum(float((g ** 2).sum()) for g in refs.values() if g is not None) ** 0.5
    num = 0.0
    for n, p in pol.named_parameters():
        g = p.grad.cpu()
        if refs[n] is None:
            assert float(g.abs().max()) == 0.0, n          # zero-filled so that every rank all-reduces the same layout
            continue
        err = float(((g - refs[n]) ** 2).sum()) ** 0.5
        assert err <= 5e-2 * float((refs[n] ** 2).sum()) ** 0.5 + 2e-5 * gnorm, (n, err)      # tolerances: tests/test_cpu.py
        num += err ** 2
    print(f"\n[{name}] sampled tours: |g - g_oracle| / |g_oracle| = {num ** 0.5 / gnorm:.2e} (|g_oracle| = {gnorm:.4e}); "
          "the same kernels against the REAL reference's autograd: tests/test_gpu_grad_reference.py")
    assert num ** 0.5 / gnorm < 5e-3, num ** 0.5 / gnorm
    assert abs(float(out["grad_norm"]) - gnorm) / gnorm < 5e-3


def test_optimizer_steps_change_the_policy_and_repack():
    """Three Adam steps: parameters move, the packed (MFMA-ordered) weights are rebuilt, the loss stays finite."""
    fx = H.load_fixture("atsp_n20_b4_pomo")
    w, pol, model, st, td_in = _model(fx)
    opt = torch.optim.Adam(pol.parameters(), lr=1e-4)
    before = {n: p.detach().clone() for n, p in pol.named_parameters()}
    key0 = pol.packed(torch.device("cuda")) is pol.packed(torch.device("cuda"))
    assert key0
    packed0 = pol.packed(torch.device("cuda"))
    losses = []
    for it in range(3):
        out = model.training_step(td_in, optimizer=opt, seed=100 + it)
        losses.append(float(out["loss"]))
    assert all(np.isfinite(losses))
    moved = sum(int(not torch.equal(before[n], p.detach())) for n, p in pol.named_parameters())
    assert moved > 150
    assert pol.packed(torch.device("cuda")) is not packed0


def test_fused_optimizer_updates_are_seen_by_the_pack_cache():
    """torch.optim.Adam(fused=True) updates parameters in place without bumping their version counters; the pack cache must
    still notice (packing.weights_fingerprint), or the next rollout would silently use the previous weights."""
    fx = H.load_fixture("atsp_n20_b4_pomo")
    w, pol, model, st, td_in = _model(fx)
    opt = torch.optim.Adam(pol.parameters(), lr=1e-3, fused=True)
    dev = torch.device("cuda")
    packed0 = pol.packed(dev)
    model.training_step(td_in, optimizer=opt, seed=7)
    assert pol.packed(dev) is not packed0
    assert pol.packed(dev) is pol.packed(dev)              # and it is stable again afterwards
    # the two optimizers agree, so a rollout after the fused step equals one after the plain step
    w2, pol2, model2, st2, td2 = _model(fx)
    opt2 = torch.optim.Adam(pol2.parameters(), lr=1e-3)
    model2.training_step(td2, optimizer=opt2, seed=7)
    pol.eval(); pol2.eval()
    env = model.env
    with torch.no_grad():
        a = pol(env.reset(td_in), env, phase="test", decode_type="multistart_greedy", num_starts=20)
        b = pol2(env.reset(td2), env, phase="test", decode_type="multistart_greedy", num_starts=20)
    same = (a["reward"] - b["reward"]).abs() < 1e-4          # greedy tours may flip at near-ties between the two optimizers
    assert float(same.float().mean()) > 0.9


def test_consecutive_fused_optimizer_steps_use_current_encoder_packs():
    """torch.optim.Adam(fused=True) bumps no version counter, so after invalidate_pack() the in-scope pack key of
    step k + 1 used to equal step k's and the ENCODER's training packs (transposed projections, FFN packs: models/enc_backward.py)
    were taken from the previous weights.  Three fused steps at a large learning rate; the gradients of the third step must equal
    those of a fresh policy that was handed the same weights (same tours: same seed, same sample)."""
    fx = H.load_fixture("atsp_n20_b4_pomo")
    w, pol, model, st, td_in = _model(fx)
    opt = torch.optim.Adam(pol.parameters(), lr=2e-3, fused=True)
    for it in range(2):
        model.training_step(td_in, optimizer=opt, seed=40 + it)
    assert getattr(pol, "_enc_train_pack", None) is None               # invalidate_pack() drops every derived pack
    weights_now = {n: p.detach().clone() for n, p in pol.named_parameters()}
    assert sum(int(not torch.equal(weights_now[n].cpu(), w[n])) for n in weights_now) > 150
    out = model.training_step(td_in, seed=99)                          # step 3 on the twice-updated weights (no optimizer: keep the grads)
    g_live = {n: p.grad.detach().clone() for n, p in pol.named_parameters()}
    # a fresh policy with the same weights: nothing cached anywhere
    w2, pol2, model2, st2, td2 = _model(fx)
    with torch.no_grad():
        for n, p in pol2.named_parameters():
            p.copy_(weights_now[n])
    pol2.invalidate_pack()
    out2 = model2.training_step(td2, seed=99)
    assert torch.equal(out["actions"], out2["actions"])
    gn = sum(float((p.grad ** 2).sum()) for _, p in pol2.named_parameters()) ** 0.5
    err = sum(float(((g_live[n] - p.grad) ** 2).sum()) for n, p in pol2.named_parameters()) ** 0.5
    assert err <= 1e-6 * gn, (err, gn)                                  # same kernels, same operands: (near) bit-equal
    enc = [n for n in g_live if n.startswith("encoder.net.layers.")]
    assert enc and all(torch.allclose(g_live[n], dict(pol2.named_parameters())[n].grad, rtol=1e-5, atol=1e-7 * gn) for n in enc)


def test_pack_scope_verifies_once_per_step_and_expires():
    """RRNet.training_step checks the weight pack once (RRNetPolicy.pack_scope: forward and backward of one step see the same
    weights); outside a scope every packed() call verifies again, so an in-place update that bumps no version counter is seen."""
    from rrnco_amd import packing
    fx = H.load_fixture("atsp_n20_b4_pomo")
    w, pol, model, st, td_in = _model(fx)
    dev = torch.device("cuda")
    calls = {"n": 0}
    real = packing.weights_fingerprint

    def counting(module, tensors=None):
        calls["n"] += 1
        return real(module, tensors)
    packing.weights_fingerprint = counting
    try:
        pol.packed(dev)                                    # (a cached pack: the step below has something to verify)
        n0 = calls["n"]
        model.training_step(td_in, seed=3)                 # forward + decoder backward + encoder backward: several packed() calls
        assert calls["n"] == n0 + 1
        pol.invalidate_pack()                              # what an optimizer step is followed by: nothing to compare with,
        model.training_step(td_in, seed=3)                 # so the next step packs without reading a fingerprint back
        assert calls["n"] == n0 + 1
        p0 = pol.packed(dev)                               # outside a scope: verified (and rebuilt: the in-scope key carries no fingerprint)
        assert calls["n"] == n0 + 2 and getattr(pol, "_pack_scope", False) is False
        assert pol.packed(dev) is p0
        with torch.no_grad():                              # what a fused optimizer does: new values, same version counter
            q = pol.decoder.pointer.ffn.lins[0].weight
            v = q._version
            q.data.mul_(1.5)
            assert q._version == v
        assert pol.packed(dev) is not p0                   # outside a scope: verified, rebuilt
    finally:
        packing.weights_fingerprint = real


def test_nab_training_kernels_match_the_torch_formula():
    """csrc/rr_train.hip: forward value and d loss / d (folded table) of the gating NAB against the same formula in torch ops."""
    from rrnco_amd import _lib as L
    from rrnco_amd.models import grad_replay as G
    fx = H.load_fixture("atsp_n100_b2_pomo")
    w = H.atsp_weights(fx)
    P = {k: v.cuda().requires_grad_() for k, v in w.items()}
    p = "encoder.net.layers.3.col_encoding_block"
    st0 = restate.atsp_reset(H.fixture_state(fx))
    D = st0["distance_matrix"].cuda()
    theta = restate.pairwise_angles(st0["locs"]).cuda()
    gout = torch.from_numpy(np.random.default_rng(0).standard_normal(tuple(D.shape)).astype(np.float32)).cuda()
    out_hip = G._nab_folded(P, p + ".angle_distance_fusion", D, theta, P[p + ".alpha"])
    out_hip.backward(gout)
    g_hip = {k: v.grad.clone() for k, v in P.items() if v.grad is not None}
    for v in P.values():
        v.grad = None
    Pc = {k: v.detach().cpu().requires_grad_() for k, v in P.items()}
    out_ref = G._nab_folded(Pc, p + ".angle_distance_fusion", D.cpu(), theta.cpu(), Pc[p + ".alpha"])     # torch-op branch
    out_ref.backward(gout.cpu())
    assert torch.allclose(out_hip.detach().cpu(), out_ref.detach(), atol=2e-5)
    assert set(g_hip) == {k for k, v in Pc.items() if v.grad is not None} and len(g_hip) == 13      # 2 x (Linear(1,E) w,b + Linear(E,E) w,b) + gate w,b + out_lin w,b + alpha
    for k, g in g_hip.items():
        ref = Pc[k].grad
        tol = 1e-3 * float(ref.abs().max()) + 1e-4
        assert float((g.cpu() - ref).abs().max()) < tol, (k, float((g.cpu() - ref).abs().max()), tol)


@pytest.mark.parametrize("name", ["atsp_n20_b4_pomo", "atsp_n100_b2_pomo"])
def test_nab_parameter_gradients_from_one_kernel_equal_the_torch_route(name, monkeypatch):
    """csrc/rr_train.hip:rr_nab_tab_bwd (round 5): the moments of rr_nab_hist_bwd -> the gradients of DistAngleFusion's parameters of all
    twelve blocks in one launch, against the route it replaces (RR_NAB_TAB_TORCH=1: nab_grad_from_hist's float64 prefix sums + torch
    autograd through the fold) on the same sampled tours (same seed): every one of the 12 x 13 tensors."""
    fx = H.load_fixture(name)
    grads = {}
    for route in ("1", "0"):
        monkeypatch.setenv("RR_NAB_TAB_TORCH", route)
        w, pol, model, st, td_in = _model(fx)
        model.training_step(td_in, seed=11)
        grads[route] = {n: p.grad.detach().clone() for n, p in pol.named_parameters()}
    monkeypatch.delenv("RR_NAB_TAB_TORCH")
    seen = 0
    for n, g in grads["0"].items():
        ref = grads["1"][n]
        if ".angle_distance_fusion." in n or n.endswith("_encoding_block.alpha"):
            seen += 1
            tol = 2e-4 * float(ref.abs().max()) + 1e-7
            assert float((g - ref).abs().max()) <= tol, (n, float((g - ref).abs().max()), tol)
            assert float(ref.abs().max()) > 0.0, n
    assert seen == 13 * 12


def test_rcvrp_training_step_gradients_match_oracle_autograd():
    """The same step for RCVRP (sampled routes of different lengths, S = N+1 starts)."""
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import RCVRPEnv
    from rrnco_amd.models.rl import RRNet
    fx = H.load_fixture("rcvrp_n20_b4_pomo")
    w = H.rcvrp_weights(fx)
    pol = H.make_policy(w, env_name="rcvrp").train()
    # every dump row the rollout does not write (steps behind an instance's longest route) holds NaN instead of whatever the allocator
    # left there: the backward must not let them through (a NaN state scalar of a dead row used to poison dK of its instance)
    pol._debug_poison_dump = True
    env = RCVRPEnv(generator_params=dict(num_loc=fx["N"]), check_solution=True)
    model = RRNet(env, policy=pol)
    inst = H.rcvrp_instance(fx)
    td_in = TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[fx["B"]])
    td_in["sample_idx"] = fx["sample_idx"].cuda()
    out = model.training_step(td_in, seed=21)
    S, B = fx["S"], fx["B"]
    acts = out["actions"].cpu()
    assert acts.shape[0] == S * B
    assert torch.allclose(out["replay_log_likelihood"], out["log_likelihood"], rtol=2e-5, atol=2e-3)
    gll = out["grad_log_likelihood"].cpu()
    wg = {k: v.clone().requires_grad_() for k, v in w.items()}
    ref = restate.rcvrp_policy(wg, restate.rcvrp_reset(inst), fx["sample_idx"], S, decode="evaluate", actions=acts[:, 1:])
    T = min(ref["actions"].shape[1], acts.shape[1])
    assert torch.equal(ref["actions"][:, :T], acts[:, :T])
    assert torch.allclose(ref["log_likelihood"], out["log_likelihood"].cpu(), rtol=2e-5, atol=2e-3)
    (ref["log_likelihood"] * gll).sum().backward()
    refs = {n: wg[n].grad for n, _ in pol.named_parameters()}
    gnorm = sum(float((g ** 2).sum()) for g in refs.values() if g is not None) ** 0.5
    num = 0.0
    for n, p in pol.named_parameters():
        g = p.grad.cpu()
        if refs[n] is None:
            assert float(g.abs().max()) == 0.0, n
            continue
        err = float(((g - refs[n]) ** 2).sum()) ** 0.5
        assert err <= 5e-2 * float((refs[n] ** 2).sum()) ** 0.5 + 2e-5 * gnorm, (n, err)
        num += err ** 2
    assert num ** 0.5 / gnorm < 5e-3


@pytest.mark.parametrize("fixture", ["rcvrptw_n20_b4_pomo", "rmtvrp_n20_b8_pomo_variants"])
def test_rcvrptw_training_step_gradients_match_oracle_autograd(fixture):
    """... and for RCVRPTW (vrptw preset) and the multi-task RMTVRP instances (backhauls, open routes, distance limits):
    duration NAB, time-window masks, MTVRP context; sampling rollout on the fused kernel, gradients by the replay."""
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import RMTVRPEnv
    from rrnco_amd.models.rl import RRNet
    fx = H.load_fixture(fixture)
    w = H.rcvrptw_weights(fx)
    pol = H.make_policy(w, env_name="rcvrptw").train()
    pol._debug_poison_dump = True          # unwritten dump rows read NaN (see the RCVRP test above)
    env = RMTVRPEnv(generator_params=dict(num_loc=fx["N"]))
    model = RRNet(env, policy=pol)
    inst = H.rcvrptw_instance(fx)
    td_in = TensorDict({k: v.cuda() for k, v in inst.items()}, batch_size=[fx["B"]])
    td_in["sample_idx"] = fx["sample_idx"].cuda()
    out = model.training_step(td_in, seed=31)
    S, B = fx["S"], fx["B"]
    acts = out["actions"].cpu()
    assert acts.shape[0] == S * B
    assert torch.allclose(out["replay_log_likelihood"], out["log_likelihood"], rtol=2e-5, atol=4e-3)
    gll = out["grad_log_likelihood"].cpu()
    wg = {k: v.clone().requires_grad_() for k, v in w.items()}
    ref = restate.rcvrptw_policy(wg, restate.rmtvrp_reset(inst), fx["sample_idx"], S, decode="evaluate", actions=acts[:, 1:])
    T = min(ref["actions"].shape[1], acts.shape[1])
    assert torch.equal(ref["actions"][:, :T], acts[:, :T])
    (ref["log_likelihood"] * gll).sum().backward()
    refs = {n: wg[n].grad for n, _ in pol.named_parameters()}
    gnorm = sum(float((g ** 2).sum()) for g in refs.values() if g is not None) ** 0.5
    num = 0.0
    for n, p in pol.named_parameters():
        g = p.grad.cpu()
        if refs[n] is None:
            assert float(g.abs().max()) == 0.0, n
            continue
        err = float(((g - refs[n]) ** 2).sum()) ** 0.5
        assert err <= 5e-2 * float((refs[n] ** 2).sum()) ** 0.5 + 2e-5 * gnorm, (n, err)
        num += err ** 2
    assert num ** 0.5 / gnorm < 5e-3


def test_loss_backward_runs_the_reference_training_pattern_and_equals_training_step():
    """rrnco/models/rl.py:111-128 as the reference runs it: out = shared_step(batch, phase="train"); out["loss"].backward().
    The policy's log-likelihood carries the graph (policy._PolicyLogLikelihood); the gradients are training_step's."""
    fx = H.load_fixture("atsp_n20_b4_pomo")
    w, pol, model, st, td_in = _model(fx)
    model.training_step(td_in, seed=11)
    g1 = {n: p.grad.clone() for n, p in pol.named_parameters()}
    for p in pol.parameters():
        p.grad = None
    out = model.shared_step(td_in, phase="train", seed=11)
    assert out["log_likelihood"].requires_grad and out["loss"].requires_grad
    out["loss"].backward()
    gn = sum(float((g ** 2).sum()) for g in g1.values()) ** 0.5
    for n, p in pol.named_parameters():
        if p.grad is None:
            assert float(g1[n].abs().max()) == 0.0, n
            continue
        assert float((p.grad - g1[n]).norm()) <= 1e-4 * gn + 1e-3 * float(g1[n].norm()), n      # float atomics: order-dependent sums
    # under no_grad / in eval mode the same call returns plain tensors
    with torch.no_grad():
        assert not model.shared_step(td_in, phase="train", seed=11)["log_likelihood"].requires_grad


def test_gradient_clipping_as_the_reference_trainer():
    """configs/trainer/default.yaml:6 gradient_clip_val = 1.0: the gradients the optimizer sees have norm <= clip."""
    fx = H.load_fixture("atsp_n20_b4_pomo")
    w, pol, model, st, td_in = _model(fx)
    out = model.training_step(td_in, seed=11, grad_clip=0.05)
    total = torch.linalg.vector_norm(torch.stack([p.grad.norm() for p in pol.parameters()]))
    assert float(out["grad_norm"]) > 0.05 and abs(float(total) - 0.05) < 1e-4


def test_batch_norm_kernels_match_torch_batch_norm():
    """rr_bnorm_fwd / rr_bnorm_bwd (csrc/rr_bign.hip, rr_train_enc.hip): nn.BatchNorm1d in train mode over the flattened rows — output,
    running statistics (momentum 0.1, unbiased variance), and the backward with two incoming gradients and accumulation, against
    torch's own batch_norm in float64."""
    import torch.nn.functional as F
    from rrnco_amd import _lib as L
    lib, st = L.lib(), L.stream()
    g = torch.Generator().manual_seed(3)
    M, E = 6 * 37, 128
    x = (torch.randn(M, E, generator=g) * 1.7 + torch.randn(E, generator=g) * 3).cuda()
    res = torch.randn(M, E, generator=g).cuda()
    gamma, beta = (torch.rand(E, generator=g) + 0.5).cuda(), torch.randn(E, generator=g).cuda()
    rm, rv = torch.randn(E, generator=g).cuda(), (torch.rand(E, generator=g) + 0.5).cuda()
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    out, usum = torch.empty_like(x), torch.empty_like(x)
    ws = torch.empty(512, dtype=torch.float64, device="cuda")
    L.check(lib.rr_bnorm_fwd(L.ptr(x), L.ptr(res), L.ptr(gamma), L.ptr(beta), L.ptr(out), L.ptr(usum), L.ptr(ws), L.ptr(rm), L.ptr(rv), 0.1, M, st), "rr_bnorm_fwd")
    xd = (x + res).double().requires_grad_()
    gd, bd = gamma.double().requires_grad_(), beta.double().requires_grad_()
    ref = F.batch_norm(xd, rm_ref, rv_ref, gd, bd, True, 0.1, 1e-5)
    assert torch.equal(usum, x + res)
    assert float((out.double() - ref.detach()).abs().max()) < 2e-5
    assert float((rm.double() - rm_ref).abs().max()) < 1e-6 and float((rv.double() - rv_ref).abs().max()) < 1e-5
    dy1, dy2 = torch.randn(M, E, generator=g).cuda(), torch.randn(M, E, generator=g).cuda()
    ref.backward((dy1 + dy2).double())
    dx0 = torch.randn(M, E, generator=g).cuda()
    dx, dgm, dbt = dx0.clone(), torch.zeros(E, device="cuda"), torch.zeros(E, device="cuda")
    L.check(lib.rr_bnorm_bwd(L.ptr(usum), L.ptr(dy1), L.ptr(dy2), L.ptr(gamma), L.ptr(dx), L.ptr(dgm), L.ptr(dbt), L.ptr(ws), M, 1, st), "rr_bnorm_bwd")
    assert float(((dx - dx0).double() - xd.grad).abs().max()) < 2e-5 * float(xd.grad.abs().max()) + 1e-6
    assert float((dgm.double() - gd.grad).abs().max()) < 1e-4 * float(gd.grad.abs().max())
    assert float((dbt.double() - bd.grad).abs().max()) < 1e-4 * float(bd.grad.abs().max())
    dx2 = torch.empty_like(x)                       # not accumulating, one incoming gradient
    L.check(lib.rr_bnorm_bwd(L.ptr(usum), L.ptr(dy1), None, L.ptr(gamma), L.ptr(dx2), L.ptr(dgm), L.ptr(dbt), L.ptr(ws), M, 0, st), "rr_bnorm_bwd")
    xd2 = (x + res).double().requires_grad_()
    F.batch_norm(xd2, None, None, gamma.double(), beta.double(), True, 0.0, 1e-5).backward(dy1.double())
    assert float((dx2.double() - xd2.grad).abs().max()) < 2e-5 * float(xd2.grad.abs().max()) + 1e-6


@pytest.mark.parametrize("problem", ["atsp", "rcvrp", "rcvrptw"])
def test_batch_norm_train_mode_step_runs_and_matches_the_torch_replay(problem):
    """normalization='batch' with module.train() (attn_freenet.py:82-83, 102-103; the constructor default of RRNetPolicy): batch
    statistics across the instances of the call.  Since round 5 the encoder of such a step runs on kernels as well (models/bign.py:
    encode_bn_train with rr_bnorm_fwd; backward: models/enc_backward.py with rr_bnorm_bwd); RR_BN_TORCH=1 keeps the torch-op encoder.
    The step's gradients equal those of the all-torch teacher-forced replay, the running statistics
    move once per step, and the eval-mode forward (kernels, folded running statistics) works afterwards."""
    from rrnco_amd.envs import ATSPEnv, RCVRPEnv, RMTVRPEnv
    from rrnco_amd.models import RRNetPolicy
    from rrnco_amd.models.rl import RRNet
    dev = torch.device("cuda")
    torch.manual_seed(5)
    pol = RRNetPolicy(env_name=problem, embed_dim=128, num_heads=8, num_encoder_layers=2, normalization="batch",
                      use_graph_context=False, nab_type="gating", init_embedding_kwargs=dict(sample_size=10)).to(dev).train()
    gp = dict(num_loc=20, device=dev)
    env = {"atsp": lambda: ATSPEnv(generator_params=gp, check_solution=False, device=dev),
           "rcvrp": lambda: RCVRPEnv(generator_params=gp, check_solution=False, device=dev),
           "rcvrptw": lambda: RMTVRPEnv(generator_params=gp, device=dev)}[problem]()
    model = RRNet(env, policy=pol)
    batch = env.generator(6, generator=torch.Generator(device=dev).manual_seed(2))
    nn_ = batch["distance_matrix"].shape[-1]
    batch["sample_idx"] = torch.stack([torch.stack([torch.randperm(nn_, device=dev)[:10] for _ in range(nn_)]) for _ in range(6)])
    rm0 = pol.state_dict()["encoder.net.layers.0.row_encoding_block.norm1.normalizer.running_mean"].clone()
    grads = {}
    from rrnco_amd import _lib as L
    lib, calls = L.lib(), {"rr_bnorm_fwd": 0, "rr_bnorm_bwd": 0}
    real = {n: getattr(lib, n) for n in calls}
    for n in calls:
        setattr(lib, n, (lambda *a, _n=n: (calls.__setitem__(_n, calls[_n] + 1), real[_n](*a))[1]))
    try:
        for replay in ("hip", "torch"):
            pol.zero_grad(set_to_none=True)
            before = dict(calls)
            out = model.training_step(batch, optimizer=None, seed=7, replay=replay)
            assert torch.isfinite(out["loss"]) and torch.allclose(out["replay_log_likelihood"], out["log_likelihood"], atol=2e-3)
            grads[replay] = {n: p.grad.clone() for n, p in pol.named_parameters() if p.grad is not None}
            # round 5: the train-mode encoder runs on kernels (2 layers x 2 blocks x 5 norms), and with replay="hip" so does its backward
            assert calls["rr_bnorm_fwd"] - before["rr_bnorm_fwd"] == 20
            assert calls["rr_bnorm_bwd"] - before["rr_bnorm_bwd"] == (20 if replay == "hip" else 0)
    finally:
        for n in calls:
            setattr(lib, n, real[n])
    rm1 = pol.state_dict()["encoder.net.layers.0.row_encoding_block.norm1.normalizer.running_mean"]
    assert not torch.equal(rm0, rm1)                                        # the forward updated the running statistics
    num = sum(float((grads["hip"][n] - grads["torch"][n]).pow(2).sum()) for n in grads["torch"])
    den = sum(float(grads["torch"][n].pow(2).sum()) for n in grads["torch"])
    assert den > 0 and (num / den) ** 0.5 < 2e-3
    pol.eval()
    out = pol(env.reset(batch), env, phase="val", decode_type="multistart_greedy", num_starts=20, return_actions=True)
    assert bool(torch.isfinite(out["reward"]).all())
    # a step taken in EVAL mode differentiates the running-statistics network the kernels ran (not an instance-norm look-alike):
    # the replayed log-likelihood must equal the rollout's
    pol.zero_grad(set_to_none=True)
    out = model.training_step(batch, optimizer=None, seed=8, replay="hip")
    assert torch.allclose(out["replay_log_likelihood"], out["log_likelihood"], atol=2e-3)


@pytest.mark.parametrize("problem", ["atsp", "rcvrp", "rcvrptw"])
def test_training_step_launches_no_blas_kernel(problem):
    """The REINFORCE step (forward repack included) runs on the library's kernels and torch elementwise / reduction glue only: no
    hipBLASLt / rocBLAS kernel (`Cijk_*`, `rocblas_*`, gemv) among the device events of a step — the init embeddings and the fold chain
    are differentiated by models/init_backward.py and rr_small_gemm; the duration NAB's fold chain (RCVRPTW) stays
    on torch autograd with its matrix products on rr_small_gemm (grad_replay._SmallMM)."""
    from torch.profiler import profile, ProfilerActivity
    if problem == "atsp":
        w, pol, model, st, td_in = _model(H.load_fixture("atsp_n20_b4_pomo"))
    else:
        from tests.test_gpu_trainkernels import _vrp_model
        pol, model, td_in = _vrp_model(problem)
    opt = torch.optim.Adam(pol.parameters(), lr=1e-4, fused=True)
    model.training_step(td_in, optimizer=opt, seed=1)                    # (first call: allocations, lazy initialisation)
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        model.training_step(td_in, optimizer=opt, seed=2)
        torch.cuda.synchronize()
    names = {e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA}
    assert any(n.startswith("k_") or n.startswith("void k_") for n in names)
    blas = sorted(n for n in names if n.startswith("Cijk_") or "rocblas" in n.lower() or "gemv" in n.lower() or "hipblas" in n.lower())
    assert not blas, blas[:5]
