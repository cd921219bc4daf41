"""`-m gpu`: the opt-in precision="16-mixed" rollout (csrc/rr_rollout_w.inc, HALF): one fp16 piece per operand, fp32 accumulation,
fp32 softmax and logits — the arithmetic the reference itself runs on a GPU (torch.autocast in test.py:183, Lightning precision
16-mixed in configs/trainer/default.yaml:8, logits cast back to fp32 in rrnco/models/decoder.py:195-196).

Its yardstick is NOT the fp32 tolerance of the default build but the reference's own mixed-precision deviation:
tests/golden/*_autocast.npz hold the REAL reference run under torch.autocast (fp16 and bf16) on the instances, weights and
neighbour samples of two fp32 fixtures (oracle/gen_golden.py autocast).  Under autocast the reference keeps 0.5 % .. 1.5 % of its
own fp32 tours and moves the log-likelihoods by 1.8 .. 3.0 on average (it also runs the ENCODER in half precision; the variant
here keeps the encoder and the decoder cache fp32-equivalent, so it has to stay well inside that band)."""
import numpy as np
import pytest
import torch

from oracle import restate
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _run(fx, w, precision):
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    from rrnco_amd.models import RRNetPolicy
    pol = H.make_policy(w, device="cuda:0")
    pol.precision = precision
    env = ATSPEnv(generator_params=dict(num_loc=fx["N"]))
    td = TensorDict({"locs": fx["locs"].cuda(), "distance_matrix": fx["distance_matrix"].cuda(),
                     "sample_idx": fx["sample_idx"].cuda()}, batch_size=[fx["B"]])
    out = pol(env.reset(td), env, phase="val", decode_type="multistart_greedy", num_starts=fx["S"], return_actions=True, range_guard="sync")
    torch.cuda.synchronize()
    assert pol.last_range_flags == 0
    return {k: v.cpu() for k, v in out.items() if torch.is_tensor(v)}


def test_policy_rejects_unknown_precision():
    from rrnco_amd.models import RRNetPolicy
    with pytest.raises(ValueError):
        RRNetPolicy(env_name="atsp", precision="8-bit")
    assert RRNetPolicy(env_name="atsp", precision="16-mixed", num_encoder_layers=1, normalization="instance").precision == "16-mixed"


@pytest.mark.parametrize("name", ["atsp_n100_b2_pomo", "atsp_n100_b2_pomo_trained"])
def test_sixteen_mixed_rollout_stays_inside_the_references_own_autocast_deviation(name):
    fx = H.load_fixture(name)
    ac = dict(np.load(H.fixture_path(name + "_autocast")))
    w = H.atsp_weights(fx)
    S, N = fx["S"], fx["N"]
    full = _run(fx, w, "32")
    half = _run(fx, w, "16-mixed")
    ref_a, ref_ll, ref_r = fx["actions"], fx["log_likelihood"], fx["reward"]
    Bp = ref_r.shape[0] // S
    best = lambda r: r.view(S, Bp).max(0).values                                     # noqa: E731
    assert restate.atsp_check(half["actions"])                                        # valid tours whatever the precision
    assert torch.isfinite(half["log_likelihood"]).all() and torch.isfinite(half["reward"]).all()
    # the variant really ran other arithmetic: its log-likelihoods are not the default build's
    assert not torch.equal(half["log_likelihood"], full["log_likelihood"])
    same_ref = (half["actions"] == ref_a).all(1).float().mean().item()
    same_full = (half["actions"] == full["actions"]).all(1).float().mean().item()
    dll = (half["log_likelihood"] - ref_ll).abs()
    gap = (best(ref_r) - best(half["reward"])).mean().item()                          # > 0: the variant's best tour is worse
    r16_same = float(ac["fp16_tours_identical"])
    r16_dll = np.abs(ac["fp16_log_likelihood"] - ref_ll.numpy())
    r16_gap = float((best(ref_r) - best(torch.from_numpy(ac["fp16_reward"]))).mean())
    print(f"\n[{name}] 16-mixed rollout vs the reference's fp32 golden: tours identical {same_ref:.4f} (default build: "
          f"{(full['actions'] == ref_a).all(1).float().mean().item():.4f}; 16-mixed vs default build {same_full:.4f}), |LL - fp32| mean {dll.mean():.3e} "
          f"max {dll.max():.3e}, best-of-{S} cost gap {gap:+.3e}")
    print(f"[{name}] the reference under torch.autocast(fp16) vs its own fp32: tours identical {r16_same:.4f}, |LL - fp32| mean {r16_dll.mean():.3e} "
          f"max {r16_dll.max():.3e}, best-of-{S} cost gap {r16_gap:+.3e}")
    # the tolerance of this variant: inside the reference's own autocast band, with room to spare (its encoder stays fp32-equivalent)
    assert same_ref >= r16_same
    assert float(dll.mean()) <= 0.5 * float(r16_dll.mean()) and float(dll.max()) <= float(r16_dll.max())
    assert gap <= max(r16_gap, 0.0) + 5e-3
    # ... and a band of its own (the reference's autocast band alone would let a half-broken kernel through): the one-piece
    # rollout keeps most of the DEFAULT build's tours (measured 0.945 random-init, 0.995 trained) and its log-likelihoods stay within a
    # fraction of a nat of the default build's on average (measured 0.087 / 0.016; a dropped operand contribution moves them by whole nats)
    dfull = (half["log_likelihood"] - full["log_likelihood"]).abs()
    assert same_full >= 0.90, same_full
    assert float(dfull.mean()) <= 0.2, float(dfull.mean())
    # the reward it reports is the exact cost of the tours it returns (k_tour_cost is fp32 whatever the policy's arithmetic)
    if "normalized_reward" in half:
        D = restate.atsp_reset(H.fixture_state(fx))["distance_matrix"]
        a = half["actions"]
        b = torch.arange(a.shape[0]) % D.shape[0]
        cost = D[b[:, None], a, a.roll(-1, 1)].sum(1)
        assert torch.allclose(-half["normalized_reward"], cost, atol=1e-4)


# ---- the opt-in 16-mixed TRAINING step: one bf16 piece per operand in the pointer MLP's and the encoder FFN's
# backward products (rr_mlp_rows modes 2 / 3, rr_mlp_wgrad16), fp32 accumulation, fp32 gradients and master weights.  Yardstick:
# tests/golden/*_autocast_grad.npz = the REAL reference's training gradient under torch.autocast against its own fp32 gradient on the
# same tours (oracle/gen_golden.py autocast_grad): 9-12 % of the gradient norm in fp16, 70-87 % in bf16 (it runs every matmul of the
# policy in half precision; the variant here only the two 128-512-128 MLPs' backward).
@pytest.mark.parametrize("name", ["atsp_n20_b4_pomo", "atsp_n100_b2_pomo"])
def test_sixteen_mixed_training_gradient_stays_inside_the_references_autocast_deviation(name):
    from tests.test_gpu_train import _model
    fx = H.load_fixture(name)
    yard = dict(np.load(H.fixture_path(name + "_autocast_grad")))
    w, pol, model, st, td_in = _model(fx)
    grads = {}
    for prec in ("32", "16-mixed"):
        pol.precision = prec
        for p in pol.parameters():
            p.grad = None
        out = model.training_step(td_in.clone(), seed=11)
        torch.cuda.synchronize()
        grads[prec] = ({n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for n, p in pol.named_parameters()},
                       out["actions"].clone(), float(out["loss"]))
    pol.precision = "32"
    (g32, a32, l32), (g16, a16, l16) = grads["32"], grads["16-mixed"]
    assert torch.equal(a32, a16) and l32 == l16                     # the forward (sampling rollout, loss) is the same arithmetic in both
    names = [str(n) for n in yard["names"]]
    assert names == [n for n, _ in pol.named_parameters()]
    tot = sum(float((g ** 2).sum()) for g in g32.values()) ** 0.5
    dev = np.array([float((g16[n] - g32[n]).norm()) for n in names])
    rel = float((dev ** 2).sum()) ** 0.5 / tot
    ref_fp16 = float((yard["dev_fp16"] ** 2).sum()) ** 0.5 / float(yard["grad_norm_fp32"])
    ref_bf16 = float((yard["dev_bf16"] ** 2).sum()) ** 0.5 / float(yard["grad_norm_fp32"])
    print(f"\n[{name}] 16-mixed training gradient: |g16 - g32| / |g32| = {rel:.3e}; the reference under autocast: fp16 {ref_fp16:.3e}, bf16 {ref_bf16:.3e}")
    assert rel > 1e-6, "the 16-mixed step ran the fp32-equivalent kernels"
    assert rel < 0.5 * ref_fp16, (rel, ref_fp16)                    # far inside the reference's own (fp16) autocast deviation
    for i, n in enumerate(names):                                   # ... and no single tensor outside the reference's per-tensor deviation
        assert dev[i] <= 0.5 * float(yard["dev_bf16"][i]) / float(yard["grad_norm_fp32"]) * tot + 2e-3 * tot, (n, dev[i] / tot)


def test_sixteen_mixed_mlp_backward_kernels_against_bf16_rounded_float64():
    """rr_mlp_rows mode 3 and rr_mlp_wgrad16 alone: their products on ONE bf16 piece per operand, fp32 accumulation — compared with
    float64 arithmetic on bf16-ROUNDED operands (what the kernels are meant to compute) and with the two-piece kernels."""
    from rrnco_amd import _lib as L
    from rrnco_amd import packing
    g = torch.Generator().manual_seed(3)
    M, E, FF = 4096, 128, 512
    W1, b1 = torch.randn(FF, E, generator=g) * 0.08, torch.randn(FF, generator=g) * 0.05
    W2, b2 = torch.randn(E, FF, generator=g) * 0.04, torch.randn(E, generator=g) * 0.05
    x, dy = torch.randn(M, E, generator=g), torch.randn(M, E, generator=g) * 0.1
    pk = packing.pack_mlp_train(W1.cuda(), b1.cuda(), W2.cuda(), b2.cuda())
    xc, dyc = x.cuda(), dy.cuda()
    outs = {}
    for mode in (1, 3):
        o = torch.empty(M, E, device="cuda")
        L.check(L.lib().rr_mlp_rows(pk["bwd"], mode, L.ptr(xc), L.ptr(dyc), L.ptr(o), None, 1, M, M, L.stream()), "rr_mlp_rows")
        outs[mode] = o.cpu().double()
    r = lambda t: t.to(torch.bfloat16).double()                                       # noqa: E731
    pre = r(x) @ r(W1).t() + b1.double()
    dpre = r(dy) @ r(W2)
    hx = torch.where(pre > 0, dpre, torch.zeros_like(dpre))
    emu = dy.double() + r(hx.float()) @ r(W1)
    exact = dy.double() + torch.where(x.double() @ W1.double().t() + b1.double() > 0, dy.double() @ W2.double(), torch.zeros_like(dpre)) @ W1.double()
    e_emu = float((outs[3] - emu).norm() / emu.norm())
    e_two = float((outs[1] - exact).norm() / exact.norm())
    e_half = float((outs[3] - exact).norm() / exact.norm())
    print(f"\nrr_mlp_rows: one-piece kernel vs float64 on bf16-rounded operands {e_emu:.2e}; vs exact: two-piece {e_two:.2e}, one-piece {e_half:.2e}")
    assert e_emu < 2e-4                    # (relu-mask flips of pre-activations within fp32 rounding of zero are the residue)
    # against exact arithmetic both forms carry the relu-kink flips (a hidden unit whose pre-activation is within rounding of zero switches
    # its whole gradient column on or off: 6e-4 of the norm here); the one-piece form adds bf16's 2^-9 per operand on top
    assert e_two < 2e-3 and 3e-3 < e_half < 5e-2
    gw = {}
    for half in (False, True):
        dW1, db1_, dW2, db2_ = (torch.zeros(FF, E, device="cuda"), torch.zeros(FF, device="cuda"), torch.zeros(E, FF, device="cuda"), torch.zeros(E, device="cuda"))
        fn = L.lib().rr_mlp_wgrad16 if half else L.lib().rr_mlp_wgrad
        L.check(fn(pk["wgrad"], L.ptr(xc), L.ptr(dyc), L.ptr(dW1), L.ptr(db1_), L.ptr(dW2), L.ptr(db2_), None, 1, M, M, None, L.stream()), "rr_mlp_wgrad")
        gw[half] = (dW1.cpu().double(), dW2.cpu().double())
    h = torch.relu(x.double() @ W1.double().t() + b1.double())
    dh = torch.where(h > 0, dy.double() @ W2.double(), torch.zeros_like(h))
    eW1, eW2 = dh.t() @ x.double(), dy.double().t() @ h
    for half, lo, hi in ((False, 0.0, 2e-3), (True, 1e-3, 5e-2)):
        e1, e2 = float((gw[half][0] - eW1).norm() / eW1.norm()), float((gw[half][1] - eW2).norm() / eW2.norm())
        print(f"rr_mlp_wgrad{'16' if half else ''}: |dW1 - exact| {e1:.2e}, |dW2 - exact| {e2:.2e}")
        assert lo <= e1 < hi and lo <= e2 < hi
