"""Range guard of the fp16 two-piece kernels (csrc/rr_common.h, models/policy.py:_forward_impl): operands that leave the fp16
range must not turn into silent garbage — the status word is raised and the call is repeated on the fp32-MFMA kernels, whose
result still equals the reference's.  The reference has no such limit (fp32 throughout: rrnco/models/decoder.py:281-323)."""
import warnings

import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _run(pol, fx, mode="sync"):
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    env = ATSPEnv(generator_params=dict(num_loc=fx["N"]))
    td = TensorDict({"locs": fx["locs"].cuda(), "distance_matrix": fx["distance_matrix"].cuda(),
                     "sample_idx": fx["sample_idx"].cuda()}, batch_size=[fx["B"]])
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        out = pol(env.reset(td), env, phase="val", decode_type="multistart_greedy", num_starts=fx["S"], return_actions=True,
                  range_guard=mode)
    return out, [str(w.message) for w in rec]


def _scaled_mlp(w, c):
    """relu is positively homogeneous: (c W1, c b1, W2 / c) is the same pointer MLP; c a power of two keeps fp32 results bit-equal."""
    w = dict(w)
    w["decoder.pointer.ffn.lins.0.weight"] = w["decoder.pointer.ffn.lins.0.weight"] * c
    w["decoder.pointer.ffn.lins.0.bias"] = w["decoder.pointer.ffn.lins.0.bias"] * c
    w["decoder.pointer.ffn.lins.1.weight"] = w["decoder.pointer.ffn.lins.1.weight"] / c
    return w


@pytest.mark.parametrize("name", ["atsp_n20_b4_pomo", "atsp_n100_b2_pomo"])
def test_in_range_call_raises_nothing_and_hidden_activations_beyond_fp16_fall_back_to_fp32(name):
    fx = H.load_fixture(name)
    w = H.atsp_weights(fx)
    pol = H.make_policy(w, device="cuda:0")
    out, msgs = _run(pol, fx)
    assert pol.last_range_flags == 0 and not any("fp16 range" in m for m in msgs)
    base = out["actions"].cpu()
    # hidden pre-activations x 2^11: the weight images stay in range (|2^6 W1| < 65504), relu(H) does not -> inf -> NaN logits
    pol2 = H.make_policy(_scaled_mlp(w, 2048.0), device="cuda:0")
    out2, msgs2 = _run(pol2, fx)
    assert pol2.last_range_flags & 4, f"guard did not fire: flags {pol2.last_range_flags}"
    assert any("fp16 range" in m for m in msgs2)
    a2 = out2["actions"].cpu()
    assert torch.isfinite(out2["log_likelihood"]).all() and torch.isfinite(out2["reward"]).all()
    same_ref = (a2 == fx["actions"]).all(1).float().mean().item()
    same_base = (a2 == base).all(1).float().mean().item()
    print(f"[{name}] scaled MLP: flags {pol2.last_range_flags:#x}; tours equal to the reference {same_ref:.4f}, to the unscaled split run {same_base:.4f}")
    assert same_ref >= (1.0 if fx["N"] <= 20 else 0.99)
    assert (out2["reward"].cpu() - fx["reward"]).abs().max().item() < 1e-4 or same_ref < 1.0


def test_weight_image_out_of_range_is_caught_at_pack_time():
    fx = H.load_fixture("atsp_n20_b4_pomo")
    pol = H.make_policy(_scaled_mlp(H.atsp_weights(fx), 16384.0), device="cuda:0")      # |2^6 W1| ~ 1e5
    out, msgs = _run(pol, fx)
    assert pol.last_range_flags & 2 and any("fp16 range" in m for m in msgs)
    assert (out["actions"].cpu() == fx["actions"]).all()


def test_pack_kernel_flags_out_of_range_and_non_finite_values():
    from rrnco_amd import _lib as L
    for bad, expect in ((None, 0), (5000.0, 1), (float("nan"), 1), (float("inf"), 1), (-4095.0, 1), (4000.0, 0)):
        x = torch.randn(2, 100, 128, device="cuda")
        if bad is not None:
            x[1, 37, 5] = bad
        d = torch.empty_like(x)
        st = torch.zeros(1, dtype=torch.int32, device="cuda")
        L.check(L.lib().rr_pack_f16x2(L.ptr(x), L.ptr(d), x.numel(), L.ptr(st), L.stream()), "rr_pack_f16x2")
        assert int(st.item()) == expect, (bad, int(st.item()))
        if expect == 0:       # the image reproduces 2^4 x to 2^-22
            h = d.view(torch.float16).view(-1, 8).float()
            rec = (h[:, :4] + h[:, 4:]) / 16.0
            assert ((rec - x.view(-1, 4)).abs() <= 2.0 ** -21 * x.view(-1, 4).abs() + 2.0 ** -28).all()


def test_deferred_mode_marks_the_log_likelihood_and_raises_at_the_next_call_then_runs_fp32():
    """The default for ATSP inference: no host read in the call; the offending call's log-likelihood is NaN, the next call (or
    check_range) raises, later calls run on the fp32 kernels and match the reference."""
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    fx = H.load_fixture("atsp_n20_b4_pomo")
    pol = H.make_policy(_scaled_mlp(H.atsp_weights(fx), 2048.0), device="cuda:0")
    env = ATSPEnv(generator_params=dict(num_loc=fx["N"]), check_solution=False)

    def call():
        td = TensorDict({"locs": fx["locs"].cuda(), "distance_matrix": fx["distance_matrix"].cuda(),
                         "sample_idx": fx["sample_idx"].cuda()}, batch_size=[fx["B"]])
        return pol(env.reset(td), env, phase="val", decode_type="multistart_greedy", num_starts=fx["S"], return_actions=True)
    out = call()                                               # auto -> deferred: returns without a host read
    assert torch.isnan(out["log_likelihood"]).all()
    assert torch.isnan(out["reward"]).all()                    # a caller that only reads the rewards (test.py:204-213) cannot miss it
    with pytest.raises(FloatingPointError):
        pol.check_range()
    out2 = call()                                              # sticky fp32 from now on
    assert (out2["actions"].cpu() == fx["actions"]).all() and torch.isfinite(out2["log_likelihood"]).all()
    pol3 = H.make_policy(_scaled_mlp(H.atsp_weights(fx), 2048.0), device="cuda:0")
    pol, pol_old = pol3, pol
    call()
    torch.cuda.synchronize()                                   # (calls only POLL earlier words: one that has not finished yet is seen by a later call)
    with pytest.raises(FloatingPointError):
        call()                                                 # ... or a later call raises, once the flagged one has finished


def test_deferred_calls_never_read_the_device_and_new_weights_drop_old_words():
    """VERDICT r04 #5 / ADVICE r04: a steady-state ATSP inference call dispatches no host read (round 4 read the previous call's word at
    its entry and so ran the host in lock-step with the device); the word travels through a pinned host word + event and is polled.
    A word raised under OLD weights must not throw at the first call with new ones."""
    from torch.utils._python_dispatch import TorchDispatchMode
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    fx = H.load_fixture("atsp_n100_b2_pomo")
    w = H.atsp_weights(fx)
    pol = H.make_policy(w, device="cuda:0")
    env = ATSPEnv(generator_params=dict(num_loc=fx["N"]), check_solution=False)

    def call(p):
        td = TensorDict({"locs": fx["locs"].cuda(), "distance_matrix": fx["distance_matrix"].cuda(),
                         "sample_idx": fx["sample_idx"].cuda()}, batch_size=[fx["B"]])
        return p(env.reset(td), env, phase="val", decode_type="multistart_greedy", num_starts=fx["S"], return_actions=True)
    for _ in range(3):
        call(pol)
    reads = []

    class Spy(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if ("_local_scalar_dense" in name or "item" in name) and args and torch.is_tensor(args[0]) and args[0].is_cuda:
                reads.append(name)                             # (a scalar read of a DEVICE tensor: the host waits for the stream)
            if "_to_copy" in name and (kwargs or {}).get("device", None) is not None and str((kwargs or {})["device"]).startswith("cpu") \
                    and not (kwargs or {}).get("non_blocking", False):
                reads.append(name)
            return func(*args, **(kwargs or {}))
    with Spy():
        for _ in range(4):
            out = call(pol)
    assert not reads, f"host reads inside deferred ATSP calls: {reads}"
    assert len(pol._range_pending or []) <= 8
    pol.check_range()
    assert pol.last_range_flags == 0 and not pol._range_pending
    assert (out["actions"].cpu() == fx["actions"]).all(1).float().mean() >= 0.99
    # a raised word of the old weights dies with them
    bad = H.make_policy(_scaled_mlp(w, 2048.0), device="cuda:0")
    o = call(bad)
    assert torch.isnan(o["reward"]).all() and bad._range_pending
    torch.cuda.synchronize()
    bad.load_state_dict({k: v.cuda() for k, v in w.items()}, strict=False)
    assert not bad._range_pending
    c = call(bad)                                              # does not raise
    bad.check_range()
    assert torch.isfinite(c["reward"]).all()


def test_deferred_mode_leaves_clean_calls_untouched_and_sticky_fp32_ends_with_new_weights():
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    fx = H.load_fixture("atsp_n20_b4_pomo")
    w = H.atsp_weights(fx)
    pol = H.make_policy(w, device="cuda:0")
    env = ATSPEnv(generator_params=dict(num_loc=fx["N"]), check_solution=False)

    def call(p, **kw):
        td = TensorDict({"locs": fx["locs"].cuda(), "distance_matrix": fx["distance_matrix"].cuda(),
                         "sample_idx": fx["sample_idx"].cuda()}, batch_size=[fx["B"]])
        return p(env.reset(td), env, phase="val", decode_type="multistart_greedy", num_starts=fx["S"], return_actions=True, **kw)
    a = call(pol, range_guard="deferred")
    b = call(pol, range_guard="sync")
    assert torch.equal(a["reward"], b["reward"]) and torch.equal(a["log_likelihood"], b["log_likelihood"])      # + 0.0 changes nothing
    pol.check_range()
    # several pending words are all read (one host read); a raised one among them is reported
    bad = H.make_policy(_scaled_mlp(w, 2048.0), device="cuda:0")
    call(bad, range_guard="deferred")
    assert len(bad._range_pending) == 1
    with pytest.raises(FloatingPointError):
        bad.check_range()
    assert bad._range_sticky_fp32
    # new weights end the fp32 exile (the flag used to stay for the life of the policy)
    bad.load_state_dict({k: v.cuda() for k, v in w.items()}, strict=False)
    bad.invalidate_pack()
    assert not bad._range_sticky_fp32
    c = call(bad, range_guard="sync")
    assert bad.last_range_flags == 0 and (c["actions"].cpu() == fx["actions"]).all()


def test_stepwise_decode_path_reads_the_guard_word_too():
    """fused=False never launched the fused rollout, so nobody read the word there: a weight image out of range
    (bit 1, set at pack time) must repeat the step-wise call on the fp32 kernels as well."""
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    fx = H.load_fixture("atsp_n20_b4_pomo")
    pol = H.make_policy(_scaled_mlp(H.atsp_weights(fx), 16384.0), device="cuda:0")
    env = ATSPEnv(generator_params=dict(num_loc=fx["N"]), check_solution=False)
    td = TensorDict({"locs": fx["locs"].cuda(), "distance_matrix": fx["distance_matrix"].cuda(),
                     "sample_idx": fx["sample_idx"].cuda()}, batch_size=[fx["B"]])
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        out = pol(env.reset(td), env, phase="val", decode_type="multistart_greedy", num_starts=fx["S"], return_actions=True,
                  range_guard="sync", fused=False)
    assert pol.last_range_flags & 2 and any("fp16 range" in str(m.message) for m in rec)
    assert (out["actions"].cpu() == fx["actions"]).all()


def test_node_softmax_overflow_in_the_recut_encoder_is_flagged_and_repeated_in_fp32():
    """the re-cut encoder layer emits exp(K - mean over the nodes K) (the reference's softmax subtracts the maximum,
    attn_freenet.py:319-321): a K more than ~88 above its node mean overflows.  With the policy's range guard rr_enc_layer_split gets the
    status word: the exponent is clamped (finite results) and bit 0 raised — the call repeats on the fp32 kernels and ends with finite
    rewards; without a guard (a direct encoder call) the embeddings are not finite, loudly."""
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    fx = H.load_fixture("atsp_n100_b2_pomo")
    w = dict(H.atsp_weights(fx))
    k = "encoder.net.layers.0.row_encoding_block.attn_free.to_k.weight"
    w[k] = w[k] * 400.0                                     # K of the first row block: hundreds above its node mean somewhere
    pol = H.make_policy(w, device="cuda:0")
    out, msgs = _run(pol, fx)
    assert pol.last_range_flags & 1, f"guard did not fire: flags {pol.last_range_flags}"
    assert any("fp16 range" in m for m in msgs)
    assert torch.isfinite(out["reward"]).all() and torch.isfinite(out["log_likelihood"]).all()
    # no guard: nothing hides the overflow
    env = ATSPEnv(generator_params=dict(num_loc=fx["N"]), check_solution=False)
    st = {k2: v.cuda() for k2, v in H.fixture_state(fx).items()}
    td = TensorDict(st, batch_size=[fx["B"]])
    td["sample_idx"] = fx["sample_idx"].cuda()
    row, col = pol.encoder(env.reset(td), packed=pol.packed(torch.device("cuda")))
    assert not bool(torch.isfinite(row).all() and torch.isfinite(col).all())
