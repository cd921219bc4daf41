"""Run-to-run determinism of the fused rollout.  The greedy decode is a pure function of its inputs: repeated launches must
return bit-identical tours and log-likelihoods.  (A write-after-read hazard between a VALU write and a just-issued
v_mfma_f32_16x16x32_f16 once made the RCVRPTW rollout's P.V differ between runs in the low bits while every parity test
passed most of the time: csrc/rr_common.h, RR_MFMA_SRC_FENCE.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu

REPEATS = 25


def _repeat(pol, env, td_in, S, **kw):
    ref = None
    for i in range(REPEATS):
        out = pol(env.reset(td_in), env, phase="val", decode_type="multistart_greedy", num_starts=S, return_actions=True, fused=True, **kw)
        a, ll, rw = out["actions"], out["log_likelihood"], out["reward"]
        if ref is None:
            ref = (a.clone(), ll.clone(), rw.clone())
            continue
        assert torch.equal(a, ref[0]), f"tours differ in repeat {i}"
        assert torch.equal(ll, ref[1]), f"log-likelihoods differ in repeat {i}: max {float((ll - ref[1]).abs().max()):.3e}"
        assert torch.equal(rw, ref[2])


@pytest.mark.parametrize("which", ["rcvrptw_n20_b4_pomo", "variants"])
def test_rcvrptw_rollout_is_deterministic(which):
    from tests import test_gpu_rcvrptw as T
    fx, w, pol, inst, env, td_in = T._setup(T.VARIANTS if which == "variants" else which)
    _repeat(pol, env, td_in, fx["S"])


def test_rcvrp_rollout_is_deterministic():
    from tests import test_gpu_rcvrp as T
    fx, w, pol, inst, env, td_in = T._setup("rcvrp_n20_b4_pomo")
    _repeat(pol, env, td_in, fx["S"])


@pytest.mark.parametrize("name", ["atsp_n20_b4_pomo", "atsp_n100_b2_pomo"])
def test_atsp_rollout_is_deterministic(name):
    from tests import helpers as H
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    fx = H.load_fixture(name)
    pol = H.make_policy(H.atsp_weights(fx), "atsp")
    env = ATSPEnv(check_solution=False, device=torch.device("cuda"))
    st = H.fixture_state(fx)
    td_in = TensorDict({k: v.cuda() for k, v in st.items()}, batch_size=[st["locs"].shape[0]])
    td_in["sample_idx"] = fx["sample_idx"].cuda()
    _repeat(pol, env, td_in, fx["S"])


def test_full_size_atsp_rollout_is_deterministic():
    """512 instances x 8 augmentations x 100 starts (BASELINE.json configs[1]): every workgroup shape of the headline launch."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from rrnco_amd.envs import ATSPEnv, ATSPGenerator
    dev = torch.device("cuda")
    pol, _ = bench.make_policy(dev)
    env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
    td = ATSPGenerator(num_loc=100, device=dev)(256, generator=torch.Generator(device=dev).manual_seed(3))
    inst = {"locs": td["locs"], "distance_matrix": td["distance_matrix"]}
    outs = []
    for _ in range(3):
        torch.manual_seed(11)
        best, out = bench.hot_path_step(pol, env, inst)
        outs.append((out["actions"].clone(), out["log_likelihood"].clone()))
    for a, ll in outs[1:]:
        assert torch.equal(a, outs[0][0]) and torch.equal(ll, outs[0][1])


@pytest.mark.parametrize("name", ["atsp_n20_b4_pomo", "atsp_n100_b2_pomo"])
def test_encoder_and_decoder_cache_are_deterministic(name):
    """The encoder block kernel and the decoder cache rebuild split operands on the fly as the rollout does (csrc/rr_enc_w.inc,
    rr_gemm_f16.h): repeated launches must return bit-identical embeddings and caches."""
    from tests import helpers as H
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    fx = H.load_fixture(name)
    pol = H.make_policy(H.atsp_weights(fx), "atsp")
    env = ATSPEnv(check_solution=False, device=torch.device("cuda"))
    st = H.fixture_state(fx)
    td_in = TensorDict({k: v.cuda() for k, v in st.items()}, batch_size=[st["locs"].shape[0]])
    td_in["sample_idx"] = fx["sample_idx"].cuda()
    packed = pol.packed(torch.device("cuda"))
    ref = None
    for i in range(REPEATS):
        td = env.reset(td_in)
        row, col = pol.encoder(td, phase="val", packed=packed)
        _, _, cache = pol.decoder.pre_decoder_hook(td, env, (row, col), fx["S"], packed=packed)
        cur = (row.clone(), col.clone(), cache.glimpse_key.clone(), cache.glimpse_val_t.clone(), cache.logit_key.clone())
        if ref is None:
            ref = cur
            continue
        for a, b, what in zip(cur, ref, ("row_emb", "col_emb", "K", "V^T", "L")):
            assert torch.equal(a, b), f"{what} differs in repeat {i}: max {float((a - b).abs().max()):.3e}"


def test_full_size_encoder_is_deterministic():
    """4 096 instance-augmentations (the headline's encoder launches: every workgroup of the chip busy, two waves per SIMD)."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv, ATSPGenerator
    from rrnco_amd.models.encoder import ATSPInitEmbedding
    from rrnco_amd.models.transforms import StateAugmentation
    dev = torch.device("cuda")
    pol, _ = bench.make_policy(dev)
    env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
    inst = ATSPGenerator(num_loc=100, device=dev)(512, generator=torch.Generator(device=dev).manual_seed(5))
    td0 = env.reset(StateAugmentation(augment_fn="dihedral8", no_aug_coords=False)(TensorDict(dict(inst.items()), batch_size=[512])))
    torch.manual_seed(3)
    td0["sample_idx"] = ATSPInitEmbedding.sample_indices(td0["distance_matrix"], 25)
    packed = pol.packed(dev)
    outs = []
    for _ in range(4):
        row, col = pol.encoder(td0, phase="val", packed=packed)
        outs.append((row.clone(), col.clone()))
    for r, c in outs[1:]:
        assert torch.equal(r, outs[0][0]) and torch.equal(c, outs[0][1])


@pytest.mark.parametrize("name", ["atsp_n20_b4_pomo", "atsp_n100_b2_pomo"])
def test_training_step_repeats(name):
    """The sampling rollout with the training dump is a pure function of (weights, instances, seed): tours and log-likelihoods
    bit-identical between repeats.  The backward kernels accumulate with float atomics (order-dependent in the last bits), so the
    gradients are held to 1e-5 of the gradient norm — a stale-operand hazard shows up orders of magnitude above that."""
    from tests import helpers as H
    from tests.test_gpu_train import _model
    fx = H.load_fixture(name)
    ref = None
    for i in range(6):
        w, pol, model, st, td_in = _model(fx)
        out = model.training_step(td_in, seed=11)
        g = torch.cat([p.grad.reshape(-1) for p in pol.parameters() if p.grad is not None]).clone()
        cur = (out["actions"].clone(), out["log_likelihood"].clone(), g)
        if ref is None:
            ref = cur
            continue
        assert torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1]), f"sampled tours / log-likelihoods differ in repeat {i}"
        rel = float((cur[2] - ref[2]).norm() / ref[2].norm())
        assert rel < 1e-5, f"gradients differ by {rel:.3e} of their norm in repeat {i}"


@pytest.mark.parametrize("name", ["atsp_n20_b4_pomo", "atsp_n100_b2_pomo", "rcvrp_n100_b2_pomo"])
def test_encoder_is_identical_across_copies_of_an_instance(name):
    """2 048 copies of ONE instance through the init embedding and the six encoder layers: every copy (one workgroup each in
    k_init_embed, scheduled among different neighbours) must come out bit-identical, call after call.  Two finds (profiles/r06/NOTES.md
    §7, §10): an unfenced schedule of k_init_embed's matrix instructions gave copies different gates (most of 2 048 differ on the
    -DRR_KO_IE_FENCE build), and round 5's k_enc_tail normalised the FIRST instance of a workgroup with per-feature parameters it read
    from LDS in front of the barrier that publishes them — a handful of copies wrong about once per hundred calls, which is why the
    n = 100 cases repeat 40 times."""
    from tests import helpers as H
    from rrnco_amd import TensorDict
    B = 2048
    if name.startswith("atsp"):
        from rrnco_amd.envs import ATSPEnv
        fx = H.load_fixture(name)
        pol = H.make_policy(H.atsp_weights(fx), "atsp")
        env = ATSPEnv(check_solution=False, device=torch.device("cuda"))
        st = H.fixture_state(fx)
    else:
        from tests import test_gpu_rcvrp as T
        fx, w, pol, st, env, _ = T._setup(name)
    one = {k: v[:1].expand(B, *v.shape[1:]).contiguous().cuda() for k, v in st.items()}
    td = TensorDict(one, batch_size=[B])
    td["sample_idx"] = fx["sample_idx"][:1].expand(B, -1, -1).contiguous().cuda()
    packed = pol.packed(torch.device("cuda"))
    for rep in range(10):      # the TRAINING forward (one workgroup per block: k_enc_block_w + k_enc_ffn, every layer's saves written)
        row, col = pol.encoder(env.reset(td[:512]), packed=packed, train_saves=[])
        for t, side in ((row, "row"), (col, "col")):
            same = (t == t[:1]).flatten(1).all(1)
            assert bool(same.all()), f"{name}: {int((~same).sum())} of 512 copies differ in the training forward's {side} embeddings (repeat {rep})"
    for rep in range(40):
        row, col = pol.encoder(env.reset(td), packed=packed)
        for t, side in ((row, "row"), (col, "col")):
            same = (t == t[:1]).flatten(1).all(1)
            assert bool(same.all()), f"{name}: {int((~same).sum())} of {B} copies differ in the {side} embeddings (repeat {rep})"
        # ... and through the decoder cache (k_dec_cache: five products per workgroup, fp32 tensors and the rollout's fp16 images)
        cache = pol.decoder._precompute_cache((row, col), packed=packed)
        parts = [("K", cache.glimpse_key), ("Vt", cache.glimpse_val_t), ("L", cache.logit_key), ("ctxB", cache.ctx_b)]
        parts += [("ctxA", cache.ctx_a)] if cache.ctx_a is not None else []
        parts += [(f"image{i}", t) for i, t in enumerate(cache.split or ())]
        for what, t in parts:
            same = (t.view(torch.int32) == t[:1].view(torch.int32)).flatten(1).all(1)      # (images: bit patterns, some are NaN as floats)
            assert bool(same.all()), f"{name}: {int((~same).sum())} of {B} copies differ in the cache's {what} (repeat {rep})"


@pytest.mark.parametrize("name", ["atsp_n100_b2_pomo", "rcvrp_n100_b2_pomo", "rcvrptw_n100_b2_pomo"])
def test_greedy_rollout_is_identical_across_copies_of_an_instance(name):
    """512 copies of ONE instance through the whole policy (encoder, cache, fused greedy rollout: one workgroup per copy), 10 calls:
    every copy's tours, log-likelihoods and rewards must equal copy 0's, bit for bit — the cross-copy form of the run-to-run tests
    above, the one that exposes a workgroup whose result depends on its start-up timing (profiles/r06/NOTES.md §10)."""
    from tests import helpers as H
    from rrnco_amd import TensorDict
    B = 512
    if name.startswith("atsp"):
        from rrnco_amd.envs import ATSPEnv
        fx = H.load_fixture(name)
        pol = H.make_policy(H.atsp_weights(fx), "atsp")
        env = ATSPEnv(check_solution=False, device=torch.device("cuda"))
        st = H.fixture_state(fx)
    elif name.startswith("rcvrptw"):
        from tests import test_gpu_rcvrptw as T
        fx, w, pol, st, env, _ = T._setup(name)
    else:
        from tests import test_gpu_rcvrp as T
        fx, w, pol, st, env, _ = T._setup(name)
    one = {k: v[:1].expand(B, *v.shape[1:]).contiguous().cuda() for k, v in st.items()}
    td = TensorDict(one, batch_size=[B])
    td["sample_idx"] = fx["sample_idx"][:1].expand(B, -1, -1).contiguous().cuda()
    S = fx["S"]
    for rep in range(10):
        # (per-step log-probabilities: the VRP paths sum them with a torch reduction whose order depends on the row's position — copies
        # differ in the last bit of the SUM there, 1.5e-5 on -200, with identical terms)
        out = pol(env.reset(td), env, phase="val", decode_type="multistart_greedy", num_starts=S, return_actions=True, fused=True,
                  return_sum_log_likelihood=False)
        for what in ("actions", "log_likelihood", "reward"):
            t = out[what].reshape(S, B, -1).transpose(0, 1).contiguous()                # rollout r = s * B + b  ->  [copy][start][...]
            t = t.view(torch.int32) if t.dtype == torch.float32 else t
            same = (t == t[:1]).flatten(1).all(1)
            assert bool(same.all()), f"{name}: {int((~same).sum())} of {B} copies differ in {what} (call {rep})"


def test_big_n_path_is_identical_across_copies_of_an_instance():
    """N = 128 (> 103: the row-parallel kernels of csrc/rr_bign.hip and the step-by-step decode loop): 96 copies of ONE instance, 5 calls —
    embeddings, tours, per-step log-probabilities and rewards of every copy equal copy 0's."""
    from oracle import restate
    from tests import helpers as H
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    N, S, ss, B = 128, 16, 25, 96
    pol = H.make_policy(restate.make_weights(restate.atsp_weight_template(128, 2, 512, ss), 3))
    inst = restate.atsp_synthetic(1, N, 11)
    sidx = restate.sample_neighbor_indices(restate.atsp_reset(inst)["distance_matrix"], ss, generator=torch.Generator().manual_seed(5))
    env = ATSPEnv(generator_params=dict(num_loc=N), check_solution=False)
    td = TensorDict({k: v[:1].expand(B, *v.shape[1:]).contiguous().cuda() for k, v in inst.items()}, batch_size=[B])
    td["sample_idx"] = sidx[:1].expand(B, -1, -1).contiguous().cuda()
    packed = pol.packed(torch.device("cuda"))
    for rep in range(5):
        t = env.reset(td)
        row, col = pol.encoder(t, packed=packed)
        out = pol(t, env, phase="val", decode_type="multistart_greedy", num_starts=S, return_actions=True, return_sum_log_likelihood=False)
        parts = {"row": row, "col": col}
        parts.update({k: out[k].reshape(S, B, -1).transpose(0, 1).contiguous() for k in ("actions", "log_likelihood", "reward")})
        for what, v in parts.items():
            v = v.view(torch.int32) if v.dtype == torch.float32 else v
            same = (v == v[:1]).flatten(1).all(1)
            assert bool(same.all()), f"N = {N}: {int((~same).sum())} of {B} copies differ in {what} (call {rep})"
