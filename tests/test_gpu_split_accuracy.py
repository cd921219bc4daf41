"""`-m gpu`: the accuracy claim behind `dtype: "f32"` as a test: the two-piece fp16 products of
csrc/rr_common.h against float64, beside the fp32 MFMA, for K = 16 / 128 / 512 and operand magnitudes 2^-12 .. 2^12 — compiled from
tests/probes/split_accuracy.hip with hipcc on the GPU box (the probe includes the library's own rr_common.h helpers)."""
import os
import shutil
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_piece_fp16_products_are_as_accurate_as_the_fp32_mfma(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    assert torch.cuda.is_available()
    exe = str(tmp_path / "split_accuracy")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", os.path.join(ROOT, "tests", "probes", "split_accuracy.hip"), "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = []
    for line in out.stdout.splitlines():
        t = line.split()
        if len(t) == 10 and t[0] == "K":
            rows.append((int(t[1]), int(t[3]), float(t[5]), float(t[7]), float(t[9])))
    assert len(rows) == 3 * 9, out.stdout
    print()
    for K, e, s, f, h in rows:
        print(f"  K {K:3d}  |b| ~ 2^{e:+3d}:  two-piece fp16 {s:.2e}   fp32 MFMA {f:.2e}   one fp16 piece {h:.2e}   (max |err| / sum|ab|)")
    for K, e, s, f, h in rows:
        if e >= 0:
            # fp32-level for operands of magnitude ~1 and above (most of their mass >= 2^-3) (what DESIGN.md section 3a promises for "O(1) activations"): within a small
            # factor of the fp32 MFMA's own rounding on the same operands.  (One accumulator takes 3 K / 32 roundings instead of
            # K / 4: the split is often the more accurate of the two.)
            assert s <= 2.5e-7, (K, e, s)
            assert s <= 3.0 * f + 2e-8, (K, e, s, f)
            # a single fp16 piece (the opt-in 16-mixed variant) is three orders of magnitude coarser: the split is not a formality
            assert h >= 30 * s, (K, e, s, h)
        else:
            # UNSCALED operands of magnitude <= 2^-3 (drawn uniformly in [-2^e, 2^e]: most entries below 2^-3): the lo piece is a subnormal fp16 number (below 2^-14 it vanishes), so the error stops
            # being relative: <= 2^-25 absolute per value, i.e. 2^-25 / 2^e relative to sum|ab| when EVERY entry is that small —
            # at 2^-12 no better than one piece.  The kernels pre-scale what is small by construction (weights x 2^6, K / V / L
            # images x 2^4, softmax weights x 2^11); activations are O(1) — measured on the dumped MLP inputs / outputs below.
            assert s <= 2.5e-7 + 2.0 ** -24 / 2.0 ** e, (K, e, s)


def test_the_rollouts_activations_live_where_the_split_is_fp32_accurate():
    """What the two-piece representation costs on the activations the rollout really splits: the pointer MLP's input g0 (the
    glimpse) and output g (the logit query) of every decoder evaluation of a training rollout on TRAINED weights (the dump of
    RolloutIO::dump_g0 / dump_g).  Per value the representation error is 2^-22 |x| for |x| >= 2^-3 and at most 2^-25 below; the
    bound relative to the vector's L1 mass — what enters sum|ab| — must stay at the fp32 level (2^-21)."""
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    from tests import helpers as H
    fx = H.load_fixture("atsp_n100_b2_pomo_trained")
    pol = H.make_policy(H.atsp_weights(fx), device="cuda:0").train()
    env = ATSPEnv(generator_params=dict(num_loc=fx["N"]))
    td = TensorDict({"locs": fx["locs"].cuda(), "distance_matrix": fx["distance_matrix"].cuda(), "sample_idx": fx["sample_idx"].cuda()},
                    batch_size=[fx["B"]])
    cap = {}
    with torch.no_grad():
        pol._forward_impl(env.reset(td), env, phase="train", decode_type="multistart_sampling", num_starts=fx["S"], capture=cap, seed=5)
    print()
    for name in ("g0", "g"):
        x = cap["dump"][name].double().abs()
        rep = torch.where(x >= 2.0 ** -3, x * 2.0 ** -22, torch.full_like(x, 2.0 ** -25)).sum(1) / x.sum(1).clamp_min(1e-30)      # per row
        small = float((x < 2.0 ** -3).double().mean())
        print(f"  {name}: rms {float(x.pow(2).mean().sqrt()):.3f}, median |x| {float(x.median()):.3f}, entries below 2^-3: {100 * small:.1f} %; "
              f"representation error / L1 mass: mean {float(rep.mean()):.2e}, worst row {float(rep.max()):.2e} (2^-22 = 2.4e-07)")
        assert float(rep.max()) <= 2.0 ** -21, name
