"""Benchmark of the hot path: solved instances/sec, ATSP n=100, B=512 per GPU, POMO (S=100 starts x 8 dihedral
augmentations) greedy — BASELINE.json configs[1].

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = reset -> x8 augmentation -> policy (encoder + persistent rollout) -> get_reward -> best-of-(aug,start)
over one batch already resident in HBM.  Instances shard over ranks with no data-path collective (weak scaling).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "real-routing-nco_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

N_NODES, BATCH, STARTS, AUG = 100, 512, 100, 8
FLOP_PER_ROLLOUT_STEP = 404_480          # pointer step K6-K7, per rollout per decode step
# what the matrix pipe executes per rollout-step: 2 720 v_mfma_f32_16x16x4_f32 x 2 048 flop x 16 rollouts per tile / 16 =
# 348 160 (keys padded 100 -> 112; the 65 536-flop context projection is two table gathers, not a GEMM: DESIGN.md §3)
EXECUTED_FLOP_PER_ROLLOUT_STEP = 348_160
PEAK_F32_MFMA_TFLOPS = 157.3             # MI355X_MICROARCH.md: fp32 matrix peak
PEAK_F16_MFMA_TFLOPS = 2500.0            # ... dense fp16 / bf16 matrix peak (measured 2 360-2 380 on both: tools/clockprobe/f16probe.hip)
# The default rollout runs every product of a rollout-step (attention scores, P.V, pointer MLP, logits) on the fp16 matrix pipe
# with each fp32 operand split in two fp16 pieces x = hi + 2^-11 lo' and three partial products kept (hi*hi, hi*lo', lo'*hi; fp32
# accumulate; measured error 4e-8 of sum |a b|, below the fp32 MFMA's own): 3 x 404 480 fp16 flop per rollout-step.  Its MFMA
# roofline is the time the fp16 pipe needs for that at its dense peak: 2 500 / 3 = 833 TFLOP/s of fp32-equivalent work.
# What the pipe executes per 16-rollout tile and decode step: 768 (MLP) + 3 x 112 (scores, P.V, logits: the k = 16 products ride
# in k = 32 instructions as [hi | lo'] x [hi | 0] and [hi | lo'] x [lo' | hi]) = 1 104 v_mfma_f32_16x16x32_f16 of 16 384 flop.
SPLIT_PRODUCTS = 3
EXECUTED_F16_FLOP_PER_TILE_STEP = 1104 * 16384
EXECUTED_F16_FLOP_PER_ROLLOUT_STEP = EXECUTED_F16_FLOP_PER_TILE_STEP // 16      # (a full tile; tools/bench_train.py)
ROLLOUT_KERNEL = "k_rollout_w<7, 0, 0, true, true>"
ROLLOUT_KERNEL_FP32 = "k_rollout_w<7, 0, 0, false, false>"
# HBM-side traffic of ONE rollout launch at the default workload: rocprofv3 PMC, separate FETCH_SIZE / WRITE_SIZE passes,
# (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B — FETCH_SIZE doubled for gfx950's 16-B/lane reads as MI355X_MICROARCH.md §HBM
# prescribes; Infinity-Cache hits are included in the counter.  Read at run time from the committed summary
# (tools/pmc_traffic.sh writes it together with a hash of the rollout's sources): a summary of other sources -> null.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r02", "bench_pmc_hbm_traffic.json")
ROLLOUT_SOURCES = ("rr_decode.hip", "rr_rollout_w.inc", "rr_common.h")


def rollout_source_hash():
    import hashlib
    h = hashlib.sha256()
    for f in ROLLOUT_SOURCES:
        with open(os.path.join(ROOT, "real-routing-nco_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_rollout_traffic(batch):
    """-> (bytes per launch or None, provenance string)."""
    try:
        with open(TRAFFIC_FILE) as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        return None, "no PMC summary committed"
    if rec.get("source_hash") != rollout_source_hash() or rec.get("batch") != batch:
        return None, f"PMC summary is of other rollout sources / batch ({rec.get('source_hash')}, batch {rec.get('batch')})"
    return (2 * rec["FETCH_SIZE_KB"] + rec["WRITE_SIZE_KB"]) * 1024.0, os.path.relpath(TRAFFIC_FILE, ROOT)


def make_policy(device, seed=1234):
    """Random-init RRNet of configs/experiment/rrnet.yaml (torch's default layer initialisation under a fixed seed; no
    checkpoint can be fetched here).  Returns the policy and a CPU copy of its weights — the latter only feeds the
    cpu_baseline leg, so that the oracle times the very same network."""
    from rrnco_amd.models import RRNetPolicy
    torch.manual_seed(seed)
    pol = RRNetPolicy(env_name="atsp", embed_dim=128, num_heads=8, num_encoder_layers=6, normalization="instance",
                      use_graph_context=False, nab_type="gating",
                      init_embedding_kwargs=dict(use_coords=True, use_polar_feats=True, use_dist=True,
                                                 use_matnet_init=False, sample_type="prob", sample_size=25))
    w = {k: v.detach().clone() for k, v in pol.state_dict().items()}
    return pol.to(device).eval(), w


def hot_path_step(pol, env, inst):
    """test.py:188-213 shaped: augment -> reset -> policy -> reward -> max over starts, then over augs.  The encoder's
    neighbour sample is drawn INSIDE the step, per forward and over all 8 x B instance-augmentations, as the reference does
    (rrnco/models/env_embeddings/atsp.py:55-67) — on the device (csrc/rr_sample.hip)."""
    from rrnco_amd import TensorDict
    from rrnco_amd.models.transforms import StateAugmentation
    from rrnco_amd.ops import unbatchify
    td = TensorDict(dict(inst), batch_size=[inst["locs"].shape[0]])
    td = StateAugmentation(augment_fn="dihedral8", no_aug_coords=False)(td)
    td = env.reset(td)
    out = pol(td, env, phase="val", decode_type="multistart_greedy", num_starts=STARTS, return_actions=True)
    rew = unbatchify(out["reward"], (AUG, STARTS))          # [B, A, S]
    best = rew.max(dim=-1).values.max(dim=-1).values
    return best, out


def cpu_baseline(w, seed=4321, budget_s=20.0, max_inst=8):
    """The oracle (op-for-op CPU restatement of the reference, fp32) timed on the host cores on a BOUNDED
    sample of the same workload: instances x 8 augmentations x 100 starts, one instance (8 instance-augs)
    per micro-batch, as many instances as fit in ~budget_s."""
    from oracle import restate
    threads = min(os.cpu_count() or 1, 32)     # torch-CPU scales poorly past a few dozen threads on these ops
    torch.set_num_threads(threads)
    done, spent = 0, 0.0
    while done < max_inst and (done == 0 or spent + spent / done < budget_s):
        inst = restate.atsp_synthetic(1, N_NODES, seed + done)
        t0 = time.perf_counter()
        with torch.inference_mode():
            st = restate.atsp_reset(restate.augment_state(inst))
            sidx = restate.sample_neighbor_indices(st["distance_matrix"], 25)
            restate.atsp_policy(w, st, sidx, STARTS, "greedy")
        spent += time.perf_counter() - t0
        done += 1
    return {"value": done / spent, "unit": "instances/s", "cores": threads, "kind": "port",
            "sample": f"{done} ATSP n=100 instance(s) x8 aug x100 starts greedy, torch-CPU fp32 oracle, {spent:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dist = world > 1
    # RR_DIST_BACKEND=gloo lets the multi-process path be exercised on a box with fewer GPUs than ranks (ranks share devices)
    backend = os.environ.get("RR_DIST_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1) if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist:
        import torch.distributed as td_
        if backend == "nccl":
            td_.init_process_group("nccl", device_id=dev)
        else:
            td_.init_process_group(backend)

    from rrnco_amd.envs import ATSPEnv, ATSPGenerator
    from rrnco_amd.models import rollout as R
    pol, w = make_policy(dev)
    env = ATSPEnv(generator_params=dict(num_loc=N_NODES, device=dev), check_solution=False, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    inst_td = ATSPGenerator(num_loc=N_NODES, device=dev)(args.batch, generator=gen)
    inst = {"locs": inst_td["locs"], "distance_matrix": inst_td["distance_matrix"]}
    torch.manual_seed(4242 + rank)           # the neighbour samples of the timed steps come from this stream

    def sync_all():
        torch.cuda.synchronize()
        if dist:
            td_.barrier()
            torch.cuda.synchronize()

    # headline = the default build (rollout pointer MLP and encoder FFN on split-bf16 operands, everything else fp32 MFMA);
    # the all-fp32-MFMA build is measured separately below
    os.environ.pop("RR_MLP_SPLIT", None)
    R.SPLIT_MLP = True
    for _ in range(args.warmup):
        hot_path_step(pol, env, inst)
    R.TIMING = []
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        torch.manual_seed(4242 + rank + 1000 * k)      # the step's neighbour sample (the variant below replays the same draws)
        best, out = hot_path_step(pol, env, inst)
    sync_all()
    dt = time.perf_counter() - t0
    kern_ms = [a.elapsed_time(b) for a, b in R.TIMING]
    R.TIMING = None
    from rrnco_amd.parallel import aggregate_throughput
    total_inst, dt = aggregate_throughput(args.batch * args.steps, dt, dist, dev if backend == "nccl" else torch.device("cpu"))

    if rank == 0:
        rollout_steps = args.batch * AUG * STARTS * (N_NODES - 1)
        k_ms = sum(kern_ms) / max(len(kern_ms), 1)
        achieved = rollout_steps * FLOP_PER_ROLLOUT_STEP / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        peak_split = PEAK_F16_MFMA_TFLOPS / SPLIT_PRODUCTS
        # executed work counts tiles: S = 100 starts are 6 full 16-rollout tiles and one quarter-full one per instance (tail tiles are
        # not packed across instances: csrc/rr_decode.hip)
        tile_steps = args.batch * AUG * ((STARTS + 15) // 16) * (N_NODES - 1)
        executed = tile_steps * EXECUTED_F16_FLOP_PER_TILE_STEP / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        traffic, traffic_src = measured_rollout_traffic(args.batch)
        line = {
            "metric": "solved instances/sec (ATSP n=100, B=512, POMO greedy)", "value": total_inst / dt,
            "unit": "instances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "arithmetic": "fp32 operands and fp32 accumulation throughout; the rollout's products (scores, P.V, pointer MLP, logits) "
                          "multiply two-piece fp16 splits x = hi + 2^-11 lo' on the fp16 matrix pipe (3 partial products; error of a dot "
                          "product 4e-8 of sum|ab| against 1.1e-7 for the fp32 MFMA, tools/clockprobe/f16probe.hip), the encoder FFN "
                          "3-way bf16 splits (6 products); tours equal to the all-fp32-MFMA build's (tests/test_gpu_fullsize.py), which "
                          "is timed under `variants`",
            "config": {"workload": f"ATSP n={N_NODES}, batch={args.batch}/GPU, POMO S={STARTS} starts x {AUG} dihedral aug, greedy "
                                   "(BASELINE.json configs[1]); random-init RRNet E=128 L=6",
                       "rollouts_per_gpu": args.batch * AUG * STARTS, "sharding": f"instances over {world} rank(s), no collective"},
            "roofline": {"bound": "mfma", "kernel": ROLLOUT_KERNEL + " (persistent wave-autonomous POMO decode)", "achieved": achieved,
                         "peak": peak_split, "unit": "TFLOP/s", "frac": achieved / peak_split,
                         "peak_note": "fp32-equivalent flop/s of the fp16 matrix pipe at its dense peak with 3 partial products per product "
                                      "(2 500 / 3); round 1's all-fp32-MFMA kernel (peak 157.3) is under `variants`",
                         "executed_mfma": {"achieved": executed, "peak": PEAK_F16_MFMA_TFLOPS, "frac": executed / PEAK_F16_MFMA_TFLOPS,
                                           "note": "1 104 v_mfma_f32_16x16x32_f16 per 16-rollout tile and decode step, 7 tiles per 100 starts"},
                         "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": k_ms,
                         "algorithmic_flop_per_launch": rollout_steps * FLOP_PER_ROLLOUT_STEP},
            "mean_best_cost": float(-best.mean().item()),
        }
        if world == 1:
            def timed(label):
                hot_path_step(pol, env, inst)                          # (warm-up of the variant's kernels / packs)
                R.TIMING = []
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for k in range(args.steps):
                    torch.manual_seed(4242 + rank + 1000 * k)
                    best_v, _ = hot_path_step(pol, env, inst)
                torch.cuda.synchronize()
                dtv = time.perf_counter() - t1
                ks = [a.elapsed_time(b) for a, b in R.TIMING]
                R.TIMING = None
                kv = sum(ks) / max(len(ks), 1)
                return {"value": args.batch * args.steps / dtv, "unit": "instances/s", "ms_per_step": dtv / args.steps * 1e3,
                        "kernel_ms": kv, "mean_best_cost": float(-best_v.mean().item()),
                        "instances_with_identical_best_cost": float((best_v == best).float().mean().item())}, kv
            line["variants"] = {}
            # the all-fp32-MFMA build (round 1's default): priced against the fp32 matrix peak, algorithmic and executed flop
            R.SPLIT_MLP = False
            os.environ["RR_MLP_SPLIT"] = "0"
            v32, k32 = timed("fp32")
            a32 = rollout_steps * FLOP_PER_ROLLOUT_STEP / (k32 * 1e-3) / 1e12
            e32 = tile_steps * 16 * EXECUTED_FLOP_PER_ROLLOUT_STEP / (k32 * 1e-3) / 1e12
            v32["roofline"] = {"bound": "mfma", "kernel": ROLLOUT_KERNEL_FP32, "achieved": a32, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": a32 / PEAK_F32_MFMA_TFLOPS,
                               "executed_mfma": {"achieved": e32, "frac": e32 / PEAK_F32_MFMA_TFLOPS,
                                                 "note": "348 160 flop per rollout-step actually issued (no context GEMM, keys padded to 112), 7 tiles of 16 per 100 starts"}}
            line["variants"]["all_fp32_mfma (RR_MLP_SPLIT=0)"] = v32
            R.SPLIT_MLP = True
            os.environ.pop("RR_MLP_SPLIT")
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w)
        print(json.dumps(line))
    if dist:
        td_.destroy_process_group()


if __name__ == "__main__":
    main()
