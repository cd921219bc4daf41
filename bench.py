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
# with each fp32 operand split in two fp16 pieces x~ = hi + lo (csrc/rr_common.h, second form) and three partial products kept
# (hi*hi, hi*lo, lo*hi; fp32 accumulate; measured error of a dot product 8e-8 of sum |a b|, at the fp32 MFMA's own level): 3 x
# 404 480 fp16 flop per rollout-step.  Its MFMA roofline is the time the fp16 pipe needs for that at its dense peak: 2 500 / 3 =
# 833 TFLOP/s of fp32-equivalent work.  What the pipe executes per 16-rollout tile and decode step: 768 (MLP) + 3 x 112 (scores,
# P.V, logits: a k = 16 product is one k = 16 instruction on the hi halves + one k = 32 instruction [hi | lo] x [lo | hi], both
# 16 cycles) = 1 104 matrix instructions of 16 384 flop-slots.
SPLIT_PRODUCTS = 3
EXECUTED_F16_FLOP_PER_TILE_STEP = 1104 * 16384
EXECUTED_F16_FLOP_PER_ROLLOUT_STEP = EXECUTED_F16_FLOP_PER_TILE_STEP // 16      # (a full tile; tools/bench_train.py)
ROLLOUT_KERNEL = "k_rollout_w<7, 0, 0, true, true, false, false>"
ROLLOUT_KERNEL_FP32 = "k_rollout_w<7, 0, 0, false, false, false, false>"
# HBM-side traffic of ONE rollout launch at the default workload: rocprofv3 PMC, separate FETCH_SIZE / WRITE_SIZE passes,
# (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B — FETCH_SIZE doubled for gfx950's 16-B/lane reads as MI355X_MICROARCH.md §HBM
# prescribes; Infinity-Cache hits are included in the counter.  Read at run time from the committed summary
# (tools/pmc_traffic.sh writes it together with a hash of the rollout's sources): a summary of other sources -> null.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06", "bench_pmc_hbm_traffic.json")
ROLLOUT_SOURCES = ("rr_decode.hip", "rr_rollout_w.inc", "rr_common.h")


def instance_seed(batch_index: int, rank: int, scaling: str) -> int:
    """Seed of the generator that draws instance batch `batch_index`: weak scaling = per rank (every rank solves its own instances),
    strong = one global batch that parallel.shard_range partitions (same seed on every rank)."""
    return 1234 + 7919 * batch_index + (rank if scaling == "weak" else 0)


def sample_seed(rank: int, step: int) -> int:
    """torch.manual_seed of timed step `step` on `rank` (the encoder's neighbour sample, atsp.py:55-67); step -1: the warm-up stream."""
    return 4242 + rank + 1000 * step


def rollout_source_hash():
    import hashlib
    h = hashlib.sha256()
    for f in ROLLOUT_SOURCES:
        with open(os.path.join(ROOT, "real-routing-nco_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


ENCODER_SOURCES = ("rr_encoder.hip", "rr_enc_w.inc", "rr_enc_split.inc", "rr_common.h", "rr_gemm_f16.h")
# one encoder layer at the headline shape = these three launches (csrc/rr_enc_split.inc; rounds 2-4: k_enc_block_w<7, true, false> + k_enc_ffn<7>)
ENCODER_LAYER_KERNELS = ("k_nab_dist_family", "k_enc_kv", "k_enc_mix<7, true>", "k_enc_tail<7>")


def encoder_source_hash():
    import hashlib
    h = hashlib.sha256()
    for f in ENCODER_SOURCES:
        with open(os.path.join(ROOT, "real-routing-nco_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_encoder_traffic(batch):
    """HBM-side bytes of one encoder layer (block kernel + FFN kernel) from the same PMC summary, or None."""
    try:
        with open(TRAFFIC_FILE) as fh:
            rec = json.load(fh).get("encoder")
    except (OSError, ValueError):
        return None
    if not rec or rec.get("source_hash") != encoder_source_hash() or rec.get("batch") != batch:
        return None
    return sum((2 * k["FETCH_SIZE_KB"] + k["WRITE_SIZE_KB"]) * 1024.0 for k in rec["kernels"].values())


def library_source_hash():
    """Hash of every kernel source of the library (the PMC summaries of configs[2..4]'s kernels are keyed by it)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "real-routing-nco_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "real-routing-nco_amd", "csrc", "*.inc"))
                    + glob.glob(os.path.join(ROOT, "real-routing-nco_amd", "csrc", "*.h"))):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_other_traffic(kernel):
    """HBM-side bytes per launch of `kernel` (a configs[2..4] dominant kernel) from tools/pmc_traffic_others.sh's summary, or None when
    the committed summary is of other sources."""
    try:
        with open(TRAFFIC_FILE) as fh:
            rec = json.load(fh).get("others")
    except (OSError, ValueError):
        return None
    if not rec or rec.get("library_source_hash") != library_source_hash():
        return None
    k = rec.get("kernels", {}).get(kernel)
    return k.get("bytes_per_launch") if k else None


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


def measured_mfma_busy(kernel, batch, encoder=False):
    """Matrix-pipe busy fraction of `kernel` from the committed PMC counter pass (tools/pmc_traffic_json.py: SQ_VALU_MFMA_BUSY_CYCLES /
    1 024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs), or None when the summary is of other sources."""
    try:
        with open(TRAFFIC_FILE) as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        return None
    want = encoder_source_hash() if encoder else rollout_source_hash()
    c = rec.get("counters", {})
    if c.get("encoder_source_hash" if encoder else "rollout_source_hash") != want or rec.get("batch") != batch:
        return None
    k = c.get("kernels", {}).get(kernel)
    return k.get("mfma_busy") if k else None


def power_limited_mfma(seconds=1.5):
    """What the fp16 matrix pipe sustains ON DATA under the board's power cap: tools/clockprobe/powerprobe.hip (register-resident
    v_mfma_f32_16x16x32_f16 loops on 16 rotating pseudo-random operand sets per wave, nothing else on the chip) run as a child process
    for `seconds`, socket power and clock sampled beside it (amdsmi).  The guide's 2 500 TFLOP/s is measured with constant operands
    (2 417 here at 756 W); on data the same loop draws the full cap and the clock drops (profiles/r04/NOTES.md section 1).  Reported
    beside `roofline.peak`, never instead of it.  -> dict or None."""
    import subprocess, threading
    exe = os.path.join(ROOT, "tools", "clockprobe", "powerprobe")
    try:
        if not os.path.exists(exe):
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", exe + ".hip", "-o", exe], check=True, capture_output=True, timeout=300)
        samples, stop = [], threading.Event()

        def sampler():
            try:
                import amdsmi
                amdsmi.amdsmi_init()
                h = amdsmi.amdsmi_get_processor_handles()[int(os.environ.get("LOCAL_RANK", 0))]
                while not stop.is_set():
                    pw = amdsmi.amdsmi_get_power_info(h)
                    ck = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
                    samples.append((time.perf_counter(), pw.get("current_socket_power", pw.get("average_socket_power")), ck.get("clk", ck.get("cur_clk")), pw.get("power_limit")))
                    time.sleep(0.01)
            except Exception:  # noqa: BLE001  (no amdsmi: the rate alone is reported)
                pass
        th = threading.Thread(target=sampler, daemon=True)
        th.start()
        t0 = time.perf_counter()
        r = subprocess.run([exe, str(seconds), "1"], capture_output=True, text=True, timeout=60)
        t1 = time.perf_counter()
        stop.set(); th.join(timeout=2)
        tf = float(r.stdout.split(" TFLOP/s")[0].split()[-1])
        busy = [x for x in samples if t0 + 0.4 * (t1 - t0) <= x[0] <= t1 and isinstance(x[1], (int, float))]
        out = {"mfma_f16_on_data_tflops": tf, "probe": "tools/clockprobe/powerprobe.hip mode 1, %.1f s" % seconds}
        if busy:
            out.update({"socket_power_w": sum(x[1] for x in busy) / len(busy), "sclk_mhz": sum(x[2] for x in busy) / len(busy),
                        "power_limit_w": (busy[0][3] / 1e6 if busy[0][3] and busy[0][3] > 1e5 else busy[0][3])})
        return out
    except Exception as e:  # noqa: BLE001
        return {"mfma_f16_on_data_tflops": None, "error": f"{type(e).__name__}: {e}"[:200]}


class clock_power_sampler:
    """Socket power and shader clock (amdsmi, every 20 ms, a host thread of its own) while a block runs: `with clock_power_sampler() as s: ...;
    s.summary()` -> {"sclk_mhz": mean, "socket_power_w": mean, "power_limit_w": ..., "samples": n} or None without amdsmi.  Reported in
    `roofline.headline_loop`: what the timed loop itself drew, beside the power-limited matrix rate of the probe."""

    def __init__(self, period=0.02):
        import threading
        self.period, self.samples, self.stop, self.th = period, [], threading.Event(), None

    def _run(self):
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            h = amdsmi.amdsmi_get_processor_handles()[int(os.environ.get("LOCAL_RANK", 0))]
            while not self.stop.is_set():
                pw = amdsmi.amdsmi_get_power_info(h)
                ck = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
                self.samples.append((pw.get("current_socket_power", pw.get("average_socket_power")), ck.get("clk", ck.get("cur_clk")), pw.get("power_limit")))
                time.sleep(self.period)
        except Exception:  # noqa: BLE001  (no amdsmi / no permission: nothing is reported)
            pass

    def __enter__(self):
        import threading
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        self.th.join(timeout=2)
        return False

    def summary(self):
        v = [x for x in self.samples[len(self.samples) // 5:] if isinstance(x[0], (int, float)) and isinstance(x[1], (int, float))]   # (the first fifth: ramp-up)
        if not v:
            return None
        lim = v[0][2]
        return {"sclk_mhz": sum(x[1] for x in v) / len(v), "socket_power_w": sum(x[0] for x in v) / len(v),
                "power_limit_w": (lim / 1e6 if isinstance(lim, (int, float)) and lim > 1e5 else lim), "samples": len(v)}


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


def hot_path_step(pol, env, inst, decode=None):
    """test.py:188-213 shaped: augment -> reset -> policy -> reward -> max over starts, then over augs.  The encoder's
    neighbour sample is drawn INSIDE the step, per forward and over all 8 x B instance-augmentations, as the reference does
    (rrnco/models/env_embeddings/atsp.py:55-67) — on the device (csrc/rr_sample.hip)."""
    from rrnco_amd import TensorDict
    from rrnco_amd.models.transforms import StateAugmentation
    from rrnco_amd.ops import unbatchify
    td = TensorDict(dict(inst), batch_size=[inst["locs"].shape[0]])
    td = StateAugmentation(augment_fn="dihedral8", no_aug_coords=False)(td)
    td = env.reset(td)
    out = pol(td, env, phase="val", num_starts=STARTS, return_actions=True, **(decode or {"decode_type": "multistart_greedy"}))
    rew = unbatchify(out["reward"], (AUG, STARTS))          # [B, A, S]
    best = rew.max(dim=-1).values.max(dim=-1).values
    return best, out


def cpu_baseline(w, seed=4321, budget_s=20.0, max_inst=8):
    """The oracle (op-for-op CPU restatement of the reference, fp32) timed on the host cores on a BOUNDED
    sample of the same workload: instances x 8 augmentations x 100 starts, one instance (8 instance-augs)
    per micro-batch, as many instances as fit in ~budget_s."""
    from oracle import restate
    host_cores = os.cpu_count() or 1

    def run(threads, budget, cap):
        torch.set_num_threads(threads)
        done, spent = 0, 0.0
        while done < cap and (done == 0 or spent + spent / done < budget):
            inst = restate.atsp_synthetic(1, N_NODES, seed + done)
            t0 = time.perf_counter()
            with torch.inference_mode():
                st = restate.atsp_reset(restate.augment_state(inst))
                sidx = restate.sample_neighbor_indices(st["distance_matrix"], 25)
                restate.atsp_policy(w, st, sidx, STARTS, "greedy")
            spent += time.perf_counter() - t0
            done += 1
        return done, spent
    threads = min(host_cores, 32)              # torch-CPU scales poorly past a few dozen threads on these ops
    done, spent = run(threads, budget_s, max_inst)
    rec = {"value": done / spent, "unit": "instances/s", "cores": threads, "threads": threads, "host_cores": host_cores, "kind": "port",
           "sample": f"{done} ATSP n=100 instance(s) x8 aug x100 starts greedy, torch-CPU fp32 oracle, {spent:.1f} s"}
    if host_cores > threads:
        # BASELINE.md section 3 asks for the box's host cores.  torch's intra-op pool oversubscribes these small ops badly past a few dozen
        # threads (measured: 505 s for ONE instance at 256 threads against 1.8 s at 32), so the wider points run in child processes under
        # hard time limits and are reported as measured or as "timed out", never waited for: 64 and 128 threads in one process, all
        # cores in one process, and — what a CPU user of the reference would actually do with 256 cores — a POOL of processes with 32
        # threads each, every process solving its own instances (instances are independent).
        import subprocess
        code = ("import sys,time,torch;sys.path.insert(0,%r);import bench;from oracle import restate;torch.set_num_threads(%d);"
                "pol,w=bench.make_policy('cpu');n=%d;t=time.perf_counter();\n"
                "for k in range(n):\n"
                " inst=restate.atsp_synthetic(1,bench.N_NODES,%d+k)\n"
                " with torch.inference_mode():\n"
                "  st=restate.atsp_reset(restate.augment_state(inst));sidx=restate.sample_neighbor_indices(st['distance_matrix'],25);"
                "restate.atsp_policy(w,st,sidx,bench.STARTS,'greedy')\n"
                "print('SECONDS',time.perf_counter()-t)")

        def child(nthreads, ninst, limit, first_seed):
            return subprocess.Popen([sys.executable, "-c", code % (ROOT, nthreads, ninst, first_seed)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                    text=True, env={**os.environ, "OMP_NUM_THREADS": str(nthreads)}), limit

        def seconds(proc, limit):
            try:
                out, _ = proc.communicate(timeout=limit)
                return float([l for l in out.splitlines() if l.startswith("SECONDS")][-1].split()[1])
            except Exception:  # noqa: BLE001  (timeout or no output)
                proc.kill()
                return None
        points = {}
        for nt, lim in ((64, 25), (128, 25), (host_cores, 45)):
            if nt > host_cores or str(nt) in points:
                continue
            sec = seconds(*child(nt, 1, lim, seed))
            points[str(nt)] = ({"value": 1.0 / sec, "unit": "instances/s", "threads": nt, "sample": f"1 instance, {sec:.1f} s, one process"} if sec
                               else {"value": None, "threads": nt, "sample": f"not finished within {lim} s in one process"})
        nproc, per = max(host_cores // threads, 1), 2
        t0 = time.perf_counter()
        procs = [child(threads, per, 60, seed + 100 * i) for i in range(nproc)]
        secs = [seconds(pr, lim) for pr, lim in procs]
        wall = time.perf_counter() - t0
        if all(x is not None for x in secs):
            # throughput over the slowest process's solve time (interpreter start-up and the imports are not the workload)
            points["pool"] = {"value": nproc * per / max(secs), "unit": "instances/s", "processes": nproc, "threads_per_process": threads,
                              "cores": nproc * threads, "sample": f"{nproc} processes x {per} instances, slowest {max(secs):.1f} s (wall incl. start-up {wall:.1f} s)"}
        else:
            points["pool"] = {"value": None, "sample": f"{nproc} processes x {threads} threads: not finished within 60 s"}
        rec["wider"] = points
        rec["all_host_cores"] = points.get(str(host_cores))
        best = max((v for v in points.values() if v and v.get("value")), key=lambda v: v["value"], default=None)
        if best and best["value"] > rec["value"]:            # the fastest measured point is the baseline
            rec.update({"value": best["value"], "cores": best.get("cores", best.get("threads")), "threads": best.get("threads", best.get("threads_per_process")),
                        "sample": best["sample"] + f" ({'pool of processes' if 'processes' in best else 'one process'}; 32-thread single process: {done / spent:.3f} instances/s)"})
    return rec


ENC_BLOCK_FLOP = 48e6                    # GEMM flop of one AttnFree_Block per instance, two blocks per layer
ENC_KERNEL = " + ".join(ENCODER_LAYER_KERNELS)      # the names rocprofv3's kernel stats list for one rr_enc_layer call
N_INSTANCE_BATCHES = 4                   # distinct instance batches rotated through the timed loop (test.py streams new batches)


class kernel_timers:
    """HIP events around the named C-ABI launchers (on torch's current stream, the stream they launch on): the launchers are
    one kernel each, so the event pair times that kernel.  with kernel_timers("rr_enc_layer", ...) as t: ...; t.ms("rr_enc_layer")."""

    def __init__(self, *names):
        self.names, self.ev, self.saved = names, {n: [] for n in names}, {}

    def __enter__(self):
        from rrnco_amd import _lib as L
        lib = L.lib()
        for n in self.names:
            fn = getattr(lib, n)
            self.saved[n] = fn

            def wrap(*a, _fn=fn, _n=n):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = _fn(*a)
                e1.record()
                self.ev[_n].append((e0, e1))
                return r
            setattr(lib, n, wrap)
        return self

    def __exit__(self, *exc):
        from rrnco_amd import _lib as L
        for n, fn in self.saved.items():
            setattr(L.lib(), n, fn)
        return False

    def ms(self, name):
        """(mean ms per call, calls)"""
        v = [a.elapsed_time(b) for a, b in self.ev[name]]
        return (sum(v) / len(v), len(v)) if v else (0.0, 0)

    def reset(self):
        for n in self.ev:
            self.ev[n] = []


def timed_loop(step, min_seconds=1.0, min_steps=2, max_steps=200):
    """Times step() over >= min_seconds WITHOUT a synchronisation between the steps (two probing steps size the loop; the host
    runs ahead of the device exactly as in the headline loop); -> (seconds per step, steps)."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(); step()
    torch.cuda.synchronize()
    est = max((time.perf_counter() - t0) / 2, 1e-4)
    n = int(min(max_steps, max(min_steps, -(-min_seconds // est))))
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, n


def graph_replay_exact_shape(step, res, pol, B, make_td=None, call=None):
    """-> the `hipgraph_replay_exact_shape` variant entry of a VRP config (never raises: a variant must not fail the line).
    Default: the whole step() captured.  make_td / call: augmentation + env.reset stay eager (make_td() per batch, copied into the
    captured call's input buffers) and only call(static_td) — the policy — is captured: for environments whose reset reads back."""
    try:
        from rrnco_amd import TensorDict
        static = None
        if make_td is not None:
            td0 = make_td()
            static = TensorDict({k: v.clone() for k, v in td0.items()}, batch_size=td0.batch_size, meta=dict(td0.meta))
        body = step if static is None else (lambda: res.__setitem__("out", call(static.clone())))
        gph, gs = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        gs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(gs):
            body()
            torch.cuda.synchronize()
            with torch.cuda.graph(gph, stream=gs):
                body()
            g_out = res["out"]
        torch.cuda.current_stream().wait_stream(gs)
        torch.cuda.synchronize()

        def gstep():
            if static is not None:
                td = make_td()
                for k, v in td.items():
                    static[k].copy_(v)
            gph.replay()
            Tg = int(g_out["steps"].item()) + 1
            res["trimmed"] = (g_out["actions"][:, :Tg], g_out["reward"])
        gstep()
        sec_g, n_g = timed_loop(gstep)
        pol.check_range()
        return {"value": B / sec_g, "ms_per_step": sec_g * 1e3, "steps": n_g,
                "note": "hipGraph replay of the call + one host read of the step count + the trim as a view: the reference's output shape "
                        "without the eager loop's launch gaps behind the read; range guard deferred to one read per dataset (evaluate.py --hipgraph)"}
    except Exception as e:
        return {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}


def other_configs(dev):
    """BASELINE.json configs[2], [3] and one configs[4] shard on this GPU, each timed for >= 1 s: instances/s, ms per step, the
    rollout kernel's time (HIP events) and a roofline block (C5: the backward's dominant kernel)."""
    from rrnco_amd import TensorDict
    from rrnco_amd.envs import ATSPEnv, RCVRPEnv, RMTVRPEnv
    from rrnco_amd.models import RRNetPolicy, rollout as R
    from rrnco_amd.models.encoder import ATSPInitEmbedding
    from rrnco_amd.models.rl import RRNet
    from rrnco_amd.models.transforms import StateAugmentation
    out = {}
    peak_split = PEAK_F16_MFMA_TFLOPS / SPLIT_PRODUCTS

    def vrp_policy(env_name):
        torch.manual_seed(1234)
        return RRNetPolicy(env_name=env_name, embed_dim=128, num_heads=8, num_encoder_layers=6, normalization="instance",
                           use_graph_context=False, nab_type="gating", init_embedding_kwargs=dict(sample_size=25)).to(dev).eval()

    def two_streams(env, pol, make_pol, step_of, B, seconds=1.5):
        """The same batches through TWO streams (rrnco_amd.parallel.run_on_streams: one host thread and one policy object per stream, same
        weights) -> (seconds per batch over both streams, batches): whole-job throughput, every batch still one complete pass."""
        from rrnco_amd.parallel import run_on_streams
        pols = [pol, make_pol()]
        pols[1].load_state_dict(pol.state_dict())
        pols[1].lazy_trim = getattr(pol, "lazy_trim", False)
        steps = [step_of(p) for p in pols]
        run_on_streams(steps, 2)                               # each stream's allocations and packs
        est = run_on_streams(steps, 2) / 4
        n = max(3, int(seconds / est / 2))
        return run_on_streams(steps, n) / (2 * n), 2 * n

    def inference(label, env, pol, B, S, aug, decode, kernel, make_pol=None, graph_variant=None):
        inst = env.generator(B, generator=torch.Generator(device=dev).manual_seed(5))
        sidx = ATSPInitEmbedding.sample_indices(env.reset(inst)["distance_matrix"], 25)
        if aug:
            sidx = sidx.repeat(8, 1, 1).contiguous()
        res = {}

        def step_of(p):
            def step():
                td = TensorDict(dict(inst.items()), batch_size=[B])
                if aug:
                    td = StateAugmentation(augment_fn="dihedral8", no_aug_coords=False)(td)
                td["sample_idx"] = sidx
                res["out"] = p(env.reset(td), env, phase="val", decode_type=decode, num_starts=S, seed=1)
            return step
        step = step_of(pol)
        step()
        sec_exact, n_exact = timed_loop(step)                   # actions trimmed to the longest route: one host read of the step count per call
        pol.lazy_trim = True                                    # the throughput form (models/policy.py): padded actions, step count on the device
        step()
        R.TIMING = []
        sec, n = timed_loop(step)
        ks = [a.elapsed_time(b) for a, b in R.TIMING]
        R.TIMING = None
        k_ms = sum(ks) / max(len(ks), 1)
        o = res["out"]
        pol.check_range()
        T, Rr = int(o["steps"].item()) + 1, int(o["actions"].shape[0])      # decode steps of the longest rollout + the multistart move
        # LIVE decoder evaluations only: a rollout counts up to the step that closes its last route (its last
        # customer + the return to the depot); what a finished rollout's tile keeps executing until the instance's longest route ends
        # is padding, reported separately as `executed`
        acts = o["actions"][:, :T]
        pos = torch.arange(T, device=acts.device)
        live_steps = int((((acts != 0).long() * pos).max(dim=1).values + 1).clamp(max=T - 1).sum())
        ach = live_steps * FLOP_PER_ROLLOUT_STEP / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        exe = Rr * (T - 1) * FLOP_PER_ROLLOUT_STEP / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        # `value` = the reference's own output contract (actions / log-probabilities trimmed to the longest route, the range guard read in
        # the call: one host read per call); the padded, read-free call form is a named variant
        out[label] = {"value": B / sec_exact, "unit": "instances/s", "ms_per_step": sec_exact * 1e3, "steps": n_exact, "kernel_ms": k_ms,
                      "call_form": "the reference's output shape: actions trimmed to the longest route, per-call range guard (one host read per call)",
                      "variants": {"lazy_trim": {"value": B / sec, "ms_per_step": sec * 1e3, "steps": n,
                                                 "note": "policy.lazy_trim = True: actions / log-probabilities keep their allocated length (depot / 0.0 "
                                                         "behind each route's end), the step count stays on the device, the range guard runs deferred "
                                                         "— no host read in the call; NOT the reference's output shape"}},
                      "rollouts": Rr, "decode_steps": T, "live_rollout_steps": live_steps, "executed_rollout_steps": Rr * (T - 1),
                      "mean_best_cost": float(-o["reward"].view(S, -1).max(0).values.mean()),
                      "roofline": {"bound": "mfma", "kernel": kernel, "achieved": ach, "peak": peak_split, "unit": "TFLOP/s",
                                   "frac": ach / peak_split, "traffic": measured_other_traffic(kernel),
                                   "traffic_note": "HBM-side bytes of one launch (tools/pmc_traffic_others.sh: separate FETCH_SIZE / WRITE_SIZE passes, "
                                                   "FETCH doubled per the guide); null: the committed summary is of other sources",
                                   "executed": {"achieved": exe, "frac": exe / peak_split},
                                   "note": "404 480 flop per LIVE rollout-step (each rollout up to the step that closes its last route); `executed` "
                                           "counts every rollout until the instance's longest route ends; fp32-equivalent peak of the fp16 pipe "
                                           "with 3 partial products"}}
        # The same call (augmentation, env.reset, policy) captured ONCE into a hipGraph — padded form inside the graph: no read while
        # capturing — and replayed, followed by the ONE host read per call that the reference's output shape needs, the step count, and
        # the trim as a view: what evaluate.py --hipgraph does per batch.  The eager `value` above pays ~70 launch gaps between that read
        # and the next call's first large kernel (the chip idles ~10 % of a C3 step); a replay is one launch.  The captured neighbour
        # sample repeats: a timing variant, like the headline's hipgraph_replay.  (graph_variant="policy": augmentation + reset eager, only
        # the policy call replayed — for environments whose reset reads back; RMTVRPEnv.reset no longer does for the vrptw preset.)
        if graph_variant == "step":
            out[label]["variants"]["hipgraph_replay_exact_shape"] = graph_replay_exact_shape(step, res, pol, B)
        elif graph_variant == "policy":      # (RMTVRPEnv.reset reads a flag back: augmentation + reset eager, the policy call captured)
            def make_td():
                td = TensorDict(dict(inst.items()), batch_size=[B])
                if aug:
                    td = StateAugmentation(augment_fn="dihedral8", no_aug_coords=False)(td)
                td["sample_idx"] = sidx
                return env.reset(td)
            out[label]["variants"]["hipgraph_replay_exact_shape"] = graph_replay_exact_shape(
                step, res, pol, B, make_td, lambda t: pol(t, env, phase="val", decode_type=decode, num_starts=S, seed=1))
        if make_pol is not None:
            sec2, n2 = two_streams(env, pol, make_pol, step_of, B)
            out[label]["two_streams"] = {"value": B / sec2, "unit": "instances/s", "ms_per_batch": sec2 * 1e3, "batches": n2,
                                         "note": "the same batches through two HIP streams, one host thread and one policy object each (same "
                                                 "weights): one stream's encoder / NAB kernels fill the CUs that the other's draining rollout "
                                                 "(workgroup = instance, length = its longest route) leaves idle, and cover its host-side step-count "
                                                 "read; `value` above stays the single-stream figure of the earlier rounds"}
        return step

    env = RCVRPEnv(generator_params=dict(num_loc=N_NODES, device=dev), check_solution=False, device=dev)
    inference("C3 RCVRP n=100 B=512 POMO S=101 greedy (configs[2])", env, vrp_policy("rcvrp"), 512, 101, False, "multistart_greedy",
              "k_rollout_w<7, 1, 0, true, true, false, false>", graph_variant="step")
    env = RMTVRPEnv(generator_params=dict(num_loc=N_NODES, device=dev), device=dev)
    c4 = "C4 RCVRPTW n=100 B=256 x8 aug S=100 sampling (configs[3])"
    c4_step = inference(c4, env, vrp_policy("rcvrptw"), 256, 100, True, "multistart_sampling", "k_rollout_w<7, 2, 1, true, true, false, false>",
                        make_pol=lambda: vrp_policy("rcvrptw"), graph_variant="step")
    # the step's second kernel: the Neural Adaptive Bias with the duration matrix (k_nab_dur_lds, 6 launches per step), VALU-bound on the
    # SiLU of its gate: per edge and gate unit one v_exp_f32 and one v_rcp_f32 — quarter-rate instructions (16 lanes per SIMD and 4 cycles)
    with kernel_timers("rr_nab_dur", "rr_nab_dur_aug") as kt:
        for _ in range(3):
            c4_step()
        torch.cuda.synchronize()
        nd_ms, nd_calls = kt.ms("rr_nab_dur_aug")              # x8 augmentation: distance / duration part shared by the copies (round 5)
        nd_kernel = "k_nab_dur_aug<8>"
        if nd_calls == 0:
            (nd_ms, nd_calls), nd_kernel = kt.ms("rr_nab_dur"), "k_nab_dur_lds<5>"
    if nd_ms > 0:
        edge_units = 256 * 8 * 2 * (N_NODES + 1) ** 2 * 128          # instances x aug x (row, col block) x edges x gate units, per launch
        peak_trans = 256 * 4 * 16 / 4 * 2.4e9                         # CUs x SIMDs x 16 lanes / 4 (quarter rate) x clock = 9.8e12 per second
        tr = 2 * edge_units / (nd_ms * 1e-3)
        # the ONE limit this kernel is held to: the vector pipe's ISSUE time of its whole instruction mix — per
        # (base edge, gate-unit pair, copy) 4 quarter-rate transcendentals (16 cycles each) + 11 packed instructions (4 cycles each) = 108
        # cycles of one SIMD for 64 lanes; `transcendental_floor` keeps the round-5 figure (the exp / rcp alone)
        issue_cycles = 108.0 if nd_kernel.startswith("k_nab_dur_aug") else 4 * 16.0 + 14 * 4.0
        items = edge_units / 2.0                                      # (edge, copy, unit pair)
        floor_ms = items / 64.0 * issue_cycles / (256 * 4) / 2.4e9 * 1e3
        out[c4]["roofline_nab_dur"] = {"bound": "valu-issue", "kernel": nd_kernel, "kernel_ms": nd_ms, "launches_per_step": nd_calls / 3,
                                       "achieved": items / (nd_ms * 1e-3) / 1e12, "peak": items / (floor_ms * 1e-3) / 1e12,
                                       "unit": "T (edge, copy, gate-unit pair) items / s", "frac": floor_ms / nd_ms,
                                       "issue_cycles_per_item_and_wave": issue_cycles,
                                       "transcendental_floor": {"achieved": tr / 1e12, "peak": peak_trans / 1e12, "frac": tr / peak_trans,
                                                                "unit": "T transcendental instructions (lane) / s"},
                                       "note": "2 transcendentals (exp, rcp of the gate's SiLU) per edge, copy and gate unit against the quarter-rate "
                                               "issue limit of the vector pipe — a FLOOR of one instruction class, not the kernel's bound: the packed "
                                               "fp32 arithmetic around them shares the issue port (11 packed instructions per unit pair and copy) and "
                                               "every (edge, copy, unit pair) reads a 16-byte table row from LDS (k_nab_dur_lds: three)",
                                       "lds_read_bytes_per_launch": edge_units // 2 * 16 * (1 if nd_kernel.startswith("k_nab_dur_aug") else 3)
                                                                    + (edge_units // 8 // 2 * 32 if nd_kernel.startswith("k_nab_dur_aug") else 0)}
    torch.cuda.empty_cache()

    # configs[4], one rank's shard: REINFORCE step on 512 ATSP instances (sampling rollout with the training dump, hand-written
    # backward, clip, fused Adam; the flat RCCL all-reduce is a no-op at one rank)
    pol, _ = make_policy(dev)
    pol.train()
    env = ATSPEnv(generator_params=dict(num_loc=N_NODES, device=dev), check_solution=False, device=dev)
    model = RRNet(env, policy=pol)
    opt = torch.optim.Adam(pol.parameters(), lr=1e-4, fused=True)
    gen = torch.Generator(device=dev).manual_seed(77)
    batches = [env.generator(512, generator=gen) for _ in range(3)]
    state = {"i": 0}

    def train_step():
        state["out"] = model.training_step(batches[state["i"] % 3], optimizer=opt, world=1, seed=2 + state["i"], grad_clip=1.0)
        state["i"] += 1
    train_step()
    bw = ("rr_dec_attn_bwd", "rr_mlp_wgrad", "rr_mlp_rows", "rr_dec_logit_bwd", "rr_gemm_tn", "rr_linear_rows", "rr_aft_bwd",
          "rr_inorm_bwd", "rr_nab_hist_bwd", "rr_enc_layer_train")
    sec, n = timed_loop(train_step)                     # the step's wall clock, nothing wrapped around its launchers
    with kernel_timers(*bw) as kt:                      # then a few steps with HIP events around the backward's launchers (~3 ms per step of host work)
        R.TIMING = []
        nk = 4
        for _ in range(nk):
            train_step()
        torch.cuda.synchronize()
        ks = [a.elapsed_time(b) for a, b in R.TIMING]
        R.TIMING = None
        per_step = {k: kt.ms(k)[0] * kt.ms(k)[1] / nk for k in bw}
    dom = max(per_step, key=per_step.get)
    rows = 512 * STARTS * (N_NODES - 2)                  # decoder evaluations of the step (the forced last move is not evaluated)
    roof = {"bound": "mfma", "kernel": dom, "kernel_ms_per_step": per_step[dom], "calls_per_step": kt.ms(dom)[1] / nk,
            "traffic": measured_other_traffic("k_mlp_wgrad<false>") if dom == "rr_mlp_wgrad" else None,
            "traffic_note": "HBM-side bytes of ONE launch of the dominant kernel (mean over its calls in a step), tools/pmc_traffic_others.sh"}
    if dom == "rr_dec_attn_bwd":
        # masked 8-head glimpse backward per decoder evaluation: recomputed scores, dP, dQ, dK, dV = 5 products of 2 N E flop
        fl = rows * 5 * 2 * N_NODES * 128
        roof.update({"achieved": fl / (per_step[dom] * 1e-3) / 1e12, "peak": peak_split, "unit": "TFLOP/s",
                     "note": "5 x 2 N E flop per decoder evaluation (scores recomputed, dP, dQ, dK, dV), two-piece bf16 operands: 2 500 / 3"})
    elif dom == "rr_mlp_wgrad":
        # pointer MLP + 12 encoder FFNs: per row the hidden layer recomputed, dH, dW1, dW2 = 4 products of 2 x 128 x 512 flop, bf16 two-piece
        fl = (rows + 12 * 512 * N_NODES) * 4 * 2 * 128 * 512
        roof.update({"achieved": fl / (per_step[dom] * 1e-3) / 1e12, "peak": peak_split, "unit": "TFLOP/s",
                     "note": "4 x 2 x 128 x 512 flop per row (hidden recomputed, dH, dW1, dW2), two-piece bf16 operands: 2 500 / 3"})
    if "achieved" in roof:
        roof["frac"] = roof["achieved"] / roof["peak"]
    o = state["out"]
    # the opt-in 16-mixed training step (configs/trainer/default.yaml:8 is the reference's own training precision): one bf16 piece per
    # operand in the pointer MLP's and the encoder FFNs' backward products, fp32 accumulation, gradients and master weights; its
    # gradient sits 2.3-2.6 % from the fp32-equivalent step's, the reference's own autocast gradient 9-12 % (fp16) / 70-87 % (bf16) from
    # its fp32 one (tests/test_gpu_mixed.py).  A VARIANT: never `value`.
    pol.precision = "16-mixed"
    train_step()
    sec16, n16 = timed_loop(train_step)
    pol.precision = "32"
    # the reference's own default training batch (configs/experiment/rrnet.yaml:39, train.py: 64 instances per device): 64 rollout
    # workgroups for 256 CUs unless the launcher splits the instances (csrc/rr_decode.hip, RolloutIO::wg_split)
    b64 = [env.generator(64, generator=gen) for _ in range(3)]

    def train_step64():
        state["out64"] = model.training_step(b64[state["i"] % 3], optimizer=opt, world=1, seed=2 + state["i"], grad_clip=1.0)
        state["i"] += 1
    train_step64()
    sec64, n64 = timed_loop(train_step64)
    out["C5 ATSP n=100 REINFORCE step, 512 instances (one rank's shard of configs[4])"] = {
        "variants": {"16_mixed_training_step (precision='16-mixed')": {
            "value": 512 / sec16, "unit": "trained instances/s", "ms_per_step": sec16 * 1e3, "steps": n16,
            "dtype": "bf16 operands (one piece) in the two 128-512-128 MLPs' backward products, f32 accumulate; everything else as the default step"},
            "train_b64 (the reference's default batch per device, rrnet.yaml:39)": {
                "value": 64 / sec64, "unit": "trained instances/s", "ms_per_step": sec64 * 1e3, "steps": n64,
                "share_of_b512_rate": (64 / sec64) / (512 / sec)}},
        "value": 512 / sec, "unit": "trained instances/s", "ms_per_step": sec * 1e3, "steps": n,
        "kernel_ms": sum(ks) / max(len(ks), 1), "kernel": "k_rollout_w<7, 0, 1, true, true, false, false> (sampling rollout with the training dump)",
        "backward_kernels_ms_per_step": {k: round(v, 3) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1])},
        "loss": float(o["loss"]), "grad_norm": float(o["grad_norm"]), "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
        "roofline": roof}
    del model, opt, pol, batches, state, b64
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=15)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch instances per GPU (the default); strong: --batch instances in all, parallel.shard_range per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip BASELINE configs[2..4] (timed on one GPU after the headline)")
    ap.add_argument("--no-variants", action="store_true", help="skip the fp32-MFMA and 16-mixed variants of the headline (diagnostic library builds hold only the headline kernel)")
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
    from rrnco_amd.parallel import aggregate_throughput, shard_range
    pol, w = make_policy(dev)
    env = ATSPEnv(generator_params=dict(num_loc=N_NODES, device=dev), check_solution=False, device=dev)
    # N_INSTANCE_BATCHES distinct synthetic batches, all resident in HBM before the timed region; timed step k solves batch k % 4
    # (test.py:188-213 streams new batches; replaying one batch would keep its matrices warm in the Infinity Cache)
    insts = []
    for nb in range(N_INSTANCE_BATCHES):
        if args.scaling == "strong":       # the B instances of ONE batch partitioned over the ranks
            lo, hi = shard_range(args.batch, rank, world)
            gen = torch.Generator(device=dev).manual_seed(instance_seed(nb, 0, "strong"))
            inst_td = ATSPGenerator(num_loc=N_NODES, device=dev)(args.batch, generator=gen)
            insts.append({"locs": inst_td["locs"][lo:hi].contiguous(), "distance_matrix": inst_td["distance_matrix"][lo:hi].contiguous()})
        else:
            gen = torch.Generator(device=dev).manual_seed(instance_seed(nb, rank, "weak"))
            inst_td = ATSPGenerator(num_loc=N_NODES, device=dev)(args.batch, generator=gen)
            insts.append({"locs": inst_td["locs"], "distance_matrix": inst_td["distance_matrix"]})
    inst = insts[0]
    local_batch = int(inst["locs"].shape[0])
    torch.manual_seed(sample_seed(rank, -1))           # the neighbour samples of the timed steps come from this stream

    def sync_all():
        torch.cuda.synchronize()
        if dist:
            td_.barrier()
            torch.cuda.synchronize()

    # headline = the default build (every GEMM of the encoder, the decoder cache and the whole rollout on two-piece fp16 operands);
    # the all-fp32-MFMA build is measured separately below
    os.environ.pop("RR_MLP_SPLIT", None)
    R.SPLIT_MLP = True
    for k in range(args.warmup):
        hot_path_step(pol, env, insts[k % N_INSTANCE_BATCHES])
    with kernel_timers("rr_enc_layer", "rr_enc_layer_split", "rr_nab_dist_family", "rr_init_embed", "rr_dec_cache") as kt, clock_power_sampler() as cps:
        R.TIMING = []
        sync_all()
        t0 = time.perf_counter()
        for k in range(args.steps):
            torch.manual_seed(sample_seed(rank, k))      # the step's neighbour sample (the variant below replays the same draws)
            best, out = hot_path_step(pol, env, insts[k % N_INSTANCE_BATCHES])
        sync_all()
        dt = time.perf_counter() - t0
        pol.check_range()                  # the range guard's deferred word (models/policy.py): a raised one fails the run loudly
        kern_ms = [a.elapsed_time(b) for a, b in R.TIMING]
        R.TIMING = None
        enc_ms, enc_calls = kt.ms("rr_enc_layer_split")          # the layer as three launches (default at this shape) ...
        fam_ms, fam_calls = kt.ms("rr_nab_dist_family")          # ... + the layer's shared distance-family lookup (x8-augmented batch)
        if enc_calls and fam_calls == enc_calls:
            enc_ms += fam_ms
        if enc_calls == 0:
            enc_ms, enc_calls = kt.ms("rr_enc_layer")            # ... or RR_ENC_SPLIT=0: block kernel + FFN kernel
        init_ms, _ = kt.ms("rr_init_embed")
        cache_ms, _ = kt.ms("rr_dec_cache")
    total_inst, dt = aggregate_throughput(local_batch * args.steps, dt, dist, dev if backend == "nccl" else torch.device("cpu"))
    devices = [f"cuda:{local}"]
    if dist:
        names = [None] * world
        td_.all_gather_object(names, f"rank {rank}: cuda:{local} ({torch.cuda.get_device_name(local)})")
        devices = names

    if rank == 0:
        rollout_steps = local_batch * AUG * STARTS * (N_NODES - 1)
        k_ms = sum(kern_ms) / max(len(kern_ms), 1)
        achieved = rollout_steps * FLOP_PER_ROLLOUT_STEP / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        peak_split = PEAK_F16_MFMA_TFLOPS / SPLIT_PRODUCTS
        # executed work counts tiles: S = 100 starts are 6 full 16-rollout tiles and one quarter-full one per instance (tail tiles are
        # not packed across instances: csrc/rr_decode.hip)
        tile_steps = local_batch * AUG * ((STARTS + 15) // 16) * (N_NODES - 1)
        executed = tile_steps * EXECUTED_F16_FLOP_PER_TILE_STEP / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        traffic, traffic_src = measured_rollout_traffic(local_batch)
        enc_ach = 2 * ENC_BLOCK_FLOP * local_batch * AUG / (enc_ms * 1e-3) / 1e12 if enc_ms > 0 else 0.0
        line = {
            "metric": "solved instances/sec (ATSP n=100, B=512, POMO greedy)", "value": total_inst / dt,
            "unit": "instances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "devices": devices,
            "configs_summary": None,         # (filled below; placed early so that a reader of the line's head sees the other configs too)
            "arithmetic": "fp32 operands and fp32 accumulation throughout; the products of the rollout (scores, P.V, pointer MLP, logits), "
                          "of the encoder (projections, AFT mixing, FFN) and of the decoder cache multiply two-piece fp16 splits of the fp32 "
                          "operands on the fp16 matrix pipe (3 partial products: hi*hi, hi*lo, lo*hi; error of a dot product 4e-8 .. 8e-8 of "
                          "sum|ab| against 1.1e-7 for the fp32 MFMA: tools/clockprobe/f16probe.hip, usplitprobe.hip); operands that leave the "
                          "fp16 range raise a status word and the call repeats on the fp32-MFMA kernels (models/policy.py); tours equal to "
                          "the all-fp32-MFMA build's (tests/test_gpu_fullsize.py), which is timed under `variants`",
            "config": {"workload": f"ATSP n={N_NODES}, batch={args.batch}{'/GPU' if args.scaling == 'weak' else ' in all'}, POMO S={STARTS} starts x "
                                   f"{AUG} dihedral aug, greedy (BASELINE.json configs[1]); random-init RRNet E=128 L=6; "
                                   f"{N_INSTANCE_BATCHES} distinct instance batches rotated through the timed steps",
                       "rollouts_per_gpu": local_batch * AUG * STARTS,
                       "sharding": f"instances over {world} rank(s), no collective ({args.scaling} scaling: {local_batch} instances on rank 0)"},
            "roofline": {"bound": "mfma", "kernel": ROLLOUT_KERNEL + " (persistent wave-autonomous POMO decode)", "achieved": achieved,
                         "peak": peak_split, "unit": "TFLOP/s", "frac": achieved / peak_split,
                         "peak_note": "fp32-equivalent flop/s of the fp16 matrix pipe at its dense peak with 3 partial products per product "
                                      "(2 500 / 3); round 1's all-fp32-MFMA kernel (peak 157.3) is under `variants`",
                         "executed_mfma": {"achieved": executed, "peak": PEAK_F16_MFMA_TFLOPS, "frac": executed / PEAK_F16_MFMA_TFLOPS,
                                           "note": "1 104 matrix instructions (16 384 flop-slots each) per 16-rollout tile and decode step, 7 tiles per 100 starts"},
                         "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": k_ms,
                         "mfma_busy": measured_mfma_busy(ROLLOUT_KERNEL, local_batch),
                         "mfma_busy_note": "matrix-pipe busy fraction from the committed rocprofv3 --pmc pass of these sources (SQ_VALU_MFMA_BUSY_CYCLES / "
                                           "1 024 SIMDs over GRBM_GUI_ACTIVE / 8); null: the summary is of other sources",
                         "algorithmic_flop_per_launch": rollout_steps * FLOP_PER_ROLLOUT_STEP,
                         "headline_loop": cps.summary(),
                         "headline_loop_note": "shader clock and socket power sampled (amdsmi, 20 ms) over the timed steps; null: amdsmi not "
                                               "available to this user.  power_limited (below, default runs): the fp16 matrix rate of a pure "
                                               "register-resident matrix stream on data under the same cap"},
            "roofline_encoder": {"bound": "mfma", "kernel": ENC_KERNEL + " (one rr_nab_dist_family + one rr_enc_layer_split call = the row and the column AttnFree_Block of a layer)",
                                 "achieved": enc_ach, "peak": peak_split, "unit": "TFLOP/s", "frac": enc_ach / peak_split if peak_split else 0.0,
                                 "kernel_ms": enc_ms, "launches_per_step": enc_calls / max(args.steps, 1),
                                 "traffic": measured_encoder_traffic(local_batch),
                                 "mfma_busy": {k: measured_mfma_busy(k, local_batch, encoder=True) for k in ENCODER_LAYER_KERNELS},
                                 "traffic_note": "HBM-side bytes of one layer (its three kernels), same PMC passes as the rollout's; "
                                                 "null: the committed summary is of other encoder sources",
                                 "algorithmic_flop_per_launch": 2 * ENC_BLOCK_FLOP * local_batch * AUG,
                                 "init_embed_ms": init_ms, "dec_cache_ms": cache_ms},
            "mean_best_cost": float(-best.mean().item()),
        }
        if world == 1 and not args.no_variants:
            def timed(label):
                hot_path_step(pol, env, insts[0])                      # (warm-up of the variant's kernels / packs)
                R.TIMING = []
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                nv = max(min(args.steps, 8), 1)
                for k in range(args.steps - nv, args.steps):     # the seeds and batches of the headline loop's last nv steps: same neighbour samples
                    torch.manual_seed(sample_seed(rank, k))
                    best_v, _ = hot_path_step(pol, env, insts[k % N_INSTANCE_BATCHES])
                torch.cuda.synchronize()
                dtv = time.perf_counter() - t1
                ks = [a.elapsed_time(b) for a, b in R.TIMING]
                R.TIMING = None
                kv = sum(ks) / max(len(ks), 1)
                return {"value": local_batch * nv / dtv, "unit": "instances/s", "ms_per_step": dtv / nv * 1e3, "steps": nv,
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
            pol.invalidate_pack()
            # the reference's own GPU arithmetic mode (torch.autocast in test.py:183 / Lightning 16-mixed): one fp16 piece per operand in
            # the fused rollout, fp32 accumulation; encoder and cache unchanged.  NOT the headline: other arithmetic, other tolerance
            # (tests/test_gpu_mixed.py: inside the reference's own autocast deviation).  Priced against the plain fp16 peak.
            pol.precision = "16-mixed"
            v16, k16 = timed("16-mixed")
            pol.precision = "32"
            a16 = rollout_steps * FLOP_PER_ROLLOUT_STEP / (k16 * 1e-3) / 1e12
            v16["dtype"] = "f16 operands (one piece), f32 accumulate, f32 softmax / logits"
            v16["roofline"] = {"bound": "mfma", "kernel": "k_rollout_w<7, 0, 0, true, true, true, false>", "achieved": a16, "peak": PEAK_F16_MFMA_TFLOPS,
                               "unit": "TFLOP/s", "frac": a16 / PEAK_F16_MFMA_TFLOPS}
            line["variants"]["16_mixed_rollout (precision='16-mixed', the reference's autocast mode)"] = v16
            # sampling with process_logits' top-k / top-p filters (decoding.py:352-358): inside the fused rollout since round 5 (FILT builds of
            # the two-piece kernels); before, such a strategy ran the per-step loop (`per_step_loop`, two steps timed)
            filt = {"decode_type": "multistart_sampling", "top_k": 10, "top_p": 0.9, "temperature": 1.0, "seed": 3}

            def timed_decode(decode, nv):
                hot_path_step(pol, env, insts[0], decode)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for k in range(nv):
                    torch.manual_seed(sample_seed(rank, k))
                    hot_path_step(pol, env, insts[k % N_INSTANCE_BATCHES], decode)
                torch.cuda.synchronize()
                return {"value": local_batch * nv / (time.perf_counter() - t1), "unit": "instances/s", "steps": nv}
            vf = timed_decode(filt, max(min(args.steps, 8), 1))
            vf["unfiltered_sampling"] = timed_decode({"decode_type": "multistart_sampling", "temperature": 1.0, "seed": 3}, max(min(args.steps, 8), 1))["value"]
            vf["per_step_loop"] = timed_decode({**filt, "fused": False}, 2)["value"]
            line["variants"]["sampling_top_k10_top_p0.9_fused"] = vf
            # the reference's own default evaluation batch (test.py:90 --batch_size 32, x8 augmentation = 256 instance-augmentations = one
            # rollout workgroup per CU, a single round of the grid): same step, 32 instances per call
            small = [{k: v[:32].contiguous() for k, v in b.items()} for b in insts]
            hot_path_step(pol, env, small[0])
            torch.cuda.synchronize()
            nsm = max(min(4 * args.steps, 60), 4)
            t1 = time.perf_counter()
            for k in range(nsm):
                torch.manual_seed(sample_seed(rank, k))
                hot_path_step(pol, env, small[k % N_INSTANCE_BATCHES])
            torch.cuda.synchronize()
            dts = time.perf_counter() - t1
            line["variants"]["eval_b32_aug8 (the reference's default evaluation batch, test.py:90)"] = {
                "value": 32 * nsm / dts, "unit": "instances/s", "ms_per_step": dts / nsm * 1e3, "steps": nsm,
                "share_of_headline_rate": (32 * nsm / dts) / line["value"]}
            # the whole step (augmentation, reset, neighbour sample, encoder, rollout, reward, best-of) captured ONCE into a hipGraph and
            # replayed: no launcher allocates through the runtime or reads back while capturing (tests/test_gpu_graph.py).  What the
            # replay saves over the eager loop is the host side of ~60 launches; the captured neighbour sample repeats (a timing variant)
            try:
                gph, gs = torch.cuda.CUDAGraph(), torch.cuda.Stream()
                with torch.cuda.stream(gs):
                    hot_path_step(pol, env, insts[0])
                    torch.cuda.synchronize()
                    with torch.cuda.graph(gph, stream=gs):
                        best_g, _ = hot_path_step(pol, env, insts[0])
                torch.cuda.synchronize()
                gph.replay()
                torch.cuda.synchronize()
                nv = max(min(args.steps, 8), 1)
                t1 = time.perf_counter()
                for _ in range(nv):
                    gph.replay()
                torch.cuda.synchronize()
                dtg = time.perf_counter() - t1
                pol.check_range()
                line["variants"]["hipgraph_replay"] = {"value": local_batch * nv / dtg, "unit": "instances/s", "ms_per_step": dtg / nv * 1e3, "steps": nv,
                                                       "mean_best_cost": float(-best_g.mean().item())}
                del gph
            except Exception as e:      # (a variant: never fails the headline)
                line["variants"]["hipgraph_replay"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not args.no_other_configs:
            del pol
            torch.cuda.empty_cache()
            line["configs"] = other_configs(dev)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w)
        if world == 1 and not args.no_other_configs:      # (a diagnostic beside the contract's fields; ~3 s)
            torch.cuda.synchronize()
            pl = power_limited_mfma()
            if pl and pl.get("mfma_f16_on_data_tflops"):
                pk = pl["mfma_f16_on_data_tflops"] / SPLIT_PRODUCTS
                pl.update({"peak_fp32_equivalent": pk, "frac": line["roofline"]["achieved"] / pk,
                           "note": "the rollout's achieved rate over what a pure fp16 matrix stream on data sustains under the 1 400 W cap, "
                                   "three partial products per product; roofline.frac above is against the guide's 2 500 / 3"})
            line["roofline"]["power_limited"] = pl
        # compact digest of the line (instances/s unless named otherwise): at its head (placeholder above) and again at its very end
        cs = {"headline": round(line["value"], 1), "rollout_ms": round(line["roofline"]["kernel_ms"], 2), "rollout_frac": round(line["roofline"]["frac"], 4),
              "encoder_layer_ms": round(line["roofline_encoder"]["kernel_ms"], 3), "encoder_frac": round(line["roofline_encoder"]["frac"], 4)}
        for k, v in (line.get("configs") or {}).items():
            tag = k.split()[0]
            cs[tag] = round(v["value"], 1)
            if "two_streams" in v:
                cs[tag + "_two_streams"] = round(v["two_streams"]["value"], 1)
            for vk, vv in (v.get("variants") or {}).items():
                if isinstance(vv, dict) and vv.get("value"):          # (a variant that failed carries value None and its error)
                    cs[tag + "_" + vk.split()[0]] = round(vv["value"], 1)
        for k, v in (line.get("variants") or {}).items():
            if isinstance(v, dict) and v.get("value"):
                cs[k.split()[0]] = round(v["value"], 1)
        if line.get("cpu_baseline"):
            cs["cpu_baseline"] = round(line["cpu_baseline"]["value"], 3)
        line["configs_summary"] = cs
        line["summary_tail"] = cs
        print(json.dumps(line))
    if dist:
        td_.destroy_process_group()


if __name__ == "__main__":
    main()
