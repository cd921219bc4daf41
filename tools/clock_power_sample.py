"""Diagnostic: GPU clock / power / temperature sampled by a host thread while the headline step (or another workload) runs.
Answers "is the rollout clock- or power-limited?": the roofline peak assumes 2.4 GHz.  Not part of the product.

    python tools/clock_power_sample.py [steps] [workload: headline|mlpsplit0|idle]
"""
import json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "real-routing-nco_amd"))
import torch

samples, stop = [], threading.Event()


def sampler_amdsmi():
    import amdsmi
    amdsmi.amdsmi_init()
    h = amdsmi.amdsmi_get_processor_handles()[0]
    while not stop.is_set():
        rec = {"t": time.perf_counter()}
        try:
            c = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
            rec["sclk"] = c.get("clk", c.get("cur_clk"))
        except Exception as e:  # noqa: BLE001
            rec["sclk_err"] = repr(e)[:80]
        try:
            p = amdsmi.amdsmi_get_power_info(h)
            rec["power"] = p.get("current_socket_power", p.get("average_socket_power"))
            rec["power_limit"] = p.get("power_limit")
        except Exception as e:  # noqa: BLE001
            rec["power_err"] = repr(e)[:80]
        try:
            m = amdsmi.amdsmi_get_gpu_metrics_info(h)
            for k in ("current_gfxclk", "average_gfxclk_frequency", "current_socket_power", "temperature_hotspot", "throttle_status",
                      "indep_throttle_status", "average_gfx_activity"):
                if k in m:
                    rec[k] = m[k]
            if "current_gfxclks" in m:
                v = [x for x in m["current_gfxclks"] if isinstance(x, (int, float)) and 0 < x < 60000]
                rec["gfxclks_min"], rec["gfxclks_max"] = (min(v), max(v)) if v else (None, None)
        except Exception as e:  # noqa: BLE001
            rec["metrics_err"] = repr(e)[:80]
        samples.append(rec)
        time.sleep(0.005)


def sampler_rocmsmi():
    while not stop.is_set():
        t = time.perf_counter()
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append({"t": t, "raw": json.loads(o)})
        except Exception as e:  # noqa: BLE001
            samples.append({"t": t, "err": repr(e)[:120]})
        time.sleep(0.05)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    what = sys.argv[2] if len(sys.argv) > 2 else "headline"
    import bench
    from rrnco_amd.envs import ATSPEnv, ATSPGenerator
    from rrnco_amd.models import rollout as R
    dev = torch.device("cuda")
    pol, _ = bench.make_policy(dev)
    env = ATSPEnv(generator_params=dict(num_loc=100, device=dev), check_solution=False, device=dev)
    inst_td = ATSPGenerator(num_loc=100, device=dev)(512, generator=torch.Generator(device=dev).manual_seed(1))
    inst = {"locs": inst_td["locs"], "distance_matrix": inst_td["distance_matrix"]}
    if what == "mlpsplit0":
        R.SPLIT_MLP = False
        os.environ["RR_MLP_SPLIT"] = "0"
    bench.hot_path_step(pol, env, inst)
    torch.cuda.synchronize()
    try:
        import amdsmi  # noqa: F401
        th = threading.Thread(target=sampler_amdsmi, daemon=True)
    except Exception:  # noqa: BLE001
        th = threading.Thread(target=sampler_rocmsmi, daemon=True)
    th.start()
    time.sleep(0.5)                      # idle samples
    t_start = time.perf_counter()
    R.TIMING = []
    gemm_tf = None
    if what.startswith("gemm"):          # a dense library GEMM on random data: what the fp16 / bf16 matrix pipe sustains under the power cap
        dt_ = torch.bfloat16 if what == "gemm_bf16" else torch.float16
        n = 8192
        a = torch.randn(n, n, device=dev, dtype=dt_); b = torch.randn(n, n, device=dev, dtype=dt_)
        for _ in range(3):
            a @ b
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps * 40):
            a @ b
        e1.record()
        torch.cuda.synchronize()
        gemm_tf = steps * 40 * 2 * n ** 3 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    elif what.startswith("cmd:"):        # sample while a child process (a clockprobe binary) runs
        r = subprocess.run(what[4:], shell=True, capture_output=True, text=True)
        print("child output:", (r.stdout + r.stderr)[-1500:])
    elif what != "idle":
        for _ in range(steps):
            bench.hot_path_step(pol, env, inst)
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    ks = [a.elapsed_time(b) for a, b in R.TIMING]
    R.TIMING = None
    time.sleep(0.3)
    stop.set(); th.join(timeout=2)
    busy = [s for s in samples if t_start + 0.2 <= s["t"] <= t_end]
    idle = [s for s in samples if s["t"] < t_start]
    print(f"workload {what}: {steps} steps, {(t_end - t_start) / max(steps, 1) * 1e3:.2f} ms per step, rollout kernel {sum(ks) / max(len(ks), 1):.2f} ms; "
          f"{len(busy)} busy samples, {len(idle)} idle samples" + (f"; GEMM 8192^3: {gemm_tf:.0f} TFLOP/s" if gemm_tf else ""))

    def stat(rows, k):
        v = [r[k] for r in rows if isinstance(r.get(k), (int, float))]
        return (f"{k}: min {min(v)} mean {sum(v) / len(v):.1f} max {max(v)} (n={len(v)})") if v else f"{k}: n/a"
    for k in ("sclk", "current_gfxclk", "average_gfxclk_frequency", "gfxclks_min", "gfxclks_max", "power", "current_socket_power", "power_limit",
              "temperature_hotspot", "throttle_status", "indep_throttle_status", "average_gfx_activity"):
        print("  busy", stat(busy, k)); print("  idle", stat(idle, k))
    errs = {k: v for r in samples[:3] for k, v in r.items() if k.endswith("err")}
    if errs:
        print("  errors:", errs)
    if samples and "raw" in samples[0]:
        print("  first raw:", json.dumps(samples[0]["raw"])[:600]); print("  mid raw:", json.dumps(busy[len(busy) // 2]["raw"])[:600] if busy else None)


if __name__ == "__main__":
    main()
