"""FETCH_SIZE / WRITE_SIZE (KB, mean per dispatch) of the default rollout kernel from the two pmc_summary outputs -> the JSON
bench.py reads (with the hash of the rollout's sources, so that a stale summary is never reported).  A third file, the counter
pass of tools/pmc_mfma.sh (gpurun_out/pmc_mfma.txt), adds `counters`: per kernel the matrix-pipe busy fraction
(SQ_VALU_MFMA_BUSY_CYCLES / 1 024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs) and vector instructions per matrix instruction."""
import json, os, re, sys
_paths = [a for a in sys.argv[1:] if os.path.exists(a)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
out = {"kernel": bench.ROLLOUT_KERNEL, "source_hash": bench.rollout_source_hash(), "batch": bench.BATCH}
for path in _paths:
    lines = open(path).read().splitlines()
    for i, l in enumerate(lines):
        if l.startswith(("void " + bench.ROLLOUT_KERNEL)[:len(l)]) and len(l) >= 40:
            m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+) mean=([0-9.e+]+)", lines[i + 1])
            if m:
                out[m.group(1) + "_KB"] = float(m.group(3)); out[m.group(1) + "_n"] = int(m.group(2))
enc = {"source_hash": bench.encoder_source_hash(), "batch": bench.BATCH, "kernels": {k: {} for k in bench.ENCODER_LAYER_KERNELS}}
for path in _paths:
    lines = open(path).read().splitlines()
    for i, l in enumerate(lines):
        for k in bench.ENCODER_LAYER_KERNELS:
            if l.startswith("void " + k + "(") or l.startswith(k + "("):      # (a non-template kernel is listed without its return type)
                m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+) mean=([0-9.e+]+)", lines[i + 1])
                if m:
                    enc["kernels"][k][m.group(1) + "_KB"] = float(m.group(3)); enc["kernels"][k][m.group(1) + "_n"] = int(m.group(2))
if all(len(v) == 4 for v in enc["kernels"].values()):
    out["encoder"] = enc
ctr = {"rollout_source_hash": bench.rollout_source_hash(), "encoder_source_hash": bench.encoder_source_hash(), "kernels": {}}
for path in _paths:
    lines = open(path).read().splitlines()
    for i, l in enumerate(lines):
        for k in (bench.ROLLOUT_KERNEL,) + tuple(bench.ENCODER_LAYER_KERNELS):
            if not (l.startswith("void " + k + "(") or l.startswith(k + "(")):
                continue
            vals = {}
            for ll in lines[i + 1:i + 12]:
                m = re.match(r"\s+(\w+)\s+n=\s*(\d+) mean=([0-9.e+]+)", ll)
                if not m:
                    break
                vals[m.group(1)] = float(m.group(3))
            if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and vals.get("GRBM_GUI_ACTIVE", 0) > 0:
                rec = ctr["kernels"].setdefault(k, {})
                rec["mfma_busy"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (vals["GRBM_GUI_ACTIVE"] / 8.0)
                if vals.get("SQ_INSTS_MFMA", 0) > 0:
                    rec["valu_per_mfma"] = (vals["SQ_INSTS_VALU"] - vals["SQ_INSTS_MFMA"]) / vals["SQ_INSTS_MFMA"]
                rec["gpu_cycles"] = vals["GRBM_GUI_ACTIVE"] / 8.0
if ctr["kernels"]:
    out["counters"] = ctr
_oth = os.path.join(ROOT, "gpurun_out", "pmc_others.json")       # tools/pmc_traffic_others.sh (configs[2..4]'s dominant kernels), when it ran
if os.path.exists(_oth):
    out["others"] = json.load(open(_oth))
print(json.dumps(out))
