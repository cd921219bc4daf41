"""FETCH_SIZE / WRITE_SIZE (KB, mean per dispatch) of the default rollout kernel from the two pmc_summary outputs -> the JSON
bench.py reads (with the hash of the rollout's sources, so that a stale summary is never reported)."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
out = {"kernel": bench.ROLLOUT_KERNEL, "source_hash": bench.rollout_source_hash(), "batch": bench.BATCH}
for path in sys.argv[1:]:
    lines = open(path).read().splitlines()
    for i, l in enumerate(lines):
        if l.startswith("void " + bench.ROLLOUT_KERNEL):
            m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+) mean=([0-9.e+]+)", lines[i + 1])
            out[m.group(1) + "_KB"] = float(m.group(3)); out[m.group(1) + "_n"] = int(m.group(2))
enc = {"source_hash": bench.encoder_source_hash(), "batch": bench.BATCH, "kernels": {k: {} for k in bench.ENCODER_LAYER_KERNELS}}
for path in sys.argv[1:]:
    lines = open(path).read().splitlines()
    for i, l in enumerate(lines):
        for k in bench.ENCODER_LAYER_KERNELS:
            if l.startswith("void " + k[:24]) and (k != bench.ENCODER_LAYER_KERNELS[0] or l.startswith("void " + k)):
                m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+) mean=([0-9.e+]+)", lines[i + 1])
                enc["kernels"][k][m.group(1) + "_KB"] = float(m.group(3)); enc["kernels"][k][m.group(1) + "_n"] = int(m.group(2))
if all(len(v) == 4 for v in enc["kernels"].values()):
    out["encoder"] = enc
print(json.dumps(out))
