#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the dominant kernels of BASELINE configs[2..4]: C3's and C4's rollout
# (tools/profile_c4.py), C5's pointer-MLP weight-gradient kernel (tools/bench_train.py).  Writes gpurun_out/pmc_others.json (merged into
# bench_pmc_hbm_traffic.json by tools/pmc_traffic_json.py when present: bench.py fills `roofline.traffic` of those configs from it).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
rm -f $R/gpurun_out/pmc_others_*.txt
for C in FETCH_SIZE WRITE_SIZE; do
  PROBLEM=rcvrp rocprofv3 --pmc $C --output-format csv -d /tmp/pmco_c3_$C -- python3 $R/tools/profile_c4.py > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmco_c3_$C > $R/gpurun_out/pmc_others_c3_$C.txt; rm -rf /tmp/pmco_c3_$C
  rocprofv3 --pmc $C --output-format csv -d /tmp/pmco_c4_$C -- python3 $R/tools/profile_c4.py > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmco_c4_$C > $R/gpurun_out/pmc_others_c4_$C.txt; rm -rf /tmp/pmco_c4_$C
  rocprofv3 --pmc $C --output-format csv -d /tmp/pmco_c5_$C -- python3 $R/tools/bench_train.py --steps 1 > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmco_c5_$C > $R/gpurun_out/pmc_others_c5_$C.txt; rm -rf /tmp/pmco_c5_$C
done
python3 - <<PY
import json, re, sys, os
sys.path.insert(0, "$R")
import bench
want = {"c3": "void k_rollout_w<7, 1, 0, true, true, false, false>(", "c4": "void k_rollout_w<7, 2, 1, true, true, false, false>(",
        "c5": "void k_mlp_wgrad<false>("}
out = {"library_source_hash": bench.library_source_hash(), "kernels": {}}
for tag, name in want.items():
    rec = {}
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        lines = open(f"$R/gpurun_out/pmc_others_{tag}_{C}.txt").read().splitlines()
        for i, l in enumerate(lines):
            if l.startswith(name[:64][:len(l)]) and len(l) >= 20:
                m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+) mean=([0-9.e+]+)", lines[i + 1])
                if m:
                    rec[C + "_KB"] = float(m.group(3)); rec[C + "_n"] = int(m.group(2))
    if len(rec) == 4:
        rec["bytes_per_launch"] = (2 * rec["FETCH_SIZE_KB"] + rec["WRITE_SIZE_KB"]) * 1024.0
        out["kernels"][name[5:-1]] = rec
print(json.dumps(out))
open("$R/gpurun_out/pmc_others.json", "w").write(json.dumps(out))
PY
