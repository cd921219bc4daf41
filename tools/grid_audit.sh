#!/bin/bash
# Diagnostic: kernels of a command whose launches leave CUs without a workgroup (grid < 256 workgroups) and still take time.
# usage: grid_audit.sh <python script and args...>   (run through gpurun)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/ga
rocprofv3 --kernel-trace --output-format csv -d /tmp/ga -o p -- python3 $R/"$@" > /tmp/ga.log 2>&1
T=$(find /tmp/ga -name "*kernel_trace.csv" | head -1)
python3 - "$T" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0, 0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    nwg = grid // max(wg, 1)
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
    a = agg[(r["Kernel_Name"][:70], nwg, wg)]
    a[0] += 1; a[1] += dur
rows = sorted(((v[1], k, v[0]) for k, v in agg.items()), reverse=True)
print("total us  calls  workgroups x threads  kernel   (launches below 256 workgroups or with a ragged last round of a 1-WG/CU kernel)")
for tot, (name, nwg, wg), n in rows[:60]:
    flag = "LOW" if nwg < 256 else ("" if nwg % 256 == 0 or nwg >= 2048 else f"rag {nwg % 256}")
    print(f"{tot:10.0f} {n:5d}  {nwg:6d} x {wg:4d}  {flag:8s} {name}")
PY
