#!/bin/bash
# rocprofv3 kernel trace with the per-K# range markers on (RR_MARKERS=1): one small ATSP call, fused and step-wise.  Run through gpurun;
# copies the marker / kernel statistics to gpurun_out/markers_<TAG>_*.csv
TAG=${1:-v1}
export TMPDIR=/tmp RR_MARKERS=1
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d /tmp/mk_$TAG -o p -- python3 $R/tools/bench_stepwise.py > /tmp/mk.log 2>&1
for f in marker_api_stats kernel_stats domain_stats; do
  S=$(find /tmp/mk_$TAG -name "*${f}.csv" | head -1)
  [ -n "$S" ] && cp "$S" $R/gpurun_out/markers_${TAG}_${f}.csv
done
ls /tmp/mk_$TAG/* | head; tail -3 /tmp/mk.log
head -20 $R/gpurun_out/markers_${TAG}_marker_api_stats.csv
