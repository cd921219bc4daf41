#!/bin/bash
# Round artefacts: bench line (with cpu_baseline), rocprofv3 kernel stats of the same command, HBM-side PMC traffic.
TAG=${1:-v3}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python3 bench.py > gpurun_out/bench_$TAG.jsonl 2> gpurun_out/bench_$TAG.err
tail -1 gpurun_out/bench_$TAG.jsonl | cut -c1-400
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_${TAG}_prof.jsonl 2> /tmp/prof_err.log
S=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$S" $R/gpurun_out/bench_${TAG}_kernel_stats.csv
head -14 $R/gpurun_out/bench_${TAG}_kernel_stats.csv | cut -c1-200
bash $R/tools/pmc_traffic.sh
