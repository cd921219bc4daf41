#!/bin/bash
# quick GPU check of the encoder re-cut: parity test, then the bench's headline leg under rocprofv3 (kernel stats)
TAG=${1:-q1}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python3 -m pytest tests/test_gpu_enc_split.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/enc_${TAG}_test.txt
cat gpurun_out/enc_${TAG}_test.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 8 --warmup 2 > $R/gpurun_out/enc_${TAG}_bench.jsonl 2> /tmp/prof_err.log
S=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$S" $R/gpurun_out/enc_${TAG}_kernel_stats.csv
head -16 $R/gpurun_out/enc_${TAG}_kernel_stats.csv | cut -c1-150
cut -c1-200 $R/gpurun_out/enc_${TAG}_bench.jsonl
