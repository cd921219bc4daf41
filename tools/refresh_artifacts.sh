#!/bin/bash
# Round artefacts in one GPU call: the bench line (headline + BASELINE configs[2..4] + cpu_baseline), rocprofv3 kernel stats of the
# same command, HBM-side PMC traffic of the rollout, MFMA / VALU / LDS counters, training-step profile.  Run through gpurun; copy
# gpurun_out/<TAG>_* into profiles/rNN/ (bench.py reads profiles/r06/bench_pmc_hbm_traffic.json).
TAG=${1:-v1}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python3 bench.py > gpurun_out/bench_$TAG.jsonl 2> gpurun_out/bench_$TAG.err
tail -1 gpurun_out/bench_$TAG.jsonl | cut -c1-300
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o p -- python3 $R/bench.py --no-cpu-baseline --no-other-configs > $R/gpurun_out/bench_${TAG}_prof.jsonl 2> /tmp/prof_err.log
S=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$S" $R/gpurun_out/bench_${TAG}_kernel_stats.csv
head -14 $R/gpurun_out/bench_${TAG}_kernel_stats.csv | cut -c1-160
bash $R/tools/pmc_mfma.sh > /dev/null
cp $R/gpurun_out/pmc_mfma.txt $R/gpurun_out/bench_${TAG}_pmc_counters.txt
bash $R/tools/pmc_traffic_others.sh > /dev/null      # configs[2..4]'s dominant kernels (merged into the JSON by the next script)
bash $R/tools/pmc_traffic.sh          # (after the counter pass: its JSON also carries mfma_busy from pmc_mfma.txt)
cp $R/gpurun_out/bench_pmc_hbm_traffic.json $R/gpurun_out/bench_${TAG}_pmc_hbm_traffic.json
cat $R/gpurun_out/pmc_FETCH_SIZE.txt $R/gpurun_out/pmc_WRITE_SIZE.txt > $R/gpurun_out/bench_${TAG}_pmc_hbm_traffic.txt
cd $R
python3 tools/profile_train_step.py 2>&1 | grep -v Warning > gpurun_out/train_${TAG}_profile.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proft_$TAG -o p -- python3 $R/tools/bench_train.py --steps 2 > $R/gpurun_out/train_${TAG}.jsonl 2> /tmp/proft_err.log
S=$(find /tmp/proft_$TAG -name "*kernel_stats.csv" | head -1)
cp "$S" $R/gpurun_out/train_${TAG}_kernel_stats.csv
cd $R
python3 tools/bench_stepwise.py 2>/dev/null | tail -2 > gpurun_out/other_$TAG.jsonl
python3 tools/stepkernels_time.py 2>/dev/null | tail -5 >> gpurun_out/other_$TAG.jsonl
python3 tools/bench_train.py --problem rcvrp --steps 3 2>/dev/null | tail -1 >> gpurun_out/other_$TAG.jsonl
python3 tools/bench_train.py --problem rcvrptw --steps 2 --batch 512 2>/dev/null | tail -1 >> gpurun_out/other_$TAG.jsonl
python3 tools/profile_c4.py 2>&1 | grep -v Warning | tail -23 > gpurun_out/c4_${TAG}_profile.txt
PROBLEM=rcvrp python3 tools/profile_c4.py 2>&1 | grep -v Warning | tail -23 > gpurun_out/c3_${TAG}_profile.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs_$TAG -o p -- python3 $R/tools/bench_stepwise.py > /dev/null 2> /tmp/profs_err.log
S=$(find /tmp/profs_$TAG -name "*kernel_stats.csv" | head -1)
cp "$S" $R/gpurun_out/stepwise_${TAG}_kernel_stats.csv
