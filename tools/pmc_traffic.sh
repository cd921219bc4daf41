#!/bin/bash
# HBM traffic of the rollout kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), full bench batch.
# Writes gpurun_out/pmc_{FETCH,WRITE}_SIZE.txt and gpurun_out/bench_pmc_hbm_traffic.json (copy the latter to
# profiles/r06/: bench.py reads roofline.traffic from it when its source hash matches the rollout's sources).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $R/gpurun_out/pmc_$C.log 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_$C > $R/gpurun_out/pmc_$C.txt
  rm -rf $R/gpurun_out/pmc_$C
done
cat $R/gpurun_out/pmc_FETCH_SIZE.txt $R/gpurun_out/pmc_WRITE_SIZE.txt
python3 $R/tools/pmc_traffic_json.py $R/gpurun_out/pmc_FETCH_SIZE.txt $R/gpurun_out/pmc_WRITE_SIZE.txt $R/gpurun_out/pmc_mfma.txt > $R/gpurun_out/bench_pmc_hbm_traffic.json
cat $R/gpurun_out/bench_pmc_hbm_traffic.json
