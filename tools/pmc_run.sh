#!/bin/bash
# usage: tools/pmc_run.sh <tag> <variant> ; collects two PMC passes of a small bench run
TAG=$1; VAR=$2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# (RR_ROLLOUT_VARIANT left the library in round 4)
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_${TAG}_a -- python3 $R/bench.py --steps 1 --warmup 0 --batch 128 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCC_HIT TCC_MISS --output-format csv -d $R/gpurun_out/pmc_${TAG}_b -- python3 $R/bench.py --steps 1 --warmup 0 --batch 128 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_b.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${TAG}_a > $R/gpurun_out/pmc_${TAG}.txt
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${TAG}_b >> $R/gpurun_out/pmc_${TAG}.txt
rm -rf $R/gpurun_out/pmc_${TAG}_a $R/gpurun_out/pmc_${TAG}_b
