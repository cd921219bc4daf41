#!/bin/bash
TAG=$1; VAR=$2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export RR_ROLLOUT_VARIANT=$VAR
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_${TAG}_a -- python3 $R/bench.py --steps 1 --warmup 0 --batch 128 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_a.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmc_${TAG}_b -- python3 $R/bench.py --steps 1 --warmup 0 --batch 128 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_b.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${TAG}_a | grep -A9 "k_rollout" > $R/gpurun_out/pmc_${TAG}.txt
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${TAG}_b | grep -A9 "k_rollout" >> $R/gpurun_out/pmc_${TAG}.txt
rm -rf $R/gpurun_out/pmc_${TAG}_a $R/gpurun_out/pmc_${TAG}_b
