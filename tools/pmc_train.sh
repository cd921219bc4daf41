#!/bin/bash
# LDS pressure / wait counters of the TRAINING step's kernels (tools/bench_train.py, 2 steps) and of configs[2..3] (tools/bench_other.py):
# the triage that found k_dec_cache's bank conflicts (profiles/r06/NOTES.md section 7), for the kernels bench.py's counter pass does not run.
#   gpurun -- 'bash tools/pmc_train.sh'   -> gpurun_out/pmc_train.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for what in train other; do
  if [ $what = train ]; then CMD="$R/tools/bench_train.py --steps 2"; else CMD="$R/tools/bench_other.py"; fi
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_t_a -- python3 $CMD > $R/gpurun_out/pmc_t_a.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_t_b -- python3 $CMD > $R/gpurun_out/pmc_t_b.log 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_t_a > $R/gpurun_out/pmc_${what}.txt
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_t_b >> $R/gpurun_out/pmc_${what}.txt
  rm -rf $R/gpurun_out/pmc_t_a $R/gpurun_out/pmc_t_b
done
