#!/bin/bash
# MFMA-pipe utilisation, effective clock and LDS pressure of the rollout / encoder kernels at the full bench batch (both the
# default fp32-MFMA build and the opt-in split-operand bf16 build run inside one bench.py process).  Two PMC passes.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_m_a -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $R/gpurun_out/pmc_m_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_m_b -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $R/gpurun_out/pmc_m_b.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_m_a > $R/gpurun_out/pmc_mfma.txt
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_m_b >> $R/gpurun_out/pmc_mfma.txt
rm -rf $R/gpurun_out/pmc_m_a $R/gpurun_out/pmc_m_b
grep -A8 "k_rollout_w<7, 0, 0\|k_enc_mix<7\|k_enc_tail<7\|k_enc_kv\|k_nab_dist_family" $R/gpurun_out/pmc_mfma.txt
