#!/bin/bash
# MFMA-pipe utilisation and LDS pressure of the training-step kernels (tools/bench_train.py, one step).  Two PMC passes.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_t_a -- python3 $R/tools/bench_train.py --steps 1 > $R/gpurun_out/pmc_t_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_t_b -- python3 $R/tools/bench_train.py --steps 1 > $R/gpurun_out/pmc_t_b.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_t_a > $R/gpurun_out/pmc_train.txt
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_t_b >> $R/gpurun_out/pmc_train.txt
rm -rf $R/gpurun_out/pmc_t_a $R/gpurun_out/pmc_t_b
grep -A9 "k_mlp_wgrad\|k_dec_attn_bwd\|k_mlp_rows<1>\|k_dec_logit_bwd" $R/gpurun_out/pmc_train.txt | cut -c1-200
