#!/bin/bash
# Instruction-cache counters of the bench step's kernels (is the straight-line 101 KB encoder block fetch-bound?).  One PMC pass.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_ic -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $R/gpurun_out/pmc_ic.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_ic > $R/gpurun_out/pmc_icache.txt
rm -rf $R/gpurun_out/pmc_ic
grep -A9 "k_rollout_w<7, 0, 0, true, true, false\|k_enc_block_w<7, true, false\|k_enc_ffn<7\|k_init_embed<7" $R/gpurun_out/pmc_icache.txt
