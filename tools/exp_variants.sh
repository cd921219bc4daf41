#!/bin/bash
# usage: tools/exp_variants.sh <tag> <variant names...> : stamped phase tables of the headline rollout for several diagnostic builds
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
for v in "$@"; do
  echo "== variant $v" >> gpurun_out/variants_$TAG.txt
  RR_STAMP_LIB=$R/real-routing-nco_amd/csrc/librrnco_hip_stamp_$v.so timeout 300 python3 tools/stamp_run.py 512 2>&1 | grep -v "Warning\|amdgpu.ids" | head -12 >> gpurun_out/variants_$TAG.txt
done
cat gpurun_out/variants_$TAG.txt
