#!/bin/bash
# usage: tools/exp_stamps.sh <tag> [RR_STAGGER values...] : stamped-build phase tables of the headline rollout
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
for sg in "$@"; do
  echo "== RR_STAGGER=$sg" >> gpurun_out/stamps_$TAG.txt
  RR_STAGGER=$sg timeout 300 python3 tools/stamp_run.py 512 2>&1 | grep -v Warning | head -12 >> gpurun_out/stamps_$TAG.txt
done
cat gpurun_out/stamps_$TAG.txt
