#!/bin/bash
# usage: tools/isa_dev.sh <tag> [extra hipcc flags] : ISA of the headline rollout instantiation -> /tmp/isa/k_<tag>.s
TAG=$1; shift
mkdir -p /tmp/isa
cd /root/repo/real-routing-nco_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRR_DEV_HEADLINE_ONLY "$@" -S --cuda-device-only rr_decode.hip -o /tmp/isa/$TAG.s 2>&1 | grep -v warning | grep -B2 -A8 "error" | head -30
awk '/^_Z11k_rollout_wILi7ELi0ELi0ELb1ELb1ELb0EEv4DecW9RolloutIOii:/{p=1} p{print} /\.end_amdhsa_kernel/{if(p){exit}}' /tmp/isa/$TAG.s > /tmp/isa/k_$TAG.s
# the 16-mixed instantiation (HALF) of the same kernel -> /tmp/isa/kh_<tag>.s
awk '/^_Z11k_rollout_wILi7ELi0ELi0ELb1ELb1ELb1EEv4DecW9RolloutIOii:/{p=1} p{print} /\.end_amdhsa_kernel/{if(p){exit}}' /tmp/isa/$TAG.s > /tmp/isa/kh_$TAG.s
echo "scratch ops: $(grep -c scratch_ /tmp/isa/k_$TAG.s)  lines: $(wc -l < /tmp/isa/k_$TAG.s)"
grep -n "s_barrier" /tmp/isa/k_$TAG.s | tr '\n' ' '; echo
