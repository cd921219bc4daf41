#!/bin/bash
# usage: tools/build_full_variant.sh <name> [flags] : full (all instantiations, no stamps) library real-routing-nco_amd/csrc/librrnco_hip_<name>.so
NAME=$1; shift
cd /root/repo/real-routing-nco_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -c rr_decode.hip -o /tmp/rr_decode_full_$NAME.o 2>&1 | grep -v warning | grep -A5 error
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/rr_decode_full_$NAME.o rr_encoder.o rr_env.o rr_sample.o rr_train.o rr_train_dec.o rr_train_enc.o rr_train_nabdur.o rr_bign.o rr_matnet.o -o librrnco_hip_$NAME.so && echo built $NAME
