#!/bin/bash
st=$1
cd $GRAFT_REPO_ROOT/real-routing-nco_amd/csrc
cp librrnco_hip.so /tmp/lib_good.so
# whatever happens (Ctrl-C, a failed step, a timeout): the product library comes back
trap 'cp /tmp/lib_good.so "$GRAFT_REPO_ROOT/real-routing-nco_amd/csrc/librrnco_hip.so"' EXIT INT TERM
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DES_DEBUG_STAGE=$st -c rr_encoder.hip -o /tmp/enc_dbg.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC rr_env.o rr_sample.o /tmp/enc_dbg.o rr_decode.o rr_train.o rr_train_dec.o rr_train_enc.o rr_train_nabdur.o rr_bign.o rr_matnet.o -o librrnco_hip.so
(cd $GRAFT_REPO_ROOT && DBG_STAGE=$st python tools/debug_enc_split.py 2>&1 | grep -v amdgpu.ids | tail -${2:-6})
cp /tmp/lib_good.so librrnco_hip.so
