#!/bin/bash
# usage: variants.sh "<flags A>" "<flags B>" ...   — builds rr_encoder.hip with each flag set on the GPU box and prints the encoder kernels' times
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R/real-routing-nco_amd/csrc
cp librrnco_hip.so /tmp/lib_good.so
# whatever happens (Ctrl-C, a failed step, a timeout): the product library comes back
trap 'cp /tmp/lib_good.so "$GRAFT_REPO_ROOT/real-routing-nco_amd/csrc/librrnco_hip.so"' EXIT INT TERM
i=0
for fl in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $fl -c rr_encoder.hip -o /tmp/enc_v.o 2>/dev/null || { echo "build failed: $fl"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC rr_env.o rr_sample.o /tmp/enc_v.o rr_decode.o rr_train.o rr_train_dec.o rr_train_enc.o rr_train_nabdur.o rr_bign.o rr_matnet.o -o librrnco_hip.so
  cd /tmp; rm -rf /tmp/pv_$i
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv_$i -o p -- python3 $R/tools/enc_time.py > /tmp/pv_$i.log 2>/dev/null
  S=$(find /tmp/pv_$i -name "*kernel_stats.csv" | head -1)
  echo "== [$fl] $(grep encoder /tmp/pv_$i.log)"
  python3 - "$S" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("k_enc_", "k_init_embed", "k_dec_cache")):
        print(f"   {n[:44]:44s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1000:9.1f} us")
PY
  cd $R/real-routing-nco_amd/csrc
done
cp /tmp/lib_good.so librrnco_hip.so
