#!/bin/bash
# rollout knock-outs on one box: time (HIP events inside bench), FETCH_SIZE, power / clock.  usage: ko_rollout.sh "<flags>" ...
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R/real-routing-nco_amd/csrc
cp librrnco_hip.so /tmp/lib_good.so
# whatever happens (Ctrl-C, a failed step, a timeout): the product library comes back
trap 'cp /tmp/lib_good.so "$GRAFT_REPO_ROOT/real-routing-nco_amd/csrc/librrnco_hip.so"' EXIT INT TERM
for fl in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DRR_DEV_HEADLINE_ONLY $fl -c rr_decode.hip -o /tmp/dec_v.o 2>/dev/null || { echo "build failed: $fl"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC rr_env.o rr_sample.o rr_encoder.o /tmp/dec_v.o rr_train.o rr_train_dec.o rr_train_enc.o rr_train_nabdur.o rr_bign.o rr_matnet.o -o librrnco_hip.so
  cd $R
  echo "== [$fl]"
  python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other-configs --no-variants 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d['roofline']
print('   headline %.0f inst/s  step %.2f ms  rollout %.2f ms  frac %.4f' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['frac']), ' power_limited:', json.dumps(r.get('power_limited'))[:200])
"
  python3 tools/clock_power_sample.py 2>/dev/null | tail -3
  cd /tmp; rm -rf /tmp/pk
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pk -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs --no-variants > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pk | grep -A1 "k_rollout_w<7, 0, 0, true, true, false, false" | tail -1
  cd $R/real-routing-nco_amd/csrc
done
cp /tmp/lib_good.so librrnco_hip.so
