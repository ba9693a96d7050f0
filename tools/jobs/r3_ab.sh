#!/bin/bash
# A/B of an environment knob on the same box: bash tools/jobs/r3_ab.sh TAG "ENV=1 ..." [bench args]; prints instrumented + plain ms for both
TAG=${1:-ab}; shift
KNOB=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2; do
for v in base knob; do
  if [ $v = knob ]; then E="env $KNOB"; else E=""; fi
  $E python bench.py --no-cpu-baseline --no-sub-results --no-extra-passes --steps 20 --warmup 5 "$@" > $O/${v}_$rep.json 2> $O/${v}_$rep.err
  python3 -c "
import json; d=json.load(open('$O/${v}_$rep.json')); r=d['roofline']; print('$v $rep', round(d['ms_per_step'],3), 'ms instrumented', round(d.get('ms_per_step_without_kernel_events',0),3), 'ms plain | dominant', r['kernel'][:40], round(r['frac'],3), 'whole', round(d.get('frac_fp32_peak_whole_path', d.get('frac_mfma16_peak_whole_path',0)),3))"
done; done
