#!/bin/bash
# per-kernel times (HIP events, serialised) of the bench batch under a list of environment settings: bash tools/jobs/r3_kern.sh TAG "bench args" "FILTER" "ENV..." ...
TAG=$1; shift
ARGS=$1; shift
FILT=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$TAG; mkdir -p $O
i=0
for E in "" "$@"; do
  i=$((i+1))
  env $E python bench.py --single-pass --no-cpu-baseline --steps 10 --warmup 3 $ARGS > $O/run_$i.json 2> $O/run_$i.err
  python3 -c "
import json; d=json.load(open('$O/run_$i.json')); print('[%s]' % '$E', round(d['ms_per_step'],3), 'ms')
tot=0
for k in d['top_kernels']:
    if '$FILT' in k['kernel']:
        tot+=k['ms_per_step']; print('    %-22s %.3f ms  %5.0f TF  %.2f TB/s' % (k['kernel'], k['ms_per_step'], k['tflops'] or 0, (k['algorithmic_gbs'] or 0)/1e3))
print('    sum', round(tot,3))"
done
