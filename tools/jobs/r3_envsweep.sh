#!/bin/bash
# plain (un-instrumented) step time of the bench batch under a list of environment settings, same box
# usage: bash tools/jobs/r3_envsweep.sh TAG "bench args" "ENV1=a ENV2=b" "ENV3=c" ...
TAG=$1; shift
ARGS=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$TAG; mkdir -p $O
i=0
for rep in 1 2; do
for E in "" "$@"; do
  i=$((i+1))
  env $E python bench.py --no-prof --no-cpu-baseline --no-sub-results --no-extra-passes --steps 20 --warmup 5 $ARGS > $O/run_$i.json 2> $O/run_$i.err
  python3 -c "
import json; d=json.load(open('$O/run_$i.json')); print('[%s]' % '$E', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,2), 'M/s')"
done; done
