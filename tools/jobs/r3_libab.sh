#!/bin/bash
# same-box A/B of library builds: bash tools/jobs/r3_libab.sh TAG "bench args" a.so b.so ... (paths relative to vits.cpp_amd/csrc/ab/); two interleaved rounds, plain ms
TAG=$1; shift
ARGS=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2 3; do
for L in "$@"; do
  VITS_HIP_LIB=$PWD/vits.cpp_amd/csrc/ab/$L python bench.py --no-cpu-baseline --no-sub-results --no-extra-passes --steps 20 --warmup 5 $ARGS > $O/${L}_$rep.json 2> $O/${L}_$rep.err
  python3 -c "
import json; d=json.load(open('$O/${L}_$rep.json')); print('$L $rep', round(d['ms_per_step'],3), 'ms instrumented', round(d.get('ms_per_step_without_kernel_events',0),3), 'ms plain')"
done; done
