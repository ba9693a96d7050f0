#!/bin/bash
# bash tools/jobs/r4_knob.sh "KNOB=VALUE" : identity hash (f16, bf16) default vs knob, then pipelined f16 timing interleaved
KV=$1
cd "$GRAFT_REPO_ROOT" || exit 1
for a in f16 bf16; do
  A=$(VITS_KNOB_ARITH=$a python tools/knob_identity.py 2>&1 | tail -1); B=$(env $KV VITS_KNOB_ARITH=$a python tools/knob_identity.py 2>&1 | tail -1)
  echo "$a identity: default $A  $KV $B  $([ "$A" = "$B" ] && echo SAME || echo DIFFERENT)"
done
for rep in 1 2 3; do
  python tools/pipe_bench.py --arith f16 --steps 30 2>&1 | grep batch | sed 's/^/default: /' | cut -c1-120
  env $KV python tools/pipe_bench.py --arith f16 --steps 30 2>&1 | grep batch | sed "s/^/$KV: /" | cut -c1-130
done
