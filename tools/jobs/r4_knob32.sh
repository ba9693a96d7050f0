#!/bin/bash
# bash tools/jobs/r4_knob32.sh "KNOB=VALUE": fp32 identity hash default vs knob, then fp32 step times (serial calls) interleaved
KV=$1
cd "$GRAFT_REPO_ROOT" || exit 1
A=$(python tools/knob_identity.py 2>&1 | tail -1); B=$(env $KV python tools/knob_identity.py 2>&1 | tail -1)
echo "f32 identity: default $A  $KV $B  $([ "$A" = "$B" ] && echo SAME || echo DIFFERENT)"
for rep in 1 2 3; do
  python tools/pipe_bench.py --arith f32 --steps 8 --mode serial 2>&1 | grep batch | sed 's/^/default: /' | cut -c1-90
  env $KV python tools/pipe_bench.py --arith f32 --steps 8 --mode serial 2>&1 | grep batch | sed "s/^/$KV: /" | cut -c1-110
done
