#!/bin/bash
# pipelined f16 step under environment knobs, interleaved with the default, same box
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for kv in "VITS_X=0" "VITS_RB_STREAMS=1" "VITS_FLOW_NCW=1" "VITS_RBB_C64K11=1" "VITS_RBB_C128=0" "VITS_NO_FLOW_FUSE=1" "VITS_LRELU_COPY_MINC=64"; do
  env $kv python tools/pipe_bench.py --arith f16 --steps 30 2>&1 | grep batch | sed "s/^/$kv: /" | cut -c1-150
done; done
