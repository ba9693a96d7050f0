#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for kv in "VITS_X=0" "VITS_NBUF=2" "VITS_NBUF=3" "VITS_MIN_BLOCKS=512" "VITS_MIN_BLOCKS=2048" "VITS_NARROW_K1=0" "VITS_ATT_SHORT=0"; do
  env $kv python tools/pipe_bench.py --arith f16 --steps 20 --stage-one 2>&1 | grep batch | sed "s/^/$kv: /" | sed 's/f16 batch 64 x 128: //' | cut -c1-150
done; done
