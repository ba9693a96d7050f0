#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for kv in VITS_LAT16_MAX_WAVES=768 VITS_LAT16_MAX_WAVES=2000 VITS_LAT16_MAX_WAVES=8000; do
  echo "== $kv"; env $kv python tools/pipe_bench.py --arith f16 --steps 30 --stage-one 2>&1 | grep -i "batch\|stage" | tail -4
done; done
