#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for a in f16 f32; do
for b in 1 4 16 64 256; do
  st=$(( b >= 64 ? 10 : 30 )); [ $a = f32 ] && [ $b -ge 64 ] && st=4
  python tools/pipe_bench.py --arith $a --batch $b --steps $st 2>&1 | grep batch | cut -c1-110
done; done
