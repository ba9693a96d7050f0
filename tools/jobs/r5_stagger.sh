#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_stagger; mkdir -p $O
for r in 1 2; do
for st in 0 50 100 150 250; do
  echo "== STAGGER=$st: $(env VITS_RB_STAGGER=$st python tools/pipe_bench.py --arith f16 --steps 30 2>&1 | tail -1)"
  echo "== STAGGER=$st one stream: $(env VITS_RB_STAGGER=$st VITS_RB_STREAMS=1 python tools/pipe_bench.py --arith f16 --steps 30 --mode serial 2>&1 | tail -1)"
done; done 2>&1 | tee $O/sweep.txt
