#!/bin/bash
# shape sweep of the 128 -> 64 stride-2 streaming transposed conv (VITS_CONVT16_R128), f16 serial + per-kernel profile of the default
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_ct; mkdir -p $O
for r in 1 2; do
for cfg in 0 411 410 211 210 421 420; do
  echo "== R128=$cfg: $(env VITS_CONVT16_R128=$cfg python tools/pipe_bench.py --arith f16 --steps 30 --mode serial 2>&1 | tail -1)"
done; done 2>&1 | tee $O/sweep.txt
# long identity fuzz while the box is there
timeout 1500 python tests/fuzz_identity.py --trials 900 --seed 2025 2>&1 | tail -3 | tee $O/fuzz.txt
