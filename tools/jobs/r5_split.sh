#!/bin/bash
# in-call split (VITS_SPLIT_MIN_BATCH / VITS_SPLIT_FIRST_PCT): identity tests, then serial ms per batch vs split share, f16 and f32
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_split; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log
for a in f16 bf16 f32; do
  st=30; [ $a = f32 ] && st=8
  for cfg in "VITS_SPLIT_MIN_BATCH=0" "VITS_SPLIT_FIRST_PCT=50" "VITS_SPLIT_FIRST_PCT=40" "VITS_SPLIT_FIRST_PCT=30" "VITS_SPLIT_FIRST_PCT=25" "VITS_SPLIT_FIRST_PCT=60"; do
    echo "== $a $cfg: $(env $cfg python tools/pipe_bench.py --arith $a --steps $st 2>&1 | tail -1)"
  done
done 2>&1 | tee $O/sweep.txt
