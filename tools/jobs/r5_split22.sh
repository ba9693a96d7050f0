#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_split22; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_arith16.py -m gpu -q -x -k "kernel_choices" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
for r in 1 2; do
for v in 0 1 2 3; do
  echo "== SPLIT22=$v f16: $(env VITS_RB16_SPLIT22=$v python tools/pipe_bench.py --arith f16 --steps 30 2>&1 | tail -1)"
done
for v in 0 3; do
  echo "== SPLIT22=$v bf16: $(env VITS_RB16_SPLIT22=$v python tools/pipe_bench.py --arith bf16 --steps 30 2>&1 | tail -1)"
done
done 2>&1 | tee $O/sweep.txt
