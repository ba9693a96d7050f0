#!/bin/bash
# two-chain flow (VITS_FLOW_CHAINS) identity + A/B, and the in-call split at batch 128 / 96
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_flow; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_arith16.py -m gpu -q -x -k "kernel_choices" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_edge_and_scale.py -m gpu -q -x > $O/pytest2.log 2>&1; echo "pytest2 exit $?"; tail -3 $O/pytest2.log
{
for r in 1 2; do
for a in f16 bf16; do
  for cfg in "VITS_FLOW_CHAINS=1" "VITS_FLOW_CHAINS=2"; do
    echo "== $cfg: $(env $cfg python tools/pipe_bench.py --arith $a --steps 30 2>&1 | tail -1)"
  done
done; done
for b in 128 96; do
  for cfg in "VITS_SPLIT_MIN_BATCH=0" "VITS_SPLIT_MIN_BATCH=64"; do
    echo "== b$b $cfg: $(env $cfg python tools/pipe_bench.py --arith f16 --steps 16 --batch $b --mode serial 2>&1 | tail -1)"
  done
done
} 2>&1 | tee $O/ab.txt
