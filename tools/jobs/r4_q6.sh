#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_q6; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -q -x > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -15 $O/pytest.log
python tools/pipe_bench.py --arith f16 --steps 30 2>&1 | grep batch
python tools/pipe_bench.py --arith f32 --steps 8 2>&1 | grep batch
