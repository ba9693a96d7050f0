#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_exact; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "ggml or emulated" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -30 $O/pytest.log
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_edge_and_scale.py -m gpu -q -k "forced_dist or gather" > $O/pytest2.log 2>&1; echo "pytest2 exit $?"; tail -5 $O/pytest2.log
