#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_lat16h; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_arith16.py -q -x -k "kernel_choices" 2>&1 | tail -3
timeout 1200 python3 tests/fuzz_identity.py --trials 200 --seed 77 2>&1 | tail -2
python3 tools/b1_knobs.py both "VITS_NO_LAT16H=1 VITS_RB_SUM3_BLOCK_ONLY=1 VITS_RB_SUM3_IN_ORDER=1" 2>&1 | tail -6 | tee $O/b1.txt
python3 tools/bN_knobs.py "1 2 4 8 64" "VITS_NO_LAT16H=1 VITS_RB_SUM3_BLOCK_ONLY=1 VITS_RB_SUM3_IN_ORDER=1" 2>&1 | tail -3 | tee $O/bN.txt
