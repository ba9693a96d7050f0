#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2_job4; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_arith16.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -15 $O/pytest.log
python tools/profile_dump.py 64 128 f16 > $O/dump_f16.txt 2>&1; head -24 $O/dump_f16.txt
python tools/profile_dump.py 1 128 f16 > $O/dump_f16_b1.txt 2>&1; head -8 $O/dump_f16_b1.txt
