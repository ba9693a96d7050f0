#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2_job8; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round2.py tests/test_gpu_model.py tests/test_gpu_streaming.py -m gpu -q -x -k "fused or oracle or golden or windowed or taps" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -8 $O/pytest.log
python tools/profile_dump.py 64 128 f32 > $O/dump_f32.txt 2>&1; head -30 $O/dump_f32.txt
VITS_NO_FUSE32=1 python tools/profile_dump.py 64 128 f32 > $O/dump_f32_nofuse.txt 2>&1; head -3 $O/dump_f32_nofuse.txt
