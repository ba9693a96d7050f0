#!/bin/bash
# one-off: the new per-handle knob test + the knob identity tests, and the list of PMC counters this box offers
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_avail; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_arith16.py tests/test_gpu_edge_and_scale.py -m gpu -q -x -k "knob or kernel_choices or whole_fp32" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
export TMPDIR=/tmp; cd /tmp
timeout 120 rocprofv3 --list-avail > $GRAFT_REPO_ROOT/$O/avail.txt 2>&1
grep -c . $GRAFT_REPO_ROOT/$O/avail.txt
