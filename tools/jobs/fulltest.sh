#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/fulltest; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -15 $O/pytest.log
