#!/bin/bash
# effective shader clock per kernel (GRBM_GUI_ACTIVE summed over the 8 XCDs / duration / 8) for a bench configuration
# usage: clock.sh [bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/clock; mkdir -p $O
cd /tmp
rm -rf $O/c
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/c --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-prof --no-cpu-baseline --no-extra-passes --steps 6 --warmup 3 "$@" > /dev/null 2> $O/err.txt
d=$(dirname $(find $O/c -name "*kernel_trace.csv" | head -1))
python3 $GRAFT_REPO_ROOT/tools/clock.py $d | head -10
rm -rf $O/c
