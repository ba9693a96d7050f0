#!/bin/bash
# effective shader clock per kernel (GRBM_GUI_ACTIVE / duration) at batch 1 and batch 64
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/clock; mkdir -p $O
cd /tmp
for b in 1 64; do
rm -rf $O/c$b
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/c$b --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --batch $b --no-prof --no-cpu-baseline --no-extra-passes --steps 6 --warmup 3 > /dev/null 2> $O/err$b.txt
d=$(dirname $(find $O/c$b -name "*kernel_trace.csv" | head -1))
echo "batch $b"; python3 $GRAFT_REPO_ROOT/tools/clock.py $d | head -8
rm -rf $O/c$b
done
