#!/bin/bash
# per-kernel durations of the whole-resblock kernels on ONE stream (VITS_RB_STREAMS=1: no overlap, durations add up), one tile per block against segments of 2 / 4 / 8 tiles (batch 64 x 128 ids, f16)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_stream; mkdir -p $O
export TMPDIR=/tmp
export VITS_RB_STREAMS=1
python3 tools/bN_knobs.py "64" "VITS_RBB_STREAM_TILES=0" "VITS_RBB_STREAM_TILES=2" "VITS_RBB_STREAM_TILES=4" "VITS_RBB_STREAM_TILES=8" "VITS_RBB_STREAM_TILES=0" 2>&1 | tee $O/ab64_serial.txt
cd /tmp
for t in 0 2 4 8; do
  export VITS_RBB_STREAM_TILES=$t
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/st_$t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bN_knobs.py 64 > /dev/null 2>&1
  f=$(find $O/st_$t -name "*kernel_stats.csv" | head -1)
  cp $f $O/kernel_stats_tiles$t.csv; rm -rf $O/st_$t
done
