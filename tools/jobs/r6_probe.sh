#!/bin/bash
# round 6, first GPU call: (1) grid barrier vs kernel boundary micro-benchmark; (2) batch-1 knob sweep of the existing kernel choices
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_probe; mkdir -p $O
{
for g in "256 256" "256 512" "512 256" "128 256" "64 256"; do timeout 120 tools/bin/grid_barrier_micro $g; done
} > $O/grid_barrier.txt 2>&1
cat $O/grid_barrier.txt
timeout 1500 python tools/b1_knobs.py f16 "VITS_NO_FUSE16=1" "VITS_FUSE16_MAXC=128" "VITS_FUSE16_MAXC=64" "VITS_NO_RBBLOCK16=1" "VITS_ATT_NW=8" "VITS_NO_FLOW_FUSE=1" \
   "VITS_RB16_NARROW_MAX=0" "VITS_RB16_NARROW_MAX=1000" "VITS_RB16_SERIAL_MAX_FRAMES=0" "VITS_FLOW_NCW=1" "VITS_LN_TW=64" "VITS_FLOW_NARROW_MAX=0" "VITS_CONVT16_SPLIT_MAX=0" > $O/b1_knobs_f16.txt 2>&1
cat $O/b1_knobs_f16.txt
timeout 900 python tools/b1_knobs.py f32 "VITS_ATT_NW=8" "VITS_NO_LAT16=1" "VITS_LAT16_MAX_WAVES=2000" "VITS_NO_WN_FUSE=1" "VITS_RB_STREAMS=1" "VITS_NO_FUSE32=1" "VITS_NO_RBBLOCK32=1" > $O/b1_knobs_f32.txt 2>&1
cat $O/b1_knobs_f32.txt
