#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2_job6; mkdir -p $O
VITS_T16_TILE0=${T0:-0} python tools/profile_dump.py 64 128 f16 > $O/dump_f16.txt 2>&1; grep "resblock_conv" $O/dump_f16.txt | sort -k9 | head -80
