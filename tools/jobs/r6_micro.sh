#!/bin/bash
# round 6: phase stamps of the batch-1 stage-one kernels (DDS layer, attention) at 128 tokens
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_micro; mkdir -p $O
{
for d in 1 3 9; do timeout 120 tools/bin/dds_micro 128 1 $d; done
timeout 120 tools/bin/att_128
VITS_ATT_NW=8 timeout 120 tools/bin/att_128
} > $O/micro.txt 2>&1
cat $O/micro.txt
