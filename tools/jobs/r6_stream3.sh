#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_stream; mkdir -p $O
export VITS_RBB_C64K11=1
for b in rbb_k11_c64 rbb_k3_c64 rbb_k3_c128; do
  for t in 0 2 3 4 6; do
    echo -n "$b tiles=$t  "; VITS_RBB_STREAM_TILES=$t tools/bin/$b | tail -1
  done
done 2>&1 | tee $O/micro2.txt
