#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_stream; mkdir -p $O
for b in rbb_k11_c32 rbb_k11_c32_rb2 rbb_k11_c32_ragged rbb_k11_c32_rb2dragged; do
  for t in 0 3 6; do
    echo -n "$b tiles=$t  "; VITS_RBB_STREAM_TILES=$t tools/bin/$b | tail -1
  done
done 2>&1 | tee $O/micro3.txt
timeout 1500 python3 tests/fuzz_identity.py --trials 300 --seed 61 2>&1 | tail -3 | tee $O/fuzz.txt
