#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for a in f16 bf16; do
  A=$(VITS_KNOB_ARITH=$a python tools/knob_identity.py 2>&1 | tail -1); B=$(VITS_NO_CONVT16L=1 VITS_KNOB_ARITH=$a python tools/knob_identity.py 2>&1 | tail -1)
  echo "$a identity: new lines kernel $A  GEMM-tile path $B  $([ "$A" = "$B" ] && echo SAME || echo DIFFERENT)"
done
bash tools/jobs/r4_libab.sh "--arith f16 --steps 30" ctold.so ctnew.so 2>&1 | cut -c1-120
for L in ctold.so ctnew.so; do echo -n "$L c5 bf16: "; VITS_HIP_LIB=$PWD/vits.cpp_amd/csrc/ab/$L python tools/pipe_bench.py --arith bf16 --batch 8 --ids 1024 --steps 10 2>&1 | grep batch | cut -c1-110; done
