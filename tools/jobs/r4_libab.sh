#!/bin/bash
# same-box A/B of library builds with the pipelined step: bash tools/jobs/r4_libab.sh "pipe_bench args" a.so b.so ... (under vits.cpp_amd/csrc/ab/)
ARGS=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3; do
for L in "$@"; do
  echo -n "$L $rep: "; VITS_HIP_LIB=$PWD/vits.cpp_amd/csrc/ab/$L python tools/pipe_bench.py $ARGS 2>&1 | grep batch
done; done
