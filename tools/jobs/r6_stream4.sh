#!/bin/bash
# in-situ A/B of the shipped policy (segments per shape; C = 64 / k = 11 as a whole-resblock kernel where segments apply) against one tile per block + pairs
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_stream; mkdir -p $O
OLD="VITS_RBB_STREAM_TILES=0 VITS_RBB_C64K11=0"
python3 tools/bN_knobs.py "64 16 8" "$OLD" "" "$OLD" "" "$OLD" 2>&1 | tee $O/ab64_shipped.txt
for i in 1 2; do
echo -n "pipelined old: "; VITS_RBB_STREAM_TILES=0 VITS_RBB_C64K11=0 python3 tools/pipe_bench.py --arith f16 --steps 30 --mode pipelined 2>&1 | tail -1
echo -n "pipelined new: "; python3 tools/pipe_bench.py --arith f16 --steps 30 --mode pipelined 2>&1 | tail -1
echo -n "pipelined bf16 old: "; VITS_RBB_STREAM_TILES=0 VITS_RBB_C64K11=0 python3 tools/pipe_bench.py --arith bf16 --steps 30 --mode pipelined 2>&1 | tail -1
echo -n "pipelined bf16 new: "; python3 tools/pipe_bench.py --arith bf16 --steps 30 --mode pipelined 2>&1 | tail -1
done 2>&1 | tee $O/pipe_shipped.txt
for i in 1 2; do
VITS_RBB_STREAM_TILES=0 VITS_RBB_C64K11=0 python3 bench.py --workload c5 --arith bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-sub-results --no-prof --no-extra-passes 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 bf16 old', d['ms_per_step'])"
python3 bench.py --workload c5 --arith bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-sub-results --no-prof --no-extra-passes 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('c5 bf16 new', d['ms_per_step'])"
done 2>&1 | tee $O/c5_shipped.txt
VITS_KNOB_ARITH=f16 python3 tools/knob_identity.py; VITS_KNOB_ARITH=f16 VITS_RBB_STREAM_MIN_BLOCKS=1 python3 tools/knob_identity.py
