#!/bin/bash
# 64-column blocks of rbpair16 at C >= 128 on small grids: kernel-choice identity, small batches with and without them
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_rb16narrow; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_arith16.py -m gpu -q -x -k "kernel_choices" > $O/knob.log 2>&1; echo "knob exit $?"; tail -2 $O/knob.log
for rep in 1 2; do
for b in 1 2 4 8; do
for kv in VITS_RB16_NARROW_MAX=64 VITS_RB16_NARROW_MAX=0 VITS_RB16_NARROW_MAX=256; do
    env $kv python bench.py --batch $b --arith f16 --no-cpu-baseline --no-extra-passes --no-prof --steps 30 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('f16 batch $b $kv ms per step', round(d['ms_per_step'],4))"
done; done; done
