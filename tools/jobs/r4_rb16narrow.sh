#!/bin/bash
# small-grid variants of the 16-bit vocoder (64-column blocks of rbpair16 at C >= 128, unit split of the stride-8 upsamplers): kernel-choice
# identity, small batches with and without them
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_rb16narrow; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_arith16.py -m gpu -q -x -k "kernel_choices" > $O/knob.log 2>&1; echo "knob exit $?"; tail -2 $O/knob.log
timeout 300 python tests/fuzz_identity.py --trials 200 --seed 51 2>&1 | tail -2
for rep in 1 2; do
for b in 1 2 4 8; do
for kv in VITS_X=1 VITS_CONVT16_SPLIT_MAX=0 VITS_RB16_NARROW_MAX=0; do
    env $kv python bench.py --batch $b --arith f16 --no-cpu-baseline --no-extra-passes --no-prof --steps 30 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('f16 batch $b $kv ms per step', round(d['ms_per_step'],4))"
done; done; done
