#!/bin/bash
# the 16-frame blocks of flow_couple16_kernel: kernel-choice identity (f16 / bf16), randomised identities, small batches with and without them
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_flownarrow; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_arith16.py -m gpu -q -x -k "kernel_choices or flow" > $O/knob.log 2>&1; echo "knob exit $?"; tail -2 $O/knob.log
timeout 300 python tests/fuzz_identity.py --trials 250 --seed 31 2>&1 | tail -2
for rep in 1 2; do
for b in 1 2 4 8 16; do
for kv in VITS_FLOW_NARROW_MAX=96 VITS_FLOW_NARROW_MAX=0 VITS_FLOW_NARROW_MAX=400; do
    env $kv python bench.py --batch $b --arith f16 --no-cpu-baseline --no-extra-passes --no-prof --steps 30 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('f16 batch $b $kv ms per step', round(d['ms_per_step'],4))"
done; done; done
