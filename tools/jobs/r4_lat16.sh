#!/bin/bash
# conv_lat16_kernel (16 x 16 tiles on v_mfma_f32_16x16x4_f32): operator tests, kernel-choice identity, randomised identities, per-shape times, batch 1 ... 16 with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_lat16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x > $O/ops.log 2>&1; echo "ops exit $?"; tail -2 $O/ops.log
timeout 900 python -m pytest tests/test_gpu_edge_and_scale.py tests/test_gpu_arith16.py -m gpu -q -x -k "knob or kernel_choices" > $O/knob.log 2>&1; echo "knob exit $?"; tail -2 $O/knob.log
timeout 300 python tests/fuzz_identity.py --trials 300 --seed 26 2>&1 | tail -2
python tools/lat16_report.py | tail -1
for rep in 1 2; do
for b in 1 2 4 8 16; do
for kv in VITS_X=1 VITS_NO_LAT16=1; do
    env $kv python bench.py --batch $b --no-cpu-baseline --no-extra-passes --no-prof --steps 30 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('batch $b $kv ms per step', round(d['ms_per_step'],4))"
done; done; done
for kv in VITS_X=1 VITS_NO_LAT16=1; do env $kv python bench.py --batch 1 --arith f16 --no-cpu-baseline --no-extra-passes --no-prof --steps 40 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$kv f16 batch 1 ms per utterance', round(d['ms_per_step'],4))"; done
