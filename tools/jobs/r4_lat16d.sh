#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_lat16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x > $O/ops.log 2>&1; echo "ops exit $?"; tail -3 $O/ops.log
timeout 300 python tests/fuzz_identity.py --trials 250 --seed 24 2>&1 | tail -3
python tools/lat16_report.py | grep "t6\|sum of"
for kv in VITS_X=1 VITS_LAT16_MAX_WAVES=0 VITS_NO_LAT16=1 VITS_X=1 VITS_LAT16_MAX_WAVES=0 VITS_NO_LAT16=1; do
  for a in f32 f16; do
    env $kv python bench.py --batch 1 --arith $a --no-cpu-baseline --no-extra-passes --no-prof --steps 40 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$kv $a batch 1 ms per utterance', round(d['ms_per_step'],4))"
  done
done
