#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
VITS_LAT16_MIN_K=64 python tools/lat16_report.py | grep "k1\|sum of"
for rep in 1 2; do
for kv in VITS_LAT16_MIN_K=512 VITS_LAT16_MIN_K=64; do
  for a in f32 f16; do
    env $kv python bench.py --batch 1 --arith $a --no-cpu-baseline --no-extra-passes --no-prof --steps 40 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$kv $a batch 1 ms per utterance', round(d['ms_per_step'],4))"
  done
done; done
VITS_LAT16_MIN_K=64 timeout 300 python tests/fuzz_identity.py --trials 150 --seed 25 2>&1 | tail -2
