#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for kv in VITS_X=1 VITS_NO_LAT16=1 VITS_X=1 VITS_NO_LAT16=1; do
  for a in f32 f16; do
    env $kv python bench.py --batch 1 --arith $a --no-cpu-baseline --no-extra-passes --no-prof --steps 40 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('$kv $a batch 1 ms per utterance', round(d['ms_per_step'],4))"
  done
done
for kv in VITS_X=1 VITS_NO_LAT16=1; do
env $kv python bench.py --batch 1 --no-cpu-baseline --no-extra-passes --no-sub-results --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); print('$kv instrumented', round(d['ms_per_step'],3))
for k in d['top_kernels']:
    if '|t6|' in k['kernel'] or '|t5|' in k['kernel']: print('   ', k['kernel'], 'ms/step', round(k['ms_per_step'],4), 'calls', k['calls_per_step'], 'us per call', round(1e3*k['ms_per_step']/k['calls_per_step'],1))"
done
timeout 300 python tests/fuzz_identity.py --trials 200 --seed 22 2>&1 | tail -2
