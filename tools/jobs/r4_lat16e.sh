#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for b in 1 2 4 8; do
for kv in VITS_LAT16_MAX_WAVES=768 VITS_LAT16_MAX_WAVES=0 VITS_NO_LAT16=1; do
    env $kv python bench.py --batch $b --no-cpu-baseline --no-extra-passes --no-prof --steps 30 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('batch $b $kv ms per step', round(d['ms_per_step'],4))"
done; done; done
