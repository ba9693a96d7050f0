#!/bin/bash
# quick check of a build: GPU tests (fail fast), then un-instrumented step times of configs 2 and 3 in fp32 and f16
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/quick; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
for a in f32 f16; do
for b in 1 64; do
python bench.py --batch $b --arith $a --no-prof --no-cpu-baseline --no-extra-passes --steps $((b==1?50:10)) --warmup 5 > $O/b${b}_$a.json 2>/dev/null
python3 -c "
import json; d=json.load(open('$O/b${b}_$a.json')); print('$a batch $b', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,2), 'M samples/s')"
done; done
