#!/bin/bash
# 16-bit modes: fused-pair tests, then batch-64 step time with and without an environment knob. usage: f16ab.sh KNOB=VALUE
KV=${1:-VITS_FUSE16_MAXC=64}
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/f16ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_arith16.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
for r in 1 2; do
for a in f16 bf16; do
python bench.py --arith $a --no-prof --no-cpu-baseline --no-extra-passes --steps 10 --warmup 3 > $O/a.json 2>/dev/null
env $KV python bench.py --arith $a --no-prof --no-cpu-baseline --no-extra-passes --steps 10 --warmup 3 > $O/b.json 2>/dev/null
python3 -c "
import json; a=json.load(open('$O/a.json')); b=json.load(open('$O/b.json')); print('$a default', round(a['ms_per_step'],3), 'ms   $KV', round(b['ms_per_step'],3), 'ms')"
done; done
