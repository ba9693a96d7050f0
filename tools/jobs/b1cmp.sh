#!/bin/bash
# A/B of an environment knob at batch 1: usage b1cmp.sh KNOB
K=${1:-VITS_WARM_W}
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/b1cmp; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
for a in f32 f16; do
for v in 1 0; do
env $K=$v python bench.py --batch 1 --arith $a --no-prof --no-cpu-baseline --no-extra-passes --steps 50 --warmup 5 > $O/b1_${a}_$v.json 2>/dev/null
python3 -c "
import json; d=json.load(open('$O/b1_${a}_$v.json')); print('$a $K=$v', round(d['ms_per_step'],3), 'ms')"
done; done
