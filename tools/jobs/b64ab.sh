#!/bin/bash
# batch-64 fp32 step time under several settings of one environment knob. usage: b64ab.sh KNOB V1 V2 ...
K=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/b64ab; mkdir -p $O
for r in 1 2; do
python bench.py --no-prof --no-cpu-baseline --no-extra-passes --steps 10 --warmup 3 > $O/a.json 2>/dev/null
python3 -c "
import json; a=json.load(open('$O/a.json')); print('default', round(a['ms_per_step'],2), 'ms')"
for v in "$@"; do
env $K=$v python bench.py --no-prof --no-cpu-baseline --no-extra-passes --steps 10 --warmup 3 > $O/b.json 2>/dev/null
python3 -c "
import json; b=json.load(open('$O/b.json')); print('$K=$v', round(b['ms_per_step'],2), 'ms')"
done; done
