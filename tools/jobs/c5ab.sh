#!/bin/bash
# config 5 (two bf16 models x 8 x 1024 ids) step time with and without an environment knob. usage: c5ab.sh KNOB=VALUE [arith]
KV=${1:-VITS_ATT_DIRECT_MAXT=100000}; A=${2:-bf16}
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/c5ab; mkdir -p $O
for r in 1 2; do
python bench.py --workload c5 --arith $A --no-prof --no-cpu-baseline --no-extra-passes --steps 5 --warmup 2 > $O/a.json 2>/dev/null
env $KV python bench.py --workload c5 --arith $A --no-prof --no-cpu-baseline --no-extra-passes --steps 5 --warmup 2 > $O/b.json 2>/dev/null
python3 -c "
import json; a=json.load(open('$O/a.json')); b=json.load(open('$O/b.json')); print('c5 $A default', round(a['ms_per_step'],2), 'ms   $KV', round(b['ms_per_step'],2), 'ms')"
done
