#!/bin/bash
# A/B at batch 1 of an environment knob: usage b1ab.sh KNOB[=VALUE] [arith...] (default value 1); three alternating runs each
K=${1:-VITS_NO_ONESHOT16}; shift
case "$K" in *=*) KV=$K;; *) KV=$K=1;; esac
AR=${@:-f16}
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/b1ab; mkdir -p $O
for a in $AR; do
for r in 1 2 3; do
python bench.py --batch 1 --arith $a --no-prof --no-cpu-baseline --no-extra-passes --steps 100 --warmup 10 > $O/a.json 2>/dev/null
env $KV python bench.py --batch 1 --arith $a --no-prof --no-cpu-baseline --no-extra-passes --steps 100 --warmup 10 > $O/b.json 2>/dev/null
python3 -c "
import json; a=json.load(open('$O/a.json')); b=json.load(open('$O/b.json')); print('$a default', round(a['ms_per_step'],3), 'ms   $K=1', round(b['ms_per_step'],3), 'ms')"
done; done
