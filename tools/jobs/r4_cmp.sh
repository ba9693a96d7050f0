#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python tools/pipe_bench.py --arith f16 --steps 16 2>&1 | grep batch
python tools/pipe_bench.py --arith f16 --steps 30 2>&1 | grep batch
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-passes --no-serving > /tmp/b.json 2>/dev/null
python3 -c "
import json; d=json.load(open('/tmp/b.json'))
for k,v in d['sub_results'].items():
    if isinstance(v,dict): print(k, round(v['ms_per_step'],3), 'serial', round((v.get('serial_calls') or {}).get('ms_per_step',0),3))"
python tools/pipe_bench.py --arith f16 --steps 16 2>&1 | grep batch
