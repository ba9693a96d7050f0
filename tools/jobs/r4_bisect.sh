#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
show() { python3 -c "
import json; d=json.load(open('/tmp/b.json')); v=d['sub_results']['c3_f16']; print('$1', round(v['ms_per_step'],3), 'serial', round(v['serial_calls']['ms_per_step'],3))"; }
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serving > /tmp/b.json 2>/dev/null; show "default"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serving --no-extra-passes > /tmp/b.json 2>/dev/null; show "no-extra"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serving --no-prof > /tmp/b.json 2>/dev/null; show "no-prof"
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serving --no-prof --no-extra-passes > /tmp/b.json 2>/dev/null; show "no-prof-no-extra"
python tools/sub_dbg.py all 2>&1 | grep c3_f16
