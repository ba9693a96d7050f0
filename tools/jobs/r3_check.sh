#!/bin/bash
# round 3 check: GPU tests, then the driver's default bench command (wall time recorded) and the forced-RCCL line
# usage: bash tools/jobs/r3_check.sh TAG [pytest args]
TAG=${1:-r3}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$TAG; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q "$@" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log
S=$(date +%s)
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $? wall $(( $(date +%s) - S )) s"
python3 - <<PY
import json
d=json.load(open('$O/bench.json'))
r=d['roofline']
print('headline', round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],2),'ms plain', round(d.get('value_without_kernel_events',0)/1e6,2), r['bound'], round(r['frac'],3), r['kernel'])
for k,v in d.get('sub_results',{}).items():
    if isinstance(v,dict): print(k, round(v['value']/1e6,2),'M/s', round(v['ms_per_step'],3),'ms', v['binding_roof'], round(v['frac_of_binding_roof'],3))
print('sub wall', d.get('sub_results',{}).get('wall_s'))
print('margin', d.get('duration_boundary_margin'))
print('cpu', {k:(v if not isinstance(v,dict) else v.get('value')) for k,v in d.get('cpu_baseline',{}).items() if k in ('value','cores','one_thread','at_reference_thread_rule')})
PY
S=$(date +%s)
VITS_BENCH_FORCE_DIST=1 VITS_BENCH_LAUNCH=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-passes > $O/bench_forcedist.json 2> $O/bench_forcedist.err; echo "forcedist exit $? wall $(( $(date +%s) - S )) s"
python3 -c "
import json; d=json.load(open('$O/bench_forcedist.json')); print('forcedist', round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],2),'ms')"
