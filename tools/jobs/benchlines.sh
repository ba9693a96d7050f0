#!/bin/bash
# Every bench line of a build into profiles/round2_$V_bench*.json (uses the PMC artefacts already in profiles/ for this build and workload).
# usage: bash tools/jobs/benchlines.sh vN
V=${1:-v2}
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2_$V; mkdir -p $O profiles
run() { name=$1; shift; timeout 900 python bench.py "$@" > profiles/round2_${V}_bench$name.json 2> $O/bench$name.err; echo "bench$name exit $?"; python3 -c "
import json; d=json.load(open('profiles/round2_${V}_bench$name.json')); r=d['roofline']; print('$name', round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],2),'ms plain', round(d.get('value_without_kernel_events',0)/1e6,2), r['bound'], round(r['frac'],3), r['kernel'], 'traffic', r.get('traffic'), 'busy', r.get('mfma_busy_frac'))"; }
run "" --steps 20 --warmup 5
VITS_BENCH_FORCE_DIST=1 VITS_BENCH_LAUNCH=1 run _forcedist --steps 10 --warmup 3 --no-cpu-baseline --no-extra-passes
run _c3_f16 --arith f16 --steps 20 --warmup 5 --no-cpu-baseline
run _c3_bf16 --arith bf16 --steps 20 --warmup 5 --no-cpu-baseline
run _c5_f32 --workload c5 --steps 5 --warmup 2
run _c5_bf16 --workload c5 --arith bf16 --steps 5 --warmup 2 --no-cpu-baseline
run _c2_f32 --batch 1 --steps 30 --warmup 5 --no-cpu-baseline
run _c2_f16 --batch 1 --arith f16 --steps 30 --warmup 5 --no-cpu-baseline
cp -r profiles $O/profiles_out
