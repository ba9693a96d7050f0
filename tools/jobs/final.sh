#!/bin/bash
# Final collection of a build: full GPU test suite, rocprofv3 stats + PMC passes (fp32 headline and f16), every bench line.
# usage: bash tools/jobs/final.sh vN
V=${1:-v2}
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2_$V; mkdir -p $O profiles
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest.log
bash tools/jobs/profile.sh r2_${V}_prof > $O/prof.log 2>&1
bash tools/jobs/profile.sh r2_${V}_prof_f16 --arith f16 > $O/prof_f16.log 2>&1
for t in "" "_f16"; do
  cp gpurun_out/r2_${V}_prof$t/pmc_traffic.json profiles/round2_${V}${t}_pmc_traffic.json
  cp gpurun_out/r2_${V}_prof$t/pmc_mfma.json profiles/round2_${V}${t}_pmc_mfma.json
  cp gpurun_out/r2_${V}_prof$t/kernel_stats.csv profiles/round2_${V}${t}_kernel_stats.csv
  cp gpurun_out/r2_${V}_prof$t/bench_under_rocprof.json profiles/round2_${V}${t}_bench_under_rocprof.json
done
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
