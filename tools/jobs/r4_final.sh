#!/bin/bash
# Final collection of a round-4 build: full GPU test suite, rocprofv3 stats + PMC passes (fp32 headline, f16, default schedule), every bench line.
# usage: bash tools/jobs/r4_final.sh vN     (results under gpurun_out/r4_vN/, profile-ready copies under gpurun_out/r4_vN/profiles_out/)
V=${1:-v1}
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_$V; mkdir -p $O $O/profiles_out
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -1
timeout 2400 python -m pytest tests -m gpu -q -rA > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log; grep -h "config 5 model\|16-bit modes vs fp32" $O/pytest.log | head
bash tools/jobs/profile.sh r4_${V}_prof "c3|b64|f32" > $O/prof.log 2>&1
bash tools/jobs/profile.sh r4_${V}_prof_f16 "c3|b64|f16" --arith f16 > $O/prof_f16.log 2>&1
for t in "" "_f16"; do
  for f in pmc_traffic.json pmc_mfma.json kernel_stats.csv bench_under_rocprof.json default_schedule.json default_kernel_stats.csv; do
    cp gpurun_out/r4_${V}_prof$t/$f profiles/round4_${V}${t}_$f; cp gpurun_out/r4_${V}_prof$t/$f $O/profiles_out/round4_${V}${t}_$f
  done
done
run() { name=$1; shift; S=$(date +%s); timeout 900 python bench.py "$@" > $O/profiles_out/round4_${V}_bench$name.json 2> $O/bench$name.err; echo "bench$name exit $? wall $(( $(date +%s) - S )) s"; python3 -c "
import json; d=json.load(open('$O/profiles_out/round4_${V}_bench$name.json')); r=d['roofline']; print('$name', round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],3),'ms plain', round(d.get('ms_per_step_without_kernel_events',0),3), r['bound'], round(r['frac'],3), r['kernel'][:50], 'traffic', r.get('traffic'), 'busy', r.get('mfma_busy_frac'), 'whole', (r.get('whole_step_traffic') or {}).get('ratio'))
ds=d.get('roofline_default_schedule')
if ds: print('   default schedule:', ds.get('available'), ds.get('kernel'), ds.get('frac'), 'overlap', ds.get('overlap_factor'))
for k,v in d.get('sub_results',{}).items():
    if isinstance(v,dict): print('   ', k, round(v['value']/1e6,2),'M/s', round(v['ms_per_step'],3),'ms', v['binding_roof'], round(v['frac_of_binding_roof'],3), 'serial', round((v.get('serial_calls') or {}).get('ms_per_step',0),3))
m=d.get('duration_boundary_margin')
if m: print('   margin: q6 outside', m.get('latents_outside_the_spline_interval_q6'), 'q8', {k:v for k,v in m.get('emulated_ggml_tables_q8',{}).items() if k!='note'})"; }
run ""
VITS_BENCH_FORCE_DIST=1 VITS_BENCH_LAUNCH=1 run _forcedist --steps 10 --warmup 3 --no-cpu-baseline --no-extra-passes
run _c3_f16 --arith f16 --steps 20 --warmup 5 --no-cpu-baseline
run _c3_bf16 --arith bf16 --steps 20 --warmup 5 --no-cpu-baseline
run _c5_f32 --workload c5 --steps 5 --warmup 2
run _c5_bf16 --workload c5 --arith bf16 --steps 5 --warmup 2 --no-cpu-baseline
run _c2_f32 --batch 1 --steps 30 --warmup 5 --no-cpu-baseline
run _c2_f16 --batch 1 --arith f16 --steps 30 --warmup 5 --no-cpu-baseline
