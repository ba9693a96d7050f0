#!/bin/bash
# copies the final collection of round 5 from gpurun_out/ into profiles/ (tracked): ONE set per configuration
cd "$(dirname "$0")/.." || exit 1
for t in f32 f16 bf16 c5_bf16 c2_f16; do
  src=gpurun_out/r5_prof_$t
  [ -d $src ] || continue
  for f in pmc_traffic.json pmc_mfma.json kernel_stats.csv bench_under_rocprof.json default_schedule.json default_kernel_stats.csv; do
    [ -f $src/$f ] && cp $src/$f profiles/round5_${t}_$f
  done
done
for a in f32 f16; do [ -f gpurun_out/r5_final/b1_${a}_kernel_stats.csv ] && cp gpurun_out/r5_final/b1_${a}_kernel_stats.csv profiles/round5_b1_${a}_kernel_stats.csv; done
[ -f gpurun_out/r5_final/pytest_gpu.log ] && cp gpurun_out/r5_final/pytest_gpu.log profiles/round5_pytest_gpu.log
ls profiles | grep round5
