#!/bin/bash
# copies the final collection of round 6 from gpurun_out/ into profiles/ (tracked): ONE set per configuration
cd "$(dirname "$0")/.." || exit 1
for t in f32 f16 bf16 c5_bf16 c2_f16 f32split; do
  src=gpurun_out/r6_prof_$t
  [ -d $src ] || continue
  for f in pmc_traffic.json pmc_mfma.json kernel_stats.csv bench_under_rocprof.json default_schedule.json default_kernel_stats.csv; do
    [ -f $src/$f ] && cp $src/$f profiles/round6_${t}_$f
  done
done
for a in f32 f16; do
  [ -f gpurun_out/r6_final/b1_${a}_kernel_stats.csv ] && cp gpurun_out/r6_final/b1_${a}_kernel_stats.csv profiles/round6_b1_${a}_kernel_stats.csv
  [ -f gpurun_out/r6_final/b1_${a}_timeline.txt ] && cp gpurun_out/r6_final/b1_${a}_timeline.txt profiles/round6_b1_${a}_timeline.txt
done
for f in pytest_gpu.log grid_barrier_micro.txt split_micro.txt dropin_latency.txt; do [ -f gpurun_out/r6_final/$f ] && cp gpurun_out/r6_final/$f profiles/round6_$f; done
for f in gpurun_out/r6_final/lines/bench*.json; do [ -f $f ] && tail -1 $f > profiles/round6_$(basename $f); done
ls profiles | grep round6
