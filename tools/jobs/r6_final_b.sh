#!/bin/bash
# part B: PMC / stats sets of the other 16-bit sub-results the bench line prints: c3 bf16, c5 bf16, c2 (batch 1) f16; batch-1 kernel stats and timelines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_final; mkdir -p $O
bash tools/jobs/profile.sh r6_prof_bf16 "c3|b64|bf16" --arith bf16 > $O/prof_bf16.log 2>&1; tail -2 $O/prof_bf16.log
bash tools/jobs/profile.sh r6_prof_c5_bf16 "c5|b8|bf16" --workload c5 --arith bf16 > $O/prof_c5_bf16.log 2>&1; tail -2 $O/prof_c5_bf16.log
bash tools/jobs/profile.sh r6_prof_c2_f16 "c2|b1|f16" --batch 1 --arith f16 > $O/prof_c2_f16.log 2>&1; tail -2 $O/prof_c2_f16.log
export TMPDIR=/tmp
cd /tmp
for a in f32 f16; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/b1_$a --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --batch 1 --arith $a --no-prof --no-cpu-baseline --no-extra-passes --steps 40 --warmup 5 > $GRAFT_REPO_ROOT/$O/bench_b1_$a.json 2> $GRAFT_REPO_ROOT/$O/err_b1_$a.txt
  cp $(find $GRAFT_REPO_ROOT/$O/b1_$a -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/b1_${a}_kernel_stats.csv
  rm -rf $GRAFT_REPO_ROOT/$O/b1_$a
done
cd $GRAFT_REPO_ROOT
for a in f32 f16; do bash tools/jobs/b1trace.sh $a > /dev/null 2>&1; cp gpurun_out/b1trace_$a/timeline.txt $O/b1_${a}_timeline.txt; done
ls gpurun_out/r6_prof_bf16 gpurun_out/r6_prof_c5_bf16 gpurun_out/r6_prof_c2_f16
