#!/bin/bash
# part C: every bench line of the round on the final build, with the round-6 artefacts in profiles/; the micro-benchmarks the design quotes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_final; mkdir -p $O/lines
run() { name=$1; shift; S=$(date +%s); timeout 900 python3 bench.py "$@" > $O/lines/bench$name.json 2> $O/lines/bench$name.err; echo "bench$name exit $? wall $(( $(date +%s) - S )) s bytes $(tail -1 $O/lines/bench$name.json | wc -c)"; cp bench_detail.json $O/lines/bench_detail$name.json 2>/dev/null; }
run "" --gpus 1 --steps 20 --warmup 5
VITS_BENCH_FORCE_DIST=1 VITS_BENCH_LAUNCH=1 run _forcedist --steps 10 --warmup 3 --no-cpu-baseline --no-extra-passes --no-sub-results
run _c3_f32split --arith f32split --steps 20 --warmup 5 --no-cpu-baseline --no-sub-results
run _c3_f16 --arith f16 --steps 20 --warmup 5 --no-cpu-baseline --no-sub-results
run _c3_bf16 --arith bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-sub-results
run _c5_f32 --workload c5 --steps 5 --warmup 2 --no-cpu-baseline --no-sub-results
run _c5_bf16 --workload c5 --arith bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-sub-results
run _c2_f32 --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-sub-results
run _c2_f16 --batch 1 --arith f16 --steps 30 --warmup 5 --no-cpu-baseline --no-sub-results
{ for g in "256 256" "64 256"; do timeout 120 tools/bin/grid_barrier_micro $g; done; } > $O/grid_barrier_micro.txt 2>&1
timeout 300 tools/bin/split_micro > $O/split_micro.txt 2>&1
timeout 300 python tools/dropin_latency.py > $O/dropin_latency.txt 2>&1
tail -1 $O/lines/bench.json
