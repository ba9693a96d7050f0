#!/bin/bash
# part C: every bench line of the round on the final build, with the round-5 artefacts in profiles/
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5_final; mkdir -p $O/lines
timeout 900 python -m pytest tests/test_gpu_arith16.py -m gpu -q -k "torch" > $O/pytest_arith_fixtures.log 2>&1; echo "fixture tests exit $?"; tail -2 $O/pytest_arith_fixtures.log
run() { name=$1; shift; S=$(date +%s); timeout 900 python3 bench.py "$@" > $O/lines/bench$name.json 2> $O/lines/bench$name.err; echo "bench$name exit $? wall $(( $(date +%s) - S )) s bytes $(tail -1 $O/lines/bench$name.json | wc -c)"; cp bench_detail.json $O/lines/bench_detail$name.json 2>/dev/null; }
run "" --gpus 1 --steps 20 --warmup 5
VITS_BENCH_FORCE_DIST=1 VITS_BENCH_LAUNCH=1 run _forcedist --steps 10 --warmup 3 --no-cpu-baseline --no-extra-passes --no-sub-results
run _c3_f16 --arith f16 --steps 20 --warmup 5 --no-cpu-baseline
run _c3_bf16 --arith bf16 --steps 20 --warmup 5 --no-cpu-baseline
run _c5_f32 --workload c5 --steps 5 --warmup 2 --no-cpu-baseline
run _c5_bf16 --workload c5 --arith bf16 --steps 5 --warmup 2 --no-cpu-baseline
run _c2_f32 --batch 1 --steps 30 --warmup 5 --no-cpu-baseline
run _c2_f16 --batch 1 --arith f16 --steps 30 --warmup 5 --no-cpu-baseline
tail -1 $O/lines/bench.json
