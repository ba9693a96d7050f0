#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2_job5; mkdir -p $O
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name exit $?"; python3 -c "
import json; d=json.load(open('$O/$name.json')); print('$name', round(d['value']/1e6,2),'M samples/s', round(d['ms_per_step'],2),'ms', 'plain', round(d.get('value_without_kernel_events',0)/1e6,2), d['roofline']['bound'], round(d['roofline']['frac'],3), d['roofline']['kernel'], 'cpu', d.get('cpu_baseline',{}).get('value'))"; }
run c5_f32 --workload c5 --steps 5 --warmup 2
run c5_bf16 --workload c5 --arith bf16 --steps 5 --warmup 2 --no-cpu-baseline
run c3_f16 --arith f16 --steps 10 --warmup 3 --no-cpu-baseline
run c3_bf16 --arith bf16 --steps 10 --warmup 3 --no-cpu-baseline
run c2_f32 --batch 1 --steps 30 --warmup 5 --no-cpu-baseline
run c2_f16 --batch 1 --arith f16 --steps 30 --warmup 5 --no-cpu-baseline
