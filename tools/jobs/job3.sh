#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2_job3; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_arith16.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -40 $O/pytest.log
for a in f16 bf16; do
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --arith $a > $O/bench_$a.json 2> $O/bench_$a.err; echo "bench $a exit $?"; tail -3 $O/bench_$a.err
python3 -c "
import json; d=json.load(open('$O/bench_$a.json')); print('$a', d['value'], d['ms_per_step'], d.get('value_without_kernel_events')); 
for k in d['top_kernels']: print(k)"
done
