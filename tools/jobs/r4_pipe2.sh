#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for a in f16 bf16 f32; do
  python tools/pipe_bench.py --arith $a --steps $([ $a = f32 ] && echo 8 || echo 30) 2>&1 | grep batch
done
VITS_FRONT_PRIO=0 python tools/pipe_bench.py --arith f16 2>&1 | grep batch
VITS_NO_PIPELINE=1 python tools/pipe_bench.py --arith f16 2>&1 | grep batch
python tools/pipe_bench.py --arith f16 --batch 8 --ids 1024 --steps 10 2>&1 | grep batch
python tools/pipe_bench.py --arith bf16 --batch 8 --ids 1024 --steps 10 2>&1 | grep batch
python tools/pipe_bench.py --arith f16 --batch 1 --ids 128 --steps 50 2>&1 | grep batch
