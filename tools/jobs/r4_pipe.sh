#!/bin/bash
# round 4, first GPU job: the pipeline / busy-guard tests, the ADVICE tests, and serial-vs-pipelined step times
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4_pipe; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_round2.py "tests/test_gpu_arith16.py::test_converter_path_keeps_its_arithmetic_under_the_profiler_and_the_grouped_schedule" -q -x > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -15 $O/pytest.log
for a in f16 bf16 f32; do
  python tools/pipe_bench.py --arith $a --steps $([ $a = f32 ] && echo 8 || echo 30) 2>&1 | tail -1
done
VITS_FRONT_PRIO=0 python tools/pipe_bench.py --arith f16 2>&1 | tail -1
VITS_NO_PIPELINE=1 python tools/pipe_bench.py --arith f16 2>&1 | tail -1
python tools/pipe_bench.py --arith f16 --batch 8 --ids 1024 --steps 10 2>&1 | tail -1
