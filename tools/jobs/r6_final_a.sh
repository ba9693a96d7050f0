#!/bin/bash
# Final collection of round 6, part A: smoke, the full GPU test suite, profile sets of the fp32 headline, of c3 f16 and of the split arithmetic.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_final; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
bash tools/jobs/profile.sh r6_prof_f32 "c3|b64|f32" > $O/prof_f32.log 2>&1; tail -3 $O/prof_f32.log
bash tools/jobs/profile.sh r6_prof_f16 "c3|b64|f16" --arith f16 > $O/prof_f16.log 2>&1; tail -3 $O/prof_f16.log
bash tools/jobs/profile.sh r6_prof_f32split "c3|b64|f32split" --arith f32split > $O/prof_f32split.log 2>&1; tail -3 $O/prof_f32split.log
ls gpurun_out/r6_prof_f32 gpurun_out/r6_prof_f16 gpurun_out/r6_prof_f32split
