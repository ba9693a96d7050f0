#!/bin/bash
# Final collection of a build: full GPU test suite, rocprofv3 stats + PMC passes (fp32 headline and f16), every bench line.
# usage: bash tools/jobs/final.sh vN
V=${1:-v2}
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2_$V; mkdir -p $O profiles
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest.log
bash tools/jobs/profile.sh r2_${V}_prof "c3|b64|f32" > $O/prof.log 2>&1
bash tools/jobs/profile.sh r2_${V}_prof_f16 "c3|b64|f16" --arith f16 > $O/prof_f16.log 2>&1
for t in "" "_f16"; do
  cp gpurun_out/r2_${V}_prof$t/pmc_traffic.json profiles/round2_${V}${t}_pmc_traffic.json
  cp gpurun_out/r2_${V}_prof$t/pmc_mfma.json profiles/round2_${V}${t}_pmc_mfma.json
  cp gpurun_out/r2_${V}_prof$t/kernel_stats.csv profiles/round2_${V}${t}_kernel_stats.csv
  cp gpurun_out/r2_${V}_prof$t/bench_under_rocprof.json profiles/round2_${V}${t}_bench_under_rocprof.json
done
bash tools/jobs/benchlines.sh $V
cp -r profiles $O/profiles_out
