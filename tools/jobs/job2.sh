#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2_job2; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_round2.py tests/test_gpu_streaming.py tests/test_gpu_model.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
bash tools/jobs/profile.sh r2_v0_prof
mkdir -p profiles
cp gpurun_out/r2_v0_prof/pmc_traffic.json profiles/round2_v0_pmc_traffic.json; cp gpurun_out/r2_v0_prof/pmc_mfma.json profiles/round2_v0_pmc_mfma.json
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_with_pmc.json 2> $O/bench.err; echo "bench exit $?"
python3 -c "
import json; d=json.load(open('$O/bench_with_pmc.json')); print(json.dumps(d['roofline'], indent=1))"
