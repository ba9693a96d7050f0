#!/bin/bash
# the driver's own command; stdout kept whole, the LAST line is what the driver parses
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5_benchline
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_benchline/stdout.txt 2> gpurun_out/r5_benchline/stderr.txt
echo rc=$?
tail -c 5000 gpurun_out/r5_benchline/stdout.txt
python3 - <<'P'
import json
ls=open('gpurun_out/r5_benchline/stdout.txt').read().strip().splitlines()
print("lines", len(ls), "last bytes", len(ls[-1])); json.loads(ls[-1])
P
