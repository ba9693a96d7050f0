#!/bin/bash
# rocprofv3 kernel stats of the batch-1 (config 2) loop in fp32 and f16: which kernels a single utterance runs on (conv_lat16_kernel, the narrow
# flow / rbpair16 blocks ...). usage: bash tools/jobs/r4_b1stats.sh TAG
TAG=${1:-v13}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_b1stats; mkdir -p $O
cd /tmp
for a in f32 f16; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/$a --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --batch 1 --arith $a --no-prof --no-cpu-baseline --no-extra-passes --steps 40 --warmup 5 > $O/bench_$a.json 2> $O/err_$a.txt
  cp $(find $O/$a -name "*kernel_stats.csv" | head -1) $O/round4_${TAG}_b1_${a}_kernel_stats.csv
  rm -rf $O/$a
  head -8 $O/round4_${TAG}_b1_${a}_kernel_stats.csv | cut -c1-160
done
