#!/bin/bash
# rocprofv3 kernel trace of the PIPELINED f16 loop (and of the serial loop): summed kernel time per batch against the wall time per batch
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_pipeprof; mkdir -p $O
cd /tmp
for mode in pipelined serial; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/$mode --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/pipe_bench.py --arith f16 --steps 20 --mode $mode > $O/$mode.txt 2> $O/$mode.err
  cp $(find $O/$mode -name "*kernel_stats.csv" | head -1) $O/${mode}_kernel_stats.csv
  rm -rf $O/$mode
  grep batch $O/$mode.txt
  python3 -c "
import csv
tot=0; n=0
for r in csv.DictReader(open('$O/${mode}_kernel_stats.csv')): tot+=float(r['TotalDurationNs'])
# batches in the trace: 3 warm-up serial + 2 reps x 20
print('$mode: summed kernel time', round(tot/1e6,2), 'ms over', 43, 'batches =', round(tot/1e6/43,3), 'ms per batch')"
done
