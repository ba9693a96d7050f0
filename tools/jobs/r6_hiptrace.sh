#!/bin/bash
# host-side API timeline of batch-1 calls (rocprofv3 --hip-runtime-trace --kernel-trace): where the host spends the time between the frame-count read and the last enqueue
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/hiptrace; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --hip-runtime-trace -d $O/tr --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --batch 1 --arith f16 --no-prof --no-cpu-baseline --no-extra-passes --steps 6 --warmup 3 > $O/bench.json 2> $O/err.txt
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv,glob
f=glob.glob('$O/tr/**/*hip_api_trace.csv',recursive=True)
print(f)
rows=list(csv.DictReader(open(f[0])))
print(rows[0].keys())
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last call: from the 110th-last kernel launch to the last launch + 8 calls
li=[i for i,r in enumerate(rows) if 'LaunchKernel' in r['Function']]
lo=li[-112]; hi=min(len(rows), li[-1]+8)
tail=rows[lo:hi]
t0=int(tail[0]['Start_Timestamp'])
prev=None
out=open('$O/api_tail.txt','w')
for r in tail:
    st=int(r['Start_Timestamp']); en=int(r['End_Timestamp'])
    gap=(st-prev)/1e3 if prev else 0
    out.write(f"{(st-t0)/1e3:9.1f} us  dur {(en-st)/1e3:7.1f}  gap {gap:6.1f}  {r['Function']}\n")
    prev=en
out.close()
PY
rm -rf $O/tr
cat $O/api_tail.txt
