#!/bin/bash
# Per-kernel totals of ONE batch-1 step under rocprofv3 --kernel-trace, for the default build and with an environment knob set.
# usage: b1agg.sh ARITH [KNOB]
A=${1:-f16}; K=$2
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in default $K; do
O=$GRAFT_REPO_ROOT/gpurun_out/b1agg_${A}_$v; mkdir -p $O
[ "$v" != default ] && export $K=1
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/tr --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --batch 1 --arith $A --no-prof --no-cpu-baseline --no-extra-passes --steps 6 --warmup 3 > $O/bench.json 2> $O/err.txt
cd $GRAFT_REPO_ROOT
python3 - <<EOF
import csv,glob,collections
f=glob.glob('$O/tr/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'embed_kernel' in r['Kernel_Name']]
step=rows[idx[-1]:]
t0=int(step[0]['Start_Timestamp']); end=max(int(r['End_Timestamp']) for r in step)
agg=collections.OrderedDict()
for r in step:
    name=r['Kernel_Name'].replace('void vits::','').replace('(vits::ConvParams)','').replace('(vits::Conv16Params)','').replace('(vits::RbPairParams)','')[:56]
    a=agg.setdefault(name,[0,0]); a[0]+=1; a[1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
print('== $A $v: step wall us', (end-t0)/1e3, 'kernels', len(step), 'sum us', sum(a[1] for a in agg.values())/1e3)
for k,(n,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:24]:
    print(f"{t/1e3:8.1f} us n={n:4d} avg {t/1e3/n:6.1f}  {k}")
EOF
rm -rf $O/tr
done
