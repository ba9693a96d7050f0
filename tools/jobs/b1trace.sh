#!/bin/bash
# Kernel timeline of batch-1 steps (rocprofv3 --kernel-trace): per-kernel start / duration / queue, for latency analysis.
A=${1:-f32}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/b1trace_$A; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/tr --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --batch 1 --arith $A --no-prof --no-cpu-baseline --no-extra-passes --steps 6 --warmup 3 > $O/bench.json 2> $O/err.txt
cd $GRAFT_REPO_ROOT
python3 - <<EOF
import csv,glob
f=glob.glob('$O/tr/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step = kernels after the last 'embed_kernel'
idx=[i for i,r in enumerate(rows) if 'embed_kernel' in r['Kernel_Name']]
s=idx[-1]
step=rows[s:]
t0=int(step[0]['Start_Timestamp'])
end=max(int(r['End_Timestamp']) for r in step)
print('step wall us', (end-t0)/1e3, 'kernels', len(step), 'sum dur us', sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in step)/1e3)
out=open('$O/timeline.txt','w')
prev_end={}
for r in step:
    st=int(r['Start_Timestamp'])-t0; du=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    q=r.get('Queue_Id','?')
    name=r['Kernel_Name'].replace('void vits::','').replace('(vits::ConvParams)','').replace('(vits::Conv16Params)','')[:60]
    gap=st-prev_end.get(q,st)
    prev_end[q]=st+du
    out.write(f"{st/1e3:9.1f} us  dur {du/1e3:7.1f}  gap {gap/1e3:6.1f}  q{q}  grid {r.get('Grid_Size_X','?')}x{r.get('Grid_Size_Y','?')}x{r.get('Grid_Size_Z','?')} wg {r.get('Workgroup_Size_X','?')}  {name}\n")
out.close()
EOF
rm -rf $O/tr
head -150 $O/timeline.txt
