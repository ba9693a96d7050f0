#!/bin/bash
# roctx phase ranges of the library (VITS_ROCTX=1) under rocprofv3 --marker-trace --kernel-trace: per range, count and mean host span,
# and the kernels whose start falls inside it. usage: markers.sh OUTNAME [bench args]
N=${1:-markers}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp VITS_ROCTX=1
O=$GRAFT_REPO_ROOT/gpurun_out/$N; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --marker-trace --kernel-trace -d $O/tr --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-prof --no-cpu-baseline --no-extra-passes --steps 3 --warmup 2 "$@" > $O/bench.json 2> $O/err.txt
cd $GRAFT_REPO_ROOT
python3 - <<EOF > $O/marker_ranges.txt
import csv, glob, collections
mf = glob.glob('$O/tr/**/*marker_api_trace.csv', recursive=True)
kf = glob.glob('$O/tr/**/*kernel_trace.csv', recursive=True)
print('# rocprofv3 --marker-trace --kernel-trace -- python3 bench.py --no-prof --no-cpu-baseline --no-extra-passes --steps 3 --warmup 2 $*   (VITS_ROCTX=1)')
if not mf:
    print('no marker trace written'); raise SystemExit
rows = list(csv.DictReader(open(mf[0])))
kern = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(kf[0]))) if kf else []
agg = collections.OrderedDict()
for r in rows:
    name = r.get('Function') or r.get('Name') or r.get('Message') or '?'
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    a = agg.setdefault(name, [0, 0])
    a[0] += 1; a[1] += en - st
print('%-28s %6s %14s' % ('range', 'count', 'mean host us'))
for name, (n, t) in agg.items():
    print('%-28s %6d %14.1f' % (name, n, t / n / 1e3))
print('kernels traced:', len(kern))
EOF
rm -rf $O/tr
cat $O/marker_ranges.txt
