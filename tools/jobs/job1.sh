#!/bin/bash
# GPU job 1 (round 2): GPU tests, default bench, forced-dist bench, FETCH/WRITE calibration
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2_job1
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest exit $?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"
VITS_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-passes > $O/bench_forcedist.json 2> $O/bench_forcedist.err; echo "forcedist exit $?"
tools/bin/fetch_calib > $O/fetch_calib.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$O/calib_fetch --output-format csv -- $GRAFT_REPO_ROOT/tools/bin/fetch_calib > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/$O/calib_write --output-format csv -- $GRAFT_REPO_ROOT/tools/bin/fetch_calib > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/fetch_calib_reduce.py $O/fetch_calibration.json $O/calib_fetch $O/calib_write
cat $O/fetch_calib.txt
head -c 1500 $O/bench.json
