#!/bin/bash
# Profile job: rocprofv3 kernel stats + PMC passes (HBM traffic, MFMA busy, LDS conflicts) over the bench workload, reduced into
# profiles-ready files under gpurun_out/$TAG/. usage: bash tools/jobs/profile.sh TAG WORKLOAD_TAG [extra bench args]
# (WORKLOAD_TAG = workload|batch|arith of the bench command, e.g. "c3|b64|f32": recorded in the artefacts, matched by bench.py)
TAG=${1:-prof}; shift
WTAG=${1:-c3|b64|f32}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O
BENCH="$GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --single-pass $*"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $BENCH --steps 5 --warmup 2 > $O/bench_under_rocprof.json 2> $O/stats.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass -d $O/pmc_$name --output-format csv -- python3 $BENCH --steps 2 --warmup 1 > /dev/null 2> $O/pmc_$name.err
done
# the library-default schedule (no per-kernel events: three streams, separate launches) for bench.py's roofline_default_schedule block
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats_default --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-prof --no-sub-results --no-extra-passes $* --steps 5 --warmup 2 > $O/bench_default_under_rocprof.json 2> $O/stats_default.err
cd $GRAFT_REPO_ROOT
cp $(find $O/stats_default -name "*kernel_stats.csv" | head -1) $O/default_kernel_stats.csv 2>/dev/null
python3 tools/default_schedule.py $O/default_schedule.json $O/default_kernel_stats.csv --workload "$WTAG" --steps-in-trace 7
rm -rf $O/stats_default
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
SPS=$(python3 -c "import json; print(json.load(open('$O/bench_under_rocprof.json'))['config']['samples_per_step'])" 2>/dev/null || echo 0)
python3 tools/pmc_traffic.py $O/pmc_traffic.json --workload "$WTAG" --steps-in-trace 3 --samples-per-step $SPS --fetch-cal 2.0 --write-cal 1.0 "$O/pmc_FETCH_SIZE/**/*counter_collection.csv" "$O/pmc_WRITE_SIZE/**/*counter_collection.csv"
python3 tools/pmc_mfma.py $O/pmc_mfma.json --workload "$WTAG" $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/pmc_SQ_LDS_BANK_CONFLICT
# keep the merged payload small: drop the raw traces
rm -rf $O/stats $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/pmc_SQ_LDS_BANK_CONFLICT
head -12 $O/kernel_stats.csv | cut -c1-150
