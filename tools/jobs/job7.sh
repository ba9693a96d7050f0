#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2_job7; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $O/pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/profile_dump.py 64 128 f16 > $O/dump.txt 2> $O/err.txt
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py "$O/pmc/**/*counter_collection.csv" > $O/summary.csv 2>&1
rm -rf $O/pmc
grep -E "kernel,|conv16" $O/summary.csv | head -30
