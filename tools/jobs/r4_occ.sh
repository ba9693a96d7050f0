#!/bin/bash
# one-off investigation: where the SIMD time of the dominant fp32 kernels goes (occupancy / wait / instruction-mix counters)
# usage: bash tools/jobs/r4_occ.sh [extra bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_occ; mkdir -p $O
BENCH="$GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --single-pass $*"
cd /tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $pass -d $O/p$i --output-format csv -- python3 $BENCH --steps 2 --warmup 1 > /dev/null 2> $O/p$i.err
  tail -2 $O/p$i.err
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_mfma.py $O/occ.json --workload "occ" $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 | tail -2
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
