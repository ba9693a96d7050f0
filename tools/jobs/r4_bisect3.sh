#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for kv in "VITS_X=0" "VITS_FRONT_PRIO=0" "VITS_FRONT_AT_LOAD=1" "VITS_FRONT_AT_LOAD=1 VITS_FRONT_PRIO=0" "GPU_MAX_HW_QUEUES=8" "VITS_RB_STREAMS=1"; do
  env DBG_ORDER=bench $kv python tools/sub_dbg.py fresh 2>&1 | grep c3_f16 | sed "s/^/bad order, $kv: /"
done
python tools/sub_dbg.py fresh 2>&1 | grep c3_f16 | sed "s/^/good order: /"
