#!/bin/bash
# GPU box: item timeline of the tile kernel (-DBFG_STAGE_TIMING=4 builds in build/, threads 0 / 64 / 448) for a workload
# usage: bash tools/r04_item_timeline.sh <halos> <nside> <paint|baryonify>
cd "${GRAFT_REPO_ROOT:-.}"
for t in 0 64 448; do
  BFG_ST_MODE=4 BFG_ST_TID=$t BFG_SO=$PWD/build/bfg_st4_$t.so python3 tools/stage_timing.py "$@" 2>&1 | grep "thread\|    t"
done
