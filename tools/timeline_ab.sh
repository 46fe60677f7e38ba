#!/bin/bash
# GPU box: item timelines (-DBFG_STAGE_TIMING=4 builds made by tools/build_variant.sh) of several builds, same workload
# usage: bash tools/timeline_ab.sh "<halos> <nside> <paint|baryonify>" build1 build2 ...   (BFG_ST_STEEP=1 for the steep catalog)
cd "${GRAFT_REPO_ROOT:-.}"
wl=$1; shift
for b in "$@"; do
  echo "== $b"
  BFG_ST_MODE=4 BFG_SO=$PWD/build/$b.so python3 tools/stage_timing.py $wl 2>&1 | grep "thread\|    t"
done
