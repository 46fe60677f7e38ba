#!/bin/bash
# GPU box: where the steep-catalog tile kernel's time goes (item timeline) and the existing variants on it
cd "${GRAFT_REPO_ROOT:-.}"
for t in 0 64 448; do
  BFG_ST_STEEP=1 BFG_ST_MODE=4 BFG_ST_TID=$t BFG_SO=$PWD/build/bfg_st4_$t.so python3 tools/stage_timing.py 1000000 1024 paint 2>&1 | grep -v "^/opt\|warn"
done
python3 tools/quick_bench.py --modes="-;BFG_TILE_LIGHT=1;BFG_DEBUG=2;BFG_DEBUG=10;BFG_DEBUG=202" --overwrite --workloads steep --reps 1 --steps 20 2>&1 | grep -v "^/opt\|warn"
