#!/bin/bash
# GPU box: the same quick_bench line for several builds of the library (one process each, two rounds)
# usage: bash tools/ab_so.sh "<workloads>" so1 so2 ...
wl=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
for round in 1 2; do
  for so in "$@"; do
    echo "== $so (round $round)"
    BFG_SO=$PWD/$so python3 tools/quick_bench.py --modes="${MODES:--}" --overwrite --workloads $wl --reps 1 --steps 20 2>&1 | grep -v "^/opt\|warn"
  done
done
