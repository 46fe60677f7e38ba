#!/bin/bash
# GPU box: instruction counters of the tile kernel per stage, by ablation (BFG_DEBUG: 0 all, 2 no pixel stage, 10 no pixel stage and no
# (pair, ring) slots, 74 also no blend) for several builds: tools/sq_ablate.sh <tag> "<bench args>" so1 so2 ...
tag=$1; args=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
out=$O/${tag}_sq_ablate.txt; : > $out
for so in "$@"; do
  export BFG_SO=$R/$so
  for dbg in 0 2 10 74; do
    export BFG_DEBUG=$dbg
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/${tag}_sqd -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --legs none $args > $O/${tag}_sqd.log 2>&1
    echo "$so BFG_DEBUG=$dbg $(python3 $R/tools/pmc_summary.py $O/${tag}_sqd 2>&1 | grep -A4 'shell_tile_kernel' | grep mean | awk '{printf "%s %s  ", $1, $3}')" >> $out
    rm -rf $O/${tag}_sqd
  done
done
cat $out
