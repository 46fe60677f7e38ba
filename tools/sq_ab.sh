#!/bin/bash
# GPU box: SQ counters of the tile / prep kernels for several builds of the library on one bench workload
# usage: bash tools/sq_ab.sh <tag> "<bench args>" so1 so2 ...      (separate --pmc passes, --kernel-trace only)
tag=$1; args=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
out=$O/${tag}_sq_ab.txt; : > $out
for so in "$@"; do
  export BFG_SO=$R/$so
  echo "== $so" >> $out
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${tag}_sqd_$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --legs none $args > $O/${tag}_sqd_$i.log 2>&1
    python3 $R/tools/pmc_summary.py $O/${tag}_sqd_$i 2>&1 | grep -A5 "shell_tile_kernel" | grep -v "^--" >> $out
    rm -rf $O/${tag}_sqd_$i
  done
done
cut -c1-110 $out
