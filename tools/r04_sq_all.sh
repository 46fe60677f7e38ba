#!/bin/bash
# GPU box: SQ counters of EVERY kernel of a bench workload (separate --pmc passes, --kernel-trace only): tools/r04_sq_all.sh <tag> [bench args]
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
out=$O/${tag}_sq_counters.txt
echo "# rocprofv3 --pmc (one pass per set) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --legs none $*" > $out
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SMEM" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${tag}_sqd_$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --legs none "$@" > $O/${tag}_sqd_$i.log 2>&1
  python3 $R/tools/pmc_summary.py $O/${tag}_sqd_$i 2>&1 | grep -A5 "halo_prep_kernel\|regrid_tile_kernel\|shell_tile_kernel\|shell_small" >> $out
  rm -rf $O/${tag}_sqd_$i
done
cat $out | cut -c1-110
