#!/bin/bash
# GPU box: SQ counters of the headline's kernels (separate --pmc passes), summaries in gpurun_out/<tag>_sq_<n>.txt
tag=${1:-sq}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/${tag}_counters_avail.txt 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${tag}_sq_$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e ${BENCH_ARGS} > $O/${tag}_sq_$i.log 2>&1
  python3 $R/tools/pmc_summary.py $O/${tag}_sq_$i > $O/${tag}_sq_$i.txt 2>&1
  grep -A5 "shell_tile_kernel" $O/${tag}_sq_$i.txt | head -8
done
