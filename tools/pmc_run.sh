#!/bin/bash
# usage: pmc_run.sh <outname> -- sets of counters run as separate passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU" "SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcx$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>$R/gpurun_out/pmcx$i.err
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcx$i 2>&1 | grep -A8 "shell_tile_kernel"
done
