#!/bin/bash
# SQ counters of the tile kernel, one rocprofv3 --pmc pass per counter set (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcx$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > /dev/null 2>$R/gpurun_out/pmcx$i.err
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcx$i 2>&1 | grep -A8 "shell_tile_kernel"
done
