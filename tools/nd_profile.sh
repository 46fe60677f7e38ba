#!/bin/bash
# GPU box: the kernels of one N-dimensional-table call (tools/nd_probe.py, ND_PROBE_ONLY=paint:4 by default): rocprofv3 stats + SQ counters
# of nd_rows_blocked_kernel -> gpurun_out/r05_nd4_kernel_stats.csv, gpurun_out/r05_sq_counters_nd4.txt
R=${GRAFT_REPO_ROOT:-$PWD}
cfg=${1:-paint:4}
cd /tmp && export TMPDIR=/tmp ND_PROBE_ONLY=$cfg
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ndp -- python3 $R/tools/nd_probe.py > $R/gpurun_out/nd_profile_probe.txt 2>&1
cp $(ls $R/gpurun_out/ndp/*/*kernel_stats.csv | head -1) $R/gpurun_out/r05_nd4_kernel_stats.csv; rm -rf $R/gpurun_out/ndp
out=$R/gpurun_out/r05_sq_counters_nd4.txt
echo "# rocprofv3 --pmc (one pass per set, --kernel-trace only) -- ND_PROBE_ONLY=$cfg python3 tools/nd_probe.py" > $out
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/ndsq_$i -- python3 $R/tools/nd_probe.py > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/ndsq_$i 2>&1 | grep -A4 "nd_rows_blocked_kernel\|nd_cell_kernel" | grep -v "^--" >> $out
  rm -rf $R/gpurun_out/ndsq_$i
done
cd $R; grep "extra axes" gpurun_out/nd_profile_probe.txt | cut -c1-70; python3 tools/stats_top.py gpurun_out/r05_nd4_kernel_stats.csv 8; grep -c mean $out
