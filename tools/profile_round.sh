#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# kernel-trace stats + HBM traffic counters (separate passes) + bench lines -> gpurun_out/<tag>_*
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_paint_stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${tag}_paint_stats.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_bary_stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload baryonify --halos 100000 > $R/gpurun_out/${tag}_bary_stats.json 2> /dev/null
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_$ctr -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/${tag}_pmc_$ctr > $R/gpurun_out/${tag}_pmc_$ctr.txt 2>&1
done
cd $R
python3 bench.py > gpurun_out/${tag}_bench_paint.json 2> gpurun_out/${tag}_bench_paint.err
python3 bench.py --workload baryonify --halos 100000 > gpurun_out/${tag}_bench_bary.json 2> gpurun_out/${tag}_bench_bary.err
find gpurun_out/${tag}_paint_stats gpurun_out/${tag}_bary_stats -name "*kernel_stats.csv" | head
tail -2 gpurun_out/${tag}_bench_paint.json gpurun_out/${tag}_bench_bary.json
grep -A3 "shell_tile" gpurun_out/${tag}_pmc_FETCH_SIZE.txt gpurun_out/${tag}_pmc_WRITE_SIZE.txt
