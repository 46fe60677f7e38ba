#!/bin/bash
# GPU box: the N = 1 anchors of BASELINE config 4 (8 x MI355X BaryonifyShell, 1e7 halos, NSIDE 2048): one rank's share and the
# whole catalog on one GPU, bench lines + rocprofv3 kernel stats (written under gpurun_out/, copied to profiles/ by hand)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
python3 $R/bench.py --workload baryonify --nside 2048 --halos 1250000 --steps 5 --warmup 2 --no-cpu-baseline > $O/r03_bench_bary2048_share.json 2> $O/r03_bench_bary2048_share.err
echo "share rc=$?"
python3 $R/bench.py --workload baryonify --nside 2048 --halos 10000000 --steps 3 --warmup 1 --no-cpu-baseline > $O/r03_bench_bary2048_whole.json 2> $O/r03_bench_bary2048_whole.err
echo "whole rc=$?"
python3 $R/bench.py --workload paint --nside 2048 --halos 10000000 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $O/r03_bench_paint2048_whole.json 2> $O/r03_bench_paint2048_whole.err
echo "paint whole rc=$?"
rm -rf $O/prof_c4 && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o c4 -- python3 $R/bench.py --workload baryonify --nside 2048 --halos 10000000 --steps 3 --warmup 1 --no-cpu-baseline > $O/r03_c4_prof.log 2>&1
echo "rocprof rc=$?"
find $O/prof_c4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r03_bary2048_whole_kernel_stats.csv
head -8 $O/r03_bary2048_whole_kernel_stats.csv | cut -c1-200
