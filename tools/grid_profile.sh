#!/bin/bash
# usage (GPU box, repo root): bash tools/grid_profile.sh <tag>  -- periodic-grid runners at scale, kernel stats
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_grid_stats -- python3 $R/tools/grid_scale.py > $R/gpurun_out/${tag}_grid.txt 2>&1
grep -v "^[WE]2026" $R/gpurun_out/${tag}_grid.txt | tail -3
grep "bfg::" $(find $R/gpurun_out/${tag}_grid_stats -name "*kernel_stats.csv") | cut -c1-170
