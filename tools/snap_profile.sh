#!/bin/bash
# usage (GPU box, repo root): bash tools/snap_profile.sh <tag>  -- BASELINE config 4 (512^3 particles, 1e5 halos) kernel stats
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_snap_stats -- python3 $R/tools/snapshot_scale.py > $R/gpurun_out/${tag}_snap.txt 2>&1
grep -v "^[WE]2026" $R/gpurun_out/${tag}_snap.txt | tail -6
grep "bfg::" $(find $R/gpurun_out/${tag}_snap_stats -name "*kernel_stats.csv") | cut -c1-160
