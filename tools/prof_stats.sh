#!/bin/bash
# GPU box, from the repo root: bash tools/prof_stats.sh <tag> [bench args...] -> gpurun_out/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats)
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e "$@" > $R/gpurun_out/${tag}_prof.json 2> $R/gpurun_out/${tag}_prof.err
f=$(find $R/gpurun_out/${tag}_prof -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.2f} total_ms {float(r['TotalDurationNs'])/1e6:8.3f} pct {r['Percentage']}")
PY
