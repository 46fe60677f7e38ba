#!/bin/bash
# GPU box: the snapshot step (BASELINE configs[4]) of several builds (build/<name>.so); usage: bash tools/snap_ab.sh name1 name2 ...
cd ${GRAFT_REPO_ROOT:-.}
for round in 1 2; do for b in "$@"; do
echo "== $b (round $round)"
BFG_SO=$PWD/build/$b.so python3 bench.py --workload snapshot --halos ${SNAP_HALOS:-100000} --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('step ms %.3f  particle kernel %.3f ms  frac %.3f  deposit %.3f ms  pairs %d' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['deposit_roofline']['kernel_ms'], r['halo_particle_pairs_per_launch']))"
done; done
