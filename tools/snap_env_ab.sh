#!/bin/bash
# GPU box: the snapshot step (BASELINE configs[4]) under several environment settings of ONE build, two rounds each, then the same with
# the particles in random order (tools/snapshot_scale.py, SNAP_SHUFFLE=1).  usage: bash tools/snap_env_ab.sh "VAR=a" "VAR=b" ...
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r06
for round in 1 2; do for e in "$@"; do
echo "== $e (round $round)"
env $e python3 bench.py --workload snapshot --halos ${SNAP_HALOS:-100000} --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('step ms %.3f  particle kernel %.3f ms  strict frac %.3f  deposit %.3f ms  pairs %d' % (d['ms_per_step'], r['kernel_ms'], r['algorithmic_frac'], d['deposit_roofline']['kernel_ms'], r['halo_particle_pairs_per_launch']))"
done; done
for e in "$@"; do
echo "== $e, particles in random order"
env $e SNAP_SHUFFLE=1 python3 tools/snapshot_scale.py 2>/dev/null | tail -4
done
