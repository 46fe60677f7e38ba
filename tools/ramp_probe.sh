#!/bin/bash
# GPU box: the main line's kernel time against the length of the clock ramp in front of it (BFG_BENCH_RAMP_S), and the steady rate of
# tools/quick_bench.py on the same box
cd ${GRAFT_REPO_ROOT:-.}
for r in 0.25 1 3 8; do
  BFG_BENCH_RAMP_S=$r python3 bench.py --legs none --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('ramp $r s: step %.4f ms  tile kernel %.4f ms  prep %.4f  ramp_steps %d  algorithmic frac %.3f' % (d['ms_per_step'], r['kernel_ms'], r['prep_kernel_ms'], d['ramp_steps'], r['algorithmic_frac']))"
done
python3 tools/quick_bench.py --modes - --overwrite --workloads paint1e6 --reps 3 --steps 20 2>&1 | grep paint1e6 | cut -c1-120
