#!/bin/bash
# GPU box: does the number of untimed warm-up steps change the timed region?  (clock ramp)  tools/warm_ab.sh
cd "${GRAFT_REPO_ROOT:-.}"
show='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print("warmup %4d  step_ms %.4f kernel_ms %.4f frac %.4f step_frac %.4f" % (j["warmup"], j["ms_per_step"], r["kernel_ms"], r["frac"], r["step_frac"]))'
for halos in 1000000 100000; do
  echo "== paint, $halos halos"
  for w in 5 50 500 2000 5; do
    python3 bench.py --steps 20 --warmup $w --halos $halos --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "$show"
  done
done
echo "== baryonify, 100000 halos"
for w in 5 500 2000; do
  python3 bench.py --steps 20 --warmup $w --halos 100000 --workload baryonify --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "$show"
done
