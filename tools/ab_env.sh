#!/bin/bash
# A/B of one environment switch on the same box: tools/ab_env.sh VAR=VALUE [reps] [extra bench args]
kv=$1; reps=${2:-3}; shift; shift
for r in $(seq $reps); do
  for mode in base "$kv"; do
    if [ "$mode" = base ]; then pre=""; else pre="$kv"; fi
    env $pre python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; j=json.load(sys.stdin); r=j['roofline']; print('$mode', 'step_ms %.3f kernel_ms %.3f binning %.3f prep %.3f frac %.4f' % (j['ms_per_step'], r['kernel_ms'], r['tile_binning_ms'] or 0, r['prep_kernel_ms'], r['frac']))"
  done
done
