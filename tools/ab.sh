#!/bin/bash
# A/B timing of two builds of libbfg_mi355.so on the same box: tools/ab.sh base.so [reps]
base=$1; reps=${2:-3}
for r in $(seq $reps); do
  for so in "$base" ""; do
    if [ -n "$so" ]; then export BFG_SO=$PWD/$so; else unset BFG_SO; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.load(sys.stdin)['roofline']; print('${so:-new}', 'kernel_ms %.3f binning %.3f prep %.3f frac %.4f' % (r['kernel_ms'], r['tile_binning_ms'] or 0, r['prep_kernel_ms'], r['frac']))"
  done
done
