#!/bin/bash
# time several builds of libbfg_mi355.so on the same box: tools/abn.sh reps a.so b.so ...
reps=$1; shift
for r in $(seq $reps); do
  for so in "$@"; do
    BFG_SO=$PWD/$so python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.load(sys.stdin)['roofline']; print('$so', 'kernel_ms %.3f binning %.3f frac %.4f' % (r['kernel_ms'], r['tile_binning_ms'] or 0, r['frac']))"
  done
done
