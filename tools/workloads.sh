p() { python -c "import sys,json; r=json.load(sys.stdin); q=r['roofline']; print('$1', 'halos/s %.3e' % r['value'], 'ms/step %.3f' % r['ms_per_step'], 'kernel_ms %.3f' % q['kernel_ms'], 'frac %.3f' % q['frac'], 'prep %.3f bin %.3f left %.3f regrid %s' % (q['prep_kernel_ms'], q['tile_binning_ms'] or 0, q['leftover_scatter_kernel_ms'] or 0, q['regrid_kernel_ms']))"; }
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --legs none 2>/dev/null | p paint
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --legs none --workload baryonify --halos 100000 2>/dev/null | p bary1e5
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --legs none --workload baryonify --halos 1000000 2>/dev/null | p bary1e6
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --legs none --steep 2>/dev/null | p steep
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --legs none --eps 20 2>/dev/null | p eps20
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --legs none --nside 2048 --halos 1250000 2>/dev/null | p nside2048
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --legs none --table stress 2>/dev/null | p stress
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --legs none --halos 100000 2>/dev/null | p paint1e5
