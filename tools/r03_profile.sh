#!/bin/bash
# GPU box, from the repo root: bash tools/r03_profile.sh <part> ...   (parts: bench stats pmc workloads api rehearsal)
# The evidence of one build.  Everything lands in gpurun_out/r03_*; what is kept is copied to profiles/ afterwards.
tag=r03
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
stats() {  # name, bench args...
  name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_${name}_prof -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e "$@" > $O/${tag}_${name}_prof.json 2> $O/${tag}_${name}_prof.err )
  f=$(find $O/${tag}_${name}_prof -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/${tag}_${name}_kernel_stats.csv
  echo "== $name"; head -6 $O/${tag}_${name}_kernel_stats.csv | cut -c1-160
}
pmc() {  # name, bench args...
  name=$1; shift
  for ctr in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && export TMPDIR=/tmp BFG_BENCH_RAMP_S=0 && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${tag}_pmc_${name}_$ctr -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e "$@" > /dev/null 2>&1 )
    python3 $R/tools/pmc_summary.py $O/${tag}_pmc_${name}_$ctr > $O/${tag}_pmc_${name}_$ctr.txt 2>&1
    grep -A2 "shell_tile_kernel\|halo_prep" $O/${tag}_pmc_${name}_$ctr.txt | head -8
  done
}
for part in "$@"; do
case $part in
bench)
  python3 bench.py > $O/${tag}_bench_paint.json 2> $O/${tag}_bench_paint.err && tail -c 400 $O/${tag}_bench_paint.json && echo
  python3 bench.py --halos 100000 --no-cpu-baseline > $O/${tag}_bench_paint1e5.json 2>/dev/null
  python3 bench.py --workload baryonify --halos 100000 > $O/${tag}_bench_bary1e5.json 2>/dev/null
  python3 bench.py --table stress --no-cpu-baseline > $O/${tag}_bench_stress.json 2>/dev/null
  python3 bench.py --steep --no-cpu-baseline > $O/${tag}_bench_steep.json 2>/dev/null ;;
stats)
  stats paint
  stats paint1e5 --halos 100000
  stats bary1e5 --workload baryonify --halos 100000
  stats stress --table stress
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_bary2048_prof -- python3 $R/bench.py --workload baryonify --nside 2048 --halos 10000000 --steps 3 --warmup 1 --no-cpu-baseline > $O/${tag}_bary2048_prof.json 2> $O/${tag}_bary2048_prof.err )
  cp "$(find $O/${tag}_bary2048_prof -name '*kernel_stats.csv' | head -1)" $O/${tag}_bary2048_whole_kernel_stats.csv; head -6 $O/${tag}_bary2048_whole_kernel_stats.csv | cut -c1-160 ;;
pmc)
  pmc paint
  pmc paint1e5 --halos 100000
  pmc bary1e5 --workload baryonify --halos 100000
  python3 tools/pmc_to_json.py $tag $O/${tag}_pmc_traffic.json \
    paint_auto_n1000000_nside1024=$O/${tag}_pmc_paint_FETCH_SIZE.txt,$O/${tag}_pmc_paint_WRITE_SIZE.txt \
    paint_auto_n100000_nside1024=$O/${tag}_pmc_paint1e5_FETCH_SIZE.txt,$O/${tag}_pmc_paint1e5_WRITE_SIZE.txt \
    baryonify_auto_n100000_nside1024=$O/${tag}_pmc_bary1e5_FETCH_SIZE.txt,$O/${tag}_pmc_bary1e5_WRITE_SIZE.txt \
    _prep_paint_n1000000=$O/${tag}_pmc_paint_FETCH_SIZE.txt,$O/${tag}_pmc_paint_WRITE_SIZE.txt,halo_prep_kernel ;;
workloads)
  bash tools/workloads.sh > $O/${tag}_other_workloads.txt 2>&1; cat $O/${tag}_other_workloads.txt ;;
api)
  python3 tools/e2e_probe.py > $O/${tag}_e2e_probe.txt 2>&1; grep -v "^/opt" $O/${tag}_e2e_probe.txt
  python3 tools/d2h_probe.py >> $O/${tag}_e2e_probe.txt 2>&1; tail -9 $O/${tag}_e2e_probe.txt ;;
rehearsal)
  bash tools/rehearse.sh ;;
esac
done
