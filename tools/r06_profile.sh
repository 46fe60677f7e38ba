#!/bin/bash
# GPU box, from the repo root: bash tools/r06_profile.sh <part> ...   (parts: bench benchall stats pmc sq sqjson snaponly workloads api rehearsal)
# The evidence of one build.  Everything lands in gpurun_out/r06/r06_*; what is kept is copied to profiles/ afterwards.
tag=r06
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
BARGS="--no-cpu-baseline --no-e2e --legs none"
wl_args() {  # workload name -> bench args
  case $1 in
    paint) echo "" ;;
    paint1e5) echo "--halos 100000" ;;
    bary1e5) echo "--workload baryonify --halos 100000" ;;
    bary2048) echo "--workload baryonify --nside 2048 --halos 1250000" ;;
    steep) echo "--steep" ;;
    stress) echo "--table stress" ;;
    snapshot) echo "--workload snapshot --halos 100000" ;;
  esac
}
stats() {  # name, steps
  name=$1; steps=${2:-20}
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_${name}_prof -- python3 $R/bench.py --steps $steps --warmup 3 $BARGS $(wl_args $name) > $O/${tag}_${name}_prof.json 2> $O/${tag}_${name}_prof.err )
  f=$(find $O/${tag}_${name}_prof -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/${tag}_${name}_kernel_stats.csv
  echo "== $name"; head -6 $O/${tag}_${name}_kernel_stats.csv | cut -c1-160
  rm -rf $O/${tag}_${name}_prof
}
pmc() {  # name
  name=$1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && export TMPDIR=/tmp BFG_BENCH_RAMP_S=0 && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${tag}_pmc_${name}_$ctr -- python3 $R/bench.py --steps 3 --warmup 1 $BARGS $(wl_args $name) > /dev/null 2>&1 )
    python3 $R/tools/pmc_summary.py $O/${tag}_pmc_${name}_$ctr > $O/${tag}_pmc_${name}_$ctr.txt 2>&1
    rm -rf $O/${tag}_pmc_${name}_$ctr
    grep -A2 "shell_tile_kernel\|halo_prep\|snap_particle\|regrid_tile" $O/${tag}_pmc_${name}_$ctr.txt | head -8
  done
}
sq() {  # name: SQ counters of every kernel of the workload, four separate passes
  name=$1
  out=$O/${tag}_sq_counters_${name}.txt
  echo "# rocprofv3 --pmc (one pass per set, --kernel-trace only) -- python3 bench.py --steps 3 --warmup 1 $BARGS $(wl_args $name)" > $out
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" "SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM"; do
    i=$((i+1))
    ( cd /tmp && export TMPDIR=/tmp BFG_BENCH_RAMP_S=0 && rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${tag}_sqd_${name}_$i -- python3 $R/bench.py --steps 3 --warmup 1 $BARGS $(wl_args $name) > $O/${tag}_sqd.log 2>&1 )
    python3 $R/tools/pmc_summary.py $O/${tag}_sqd_${name}_$i 2>&1 | grep -A4 "halo_prep_kernel\|regrid_tile_kernel\|regrid_list_kernel\|shell_tile_kernel\|snap_particle_kernel\|dep_tile_kernel\|dep_key_kernel" | grep -v "^--" >> $out
    rm -rf $O/${tag}_sqd_${name}_$i
  done
  echo "== $name"; grep -c mean $out
}
for part in "$@"; do
case $part in
bench)
  python3 bench.py > $O/${tag}_bench_paint.json 2> $O/${tag}_bench_paint.err; tail -c 600 $O/${tag}_bench_paint.json; echo ;;
benchall)
  for w in paint1e5 bary1e5 stress steep; do python3 bench.py $BARGS $(wl_args $w) > $O/${tag}_bench_$w.json 2>/dev/null; done
  python3 bench.py $BARGS $(wl_args bary2048) --steps 10 > $O/${tag}_bench_bary2048_share.json 2>/dev/null
  python3 bench.py --workload baryonify --nside 2048 --halos 10000000 --steps 5 --warmup 2 $BARGS > $O/${tag}_bench_bary2048_whole.json 2>/dev/null
  python3 bench.py $BARGS $(wl_args snapshot) > $O/${tag}_bench_snapshot.json 2>/dev/null ;;
stats)
  stats paint; stats paint1e5; stats bary1e5; stats stress; stats steep; stats bary2048 10; stats snapshot 50 ;;
pmc)
  pmc paint; pmc paint1e5; pmc bary1e5; pmc steep; pmc bary2048; pmc snapshot
  python3 tools/pmc_to_json.py $tag $O/${tag}_pmc_traffic.json \
    paint_auto_n1000000_nside1024=$O/${tag}_pmc_paint_FETCH_SIZE.txt,$O/${tag}_pmc_paint_WRITE_SIZE.txt \
    paint_auto_n100000_nside1024=$O/${tag}_pmc_paint1e5_FETCH_SIZE.txt,$O/${tag}_pmc_paint1e5_WRITE_SIZE.txt \
    baryonify_auto_n100000_nside1024=$O/${tag}_pmc_bary1e5_FETCH_SIZE.txt,$O/${tag}_pmc_bary1e5_WRITE_SIZE.txt \
    baryonify_auto_n1250000_nside2048=$O/${tag}_pmc_bary2048_FETCH_SIZE.txt,$O/${tag}_pmc_bary2048_WRITE_SIZE.txt \
    paint_auto_n1000000_nside1024_default_steep_eps10=$O/${tag}_pmc_steep_FETCH_SIZE.txt,$O/${tag}_pmc_steep_WRITE_SIZE.txt \
    snapshot_n100000_part512=$O/${tag}_pmc_snapshot_FETCH_SIZE.txt,$O/${tag}_pmc_snapshot_WRITE_SIZE.txt,snap_particle_kernel \
    snapshot_n100000_part512_deposit=$O/${tag}_pmc_snapshot_FETCH_SIZE.txt,$O/${tag}_pmc_snapshot_WRITE_SIZE.txt,dep_key_kernel+dep_tile_kernel \
    _prep_paint_n1000000=$O/${tag}_pmc_paint_FETCH_SIZE.txt,$O/${tag}_pmc_paint_WRITE_SIZE.txt,halo_prep_kernel \
    _regrid_bary_n100000=$O/${tag}_pmc_bary1e5_FETCH_SIZE.txt,$O/${tag}_pmc_bary1e5_WRITE_SIZE.txt,regrid_tile_kernel ;;
sq)
  sq paint; sq paint1e5; sq bary1e5; sq bary2048; sq steep; sq snapshot ;;
sqjson)
  python3 tools/sq_to_json.py $tag $O/${tag}_sq_counters.json \
    "paint_auto_n1000000_nside1024=$O/${tag}_sq_counters_paint.txt,shell_tile_kernel<0" \
    "paint_auto_n100000_nside1024=$O/${tag}_sq_counters_paint1e5.txt,shell_tile_kernel<0" \
    "baryonify_auto_n100000_nside1024=$O/${tag}_sq_counters_bary1e5.txt,shell_tile_kernel<1" \
    "baryonify_auto_n1250000_nside2048=$O/${tag}_sq_counters_bary2048.txt,shell_tile_kernel<1" \
    "paint_auto_n1000000_nside1024_default_steep_eps10=$O/${tag}_sq_counters_steep.txt,shell_tile_kernel<0" \
    "snapshot_n100000_part512=$O/${tag}_sq_counters_snapshot.txt,snap_particle_kernel" \
    "snapshot_n100000_part512_deposit=$O/${tag}_sq_counters_snapshot.txt,dep_tile_kernel" \
    "_regrid_n100000_nside1024=$O/${tag}_sq_counters_bary1e5.txt,regrid_tile_kernel" \
    "_regrid_n1250000_nside2048=$O/${tag}_sq_counters_bary2048.txt,regrid_tile_kernel" \
    "_prep_paint_n1000000=$O/${tag}_sq_counters_paint.txt,halo_prep_kernel" ;;
snaponly)   # the snapshot workload alone (after a change to its kernels): stats, traffic, SQ counters, bench line
  stats snapshot 50; pmc snapshot; sq snapshot
  python3 bench.py $BARGS $(wl_args snapshot) > $O/${tag}_bench_snapshot.json 2>/dev/null; tail -c 300 $O/${tag}_bench_snapshot.json; echo ;;
workloads)
  bash tools/workloads.sh > $O/${tag}_other_workloads.txt 2>&1; cat $O/${tag}_other_workloads.txt ;;
api)
  python3 tools/e2e_probe.py > $O/${tag}_e2e_probe.txt 2>&1; grep -v "^/opt" $O/${tag}_e2e_probe.txt
  python3 tools/bary_api_probe.py >> $O/${tag}_e2e_probe.txt 2>&1; tail -9 $O/${tag}_e2e_probe.txt ;;
rehearsal)
  BFG_BENCH_ONE_DEVICE=1 BFG_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 5 > $O/${tag}_rehearsal_n2_strong.json 2> $O/${tag}_rehearsal_n2_strong.err
  tail -c 300 $O/${tag}_rehearsal_n2_strong.json; echo ;;
esac
done
