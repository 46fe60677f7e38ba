#!/bin/bash
# GPU box, from the repo root: bash tools/r04_profile.sh <part> ...   (parts: bench stats pmc workloads api rehearsal)
# The evidence of one build.  Everything lands in gpurun_out/r03_*; what is kept is copied to profiles/ afterwards.
tag=r04
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
stats() {  # name, bench args...
  name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_${name}_prof -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --legs none "$@" > $O/${tag}_${name}_prof.json 2> $O/${tag}_${name}_prof.err )
  f=$(find $O/${tag}_${name}_prof -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/${tag}_${name}_kernel_stats.csv
  echo "== $name"; head -6 $O/${tag}_${name}_kernel_stats.csv | cut -c1-160
}
pmc() {  # name, bench args...
  name=$1; shift
  for ctr in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && export TMPDIR=/tmp BFG_BENCH_RAMP_S=0 && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${tag}_pmc_${name}_$ctr -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --legs none "$@" > /dev/null 2>&1 )
    python3 $R/tools/pmc_summary.py $O/${tag}_pmc_${name}_$ctr > $O/${tag}_pmc_${name}_$ctr.txt 2>&1
    grep -A2 "shell_tile_kernel\|halo_prep" $O/${tag}_pmc_${name}_$ctr.txt | head -8
  done
}
for part in "$@"; do
case $part in
bench)
  python3 bench.py > $O/${tag}_bench_paint.json 2> $O/${tag}_bench_paint.err && tail -c 400 $O/${tag}_bench_paint.json && echo
  python3 bench.py --halos 100000 --no-cpu-baseline --legs none > $O/${tag}_bench_paint1e5.json 2>/dev/null
  python3 bench.py --workload baryonify --halos 100000 --legs none > $O/${tag}_bench_bary1e5.json 2>/dev/null
  python3 bench.py --table stress --no-cpu-baseline --legs none > $O/${tag}_bench_stress.json 2>/dev/null
  python3 bench.py --steep --no-cpu-baseline --legs none > $O/${tag}_bench_steep.json 2>/dev/null
  python3 bench.py --workload baryonify --nside 2048 --halos 1250000 --steps 10 --legs none > $O/${tag}_bench_bary2048_share.json 2>/dev/null
  python3 bench.py --workload baryonify --nside 2048 --halos 10000000 --steps 5 --warmup 2 --legs none > $O/${tag}_bench_bary2048_whole.json 2>/dev/null ;;
stats)
  stats paint
  stats paint1e5 --halos 100000
  stats bary1e5 --workload baryonify --halos 100000
  stats stress --table stress
  stats steep --steep
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_bary2048_prof -- python3 $R/bench.py --workload baryonify --nside 2048 --halos 10000000 --steps 3 --warmup 1 --no-cpu-baseline --legs none > $O/${tag}_bary2048_prof.json 2> $O/${tag}_bary2048_prof.err )
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
  # --gpus 2 on a one-GPU box: (1) as plain python: the launcher's children refuse; (2) the same under gloo on one device: the line
  for sc in strong; do
    BFG_BENCH_ONE_DEVICE=1 BFG_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 5 > $O/${tag}_rehearsal_n2_$sc.json 2> $O/${tag}_rehearsal_n2_$sc.err
    tail -c 300 $O/${tag}_rehearsal_n2_$sc.json; echo
  done ;;
failures)
  F=$O/${tag}_bench_failure_modes.txt; : > $F
  echo "### python3 bench.py --gpus 2   (plain python on a one-GPU box: bench.py spawns its two ranks, they refuse, the launcher reports)" >> $F
  ( time python3 bench.py --gpus 2 --steps 2 ) >> $F 2>&1; echo "exit code of the launcher: $?" >> $F
  echo "### one rank wedged before the group forms (BFG_BENCH_TEST_STALL=1: rank 1 blocks in a C call), deadline 20 s, gloo on one device" >> $F
  ( time BFG_BENCH_TEST_STALL=1 BFG_BENCH_DEADLINE_S=20 BFG_BENCH_TIMEOUT_S=15 BFG_BENCH_ONE_DEVICE=1 BFG_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 2 ) 2>&1 | grep -v "^\[W\|amdgpu.ids\|Gloo" >> $F; echo "exit code of the launcher: ${PIPESTATUS[0]}" >> $F
  echo "### the extra legs overrun their deadline (BFG_BENCH_LEGS_DEADLINE_S=0.5): the main line is printed, exit 0" >> $F
  ( BFG_BENCH_LEGS_DEADLINE_S=0.5 python3 bench.py --steps 5 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print({k: r[k] for k in ('value','n_gpus','scaling','legs')})" ) >> $F 2>&1; echo "exit code: $?" >> $F
  echo "### under torch.distributed.run with too few GPUs (as in round 3)" >> $F
  ( python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 2>&1 | grep "bench.py: FAILED" ) >> $F; echo "exit code of torchrun: ${PIPESTATUS[0]}" >> $F
  cat $F ;;
esac
done
