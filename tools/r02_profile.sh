#!/bin/bash
# GPU box, from the repo root: bash tools/r02_profile.sh <tag>
# The evidence of one build: N = 1 bench line, rocprofv3 kernel stats of BASELINE configs 1 (headline), 2, 3, HBM traffic
# counters of the headline's tile kernel (two separate --pmc passes), stage cycles, the other workloads, per-rank shard
# compute, the PCIe-inclusive Python-API probe.  Everything lands in gpurun_out/<tag>_*; copy what is kept to profiles/.
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
python3 bench.py > $O/${tag}_bench_paint.json 2> $O/${tag}_bench_paint.err && tail -c 600 $O/${tag}_bench_paint.json && echo
python3 bench.py --halos 100000 --no-cpu-baseline > $O/${tag}_bench_paint1e5.json 2>/dev/null
python3 bench.py --workload baryonify --halos 100000 > $O/${tag}_bench_bary1e5.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
stats() {  # name, bench args...
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_${name}_prof -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e "$@" > $O/${tag}_${name}_prof.json 2> $O/${tag}_${name}_prof.err
  f=$(find $O/${tag}_${name}_prof -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/${tag}_${name}_kernel_stats.csv
  echo "== $name"; head -8 $O/${tag}_${name}_kernel_stats.csv | cut -c1-150
}
stats paint
stats paint1e5 --halos 100000
stats bary1e5 --workload baryonify --halos 100000
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${tag}_pmc_$ctr -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $O/${tag}_pmc_$ctr > $O/${tag}_pmc_$ctr.txt 2>&1
  grep -A2 "shell_tile_kernel\|halo_row4\|FillFunc" $O/${tag}_pmc_$ctr.txt | head -12
done
cd $R
# stage cycles (a -DBFG_STAGE_TIMING=1 build, made here)
( cd baryonforge_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -ffp-contract=on -DBFG_STAGE_TIMING=2 -o /tmp/bfg_st.so csrc/bfg_mi355.hip 2>/dev/null )
for n in 1000000 100000; do BFG_SO=/tmp/bfg_st.so python3 tools/stage_timing.py $n 1024 paint; done > $O/${tag}_stage_cycles.txt 2>&1
BFG_SO=/tmp/bfg_st.so python3 tools/stage_timing.py 100000 1024 bary >> $O/${tag}_stage_cycles.txt 2>&1
grep -v "^/opt" $O/${tag}_stage_cycles.txt
python3 tools/pmc_to_json.py $O/${tag}_pmc_FETCH_SIZE.txt $O/${tag}_pmc_WRITE_SIZE.txt ${tag} $O/${tag}_pmc_traffic.json > /dev/null
bash tools/workloads.sh > $O/${tag}_other_workloads.txt 2>&1; cat $O/${tag}_other_workloads.txt
python3 tools/shard_scale.py > $O/${tag}_shard_scale.txt 2>&1; grep -v "^/opt" $O/${tag}_shard_scale.txt
python3 tools/e2e_probe.py > $O/${tag}_e2e_probe.txt 2>&1; grep -v "^/opt" $O/${tag}_e2e_probe.txt
# 2-rank rehearsals of the multi-GPU bench on this one GPU (gloo, both ranks on cuda:0)
export BFG_BENCH_BACKEND=gloo BFG_BENCH_ONE_DEVICE=1
for sc in weak strong; do
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 --scaling $sc > $O/${tag}_rehearsal_n2_${sc}.json 2> $O/${tag}_rehearsal_n2_${sc}.err && tail -c 300 $O/${tag}_rehearsal_n2_${sc}.json && echo
done
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --workload baryonify --halos 100000 > $O/${tag}_rehearsal_n2_bary.json 2> $O/${tag}_rehearsal_n2_bary.err && tail -c 300 $O/${tag}_rehearsal_n2_bary.json && echo
