#!/bin/bash
# GPU box, from the repo root: bash tools/r02_check.sh <tag>  -- gpu tests, N=1 bench, 2-rank gloo rehearsals (weak / strong / baryonify)
tag=${1:-r02}
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${tag}_pytest.log 2>&1
rc=$?
tail -5 $O/${tag}_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > $O/${tag}_bench_n1.json 2> $O/${tag}_bench_n1.err || { tail -20 $O/${tag}_bench_n1.err; exit 1; }
cat $O/${tag}_bench_n1.json
export BFG_BENCH_BACKEND=gloo BFG_BENCH_ONE_DEVICE=1
for sc in weak strong; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 --scaling $sc > $O/${tag}_bench_n2_${sc}_gloo.json 2> $O/${tag}_bench_n2_${sc}.err || { tail -20 $O/${tag}_bench_n2_${sc}.err; exit 1; }
  tail -1 $O/${tag}_bench_n2_${sc}_gloo.json
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --workload baryonify --halos 100000 > $O/${tag}_bench_n2_bary_gloo.json 2> $O/${tag}_bench_n2_bary.err || { tail -20 $O/${tag}_bench_n2_bary.err; exit 1; }
tail -1 $O/${tag}_bench_n2_bary_gloo.json
