#!/bin/bash
# round 4, first GPU check: sliced-call fixes, the 2-rank test with an empty shard, the self-launching bench
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -x -k "sliced" 2>&1 | tail -5 &&
python -m pytest tests/test_gpu_distributed.py -q -x 2>&1 | tail -15 &&
python bench.py --steps 20 --warmup 3 > gpurun_out/r04_bench_n1.json 2> gpurun_out/r04_bench_n1.err &&
{ python bench.py --gpus 2 > gpurun_out/r04_fail_gpus2.txt 2>&1; echo "exit code $?" >> gpurun_out/r04_fail_gpus2.txt; } &&
BFG_BENCH_ONE_DEVICE=1 BFG_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 > gpurun_out/r04_rehearsal_n2.json 2> gpurun_out/r04_rehearsal_n2.err
echo "rc=$?"
tail -n 3 gpurun_out/r04_bench_n1.err; tail -n 3 gpurun_out/r04_rehearsal_n2.err
cat gpurun_out/r04_fail_gpus2.txt | tail -5
