#!/bin/bash
# GPU box (one GPU): rehearsal of bench.py --gpus 2 with both ranks on cuda:0 over gloo (the driver's runs use one GPU per
# rank and nccl = RCCL); checks the control flow, the JSON fields and the failure modes, not the xGMI numbers.
export BFG_BENCH_BACKEND=gloo BFG_BENCH_ONE_DEVICE=1
run() { python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 "${@:2}"; }
run 29611 --steps 6 --warmup 2 --halos 200000 --scaling strong --exchange auto > gpurun_out/r03_rehearsal_n2_strong.json 2> gpurun_out/r03_rehearsal_n2_strong.err
echo "strong rc=$?"
run 29612 --steps 6 --warmup 2 --halos 200000 --scaling weak --slices 8 --exchange auto > gpurun_out/r03_rehearsal_n2_weak.json 2> gpurun_out/r03_rehearsal_n2_weak.err
echo "weak rc=$?"
run 29615 --steps 6 --warmup 2 --halos 200000 --scaling strong > gpurun_out/r03_rehearsal_n2_strong_allreduce.json 2> gpurun_out/r03_rehearsal_n2_strong_allreduce.err
echo "strong allreduce rc=$?"
run 29613 --steps 4 --warmup 1 --halos 100000 --workload baryonify > gpurun_out/r03_rehearsal_n2_bary.json 2> gpurun_out/r03_rehearsal_n2_bary.err
echo "bary rc=$?"
# failure modes: must exit non-zero with one line, quickly
unset BFG_BENCH_ONE_DEVICE
( time python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29614 bench.py --gpus 2 --steps 2 ) > gpurun_out/r03_fail_devices.txt 2>&1
echo "too few GPUs rc=$?"; grep "bench.py: FAILED" gpurun_out/r03_fail_devices.txt | head -2
( time python3 bench.py --gpus 2 --steps 2 ) > gpurun_out/r03_fail_world.txt 2>&1
echo "gpus/world mismatch rc=$?"; grep "bench.py: FAILED" gpurun_out/r03_fail_world.txt | head -2
