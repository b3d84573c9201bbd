#!/bin/bash
mkdir -p gpurun_out
SDDP_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/bench_2rank_gloo.log 2>&1; echo "rc=$?"
tail -5 gpurun_out/bench_2rank_gloo.log | cut -c1-400
