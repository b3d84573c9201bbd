#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for B in 1024 4096; do
timeout -k 10 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --batch $B > gpurun_out/bench_p$B.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_p$B.log
tail -n 2 gpurun_out/bench_p$B.log | head -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','ms_per_step','mean_iters','converged_frac','pipelined_2_streams_solves_per_s')})"
done
