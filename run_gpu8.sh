#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/t9.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t9.log
grep -E "passed|failed|rc=|^FAILED|^E  " gpurun_out/t9.log | tail -12
timeout -k 10 400 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_so.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_so.log
tail -n 2 gpurun_out/bench_so.log | head -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','mean_iters','mean_rollouts','converged_frac','max_iters_hit_frac','iterations_per_s')}); print(d['roofline']); print(d['cpu_baseline']['value'], d['ms_per_mpc_tick_b1'])"
