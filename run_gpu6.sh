#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/t7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t7.log
grep -E "passed|failed|rc=|^FAILED|^E  " gpurun_out/t7.log | tail -12
for B in 1024 2048 4096; do
  timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --batch $B > gpurun_out/bench_B$B.log 2>&1
  echo "B $B: $(grep -o '"value": [0-9.]*' gpurun_out/bench_B$B.log | head -1) $(grep -o '"kernel_ms": [0-9.]*' gpurun_out/bench_B$B.log) $(grep -o '"iterations_per_s": [0-9.]*' gpurun_out/bench_B$B.log)"
done
SDDP_LIB=$PWD/build/libsddp_stamps.so timeout -k 10 200 python prof_stamps.py 1024 > gpurun_out/stamps_1024.log 2>&1; tail -12 gpurun_out/stamps_1024.log
