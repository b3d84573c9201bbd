#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/t6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t6.log
grep -E "passed|failed|rc=|^FAILED|^E  " gpurun_out/t6.log | tail -12
for occ in 2 4; do
  SDDP_LIB=$PWD/build/libsddp_occ$occ.so timeout -k 10 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_occ$occ.log 2>&1
  echo "occ $occ: $(grep -o '"value": [0-9.]*' gpurun_out/bench_occ$occ.log | head -1) $(grep -o '"kernel_ms": [0-9.]*' gpurun_out/bench_occ$occ.log)"
done
SDDP_LIB=$PWD/build/libsddp_stamps.so timeout -k 10 200 python prof_stamps.py 1024 > gpurun_out/stamps_1024.log 2>&1; tail -12 gpurun_out/stamps_1024.log
