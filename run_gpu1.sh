#!/bin/bash
# first GPU session: parity tests, smoke, bench, rocprof kernel trace
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q > gpurun_out/t3.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t3.log
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/bench1.log 2>&1
echo "bench rc=$?" >> gpurun_out/bench1.log
grep -E "passed|failed|rc=" gpurun_out/t3.log | tail -5
tail -3 gpurun_out/smoke.log
tail -3 gpurun_out/bench1.log
