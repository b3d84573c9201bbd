#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/t10.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t10.log
grep -E "passed|failed|rc=|^FAILED|^E  " gpurun_out/t10.log | tail -8
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout -k 10 400 python bench.py > gpurun_out/bench_default.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench_default.log | cut -c1-120
