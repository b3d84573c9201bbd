#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/t5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t5.log
timeout -k 10 400 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_full.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_full.log
grep -E "passed|failed|rc=|^FAILED|^E  " gpurun_out/t5.log | tail -12
tail -n 2 gpurun_out/bench_full.log | cut -c1-300; grep -o '"pcie_inclusive[^,]*,' gpurun_out/bench_full.log; grep -o '"cpu_baseline.*' gpurun_out/bench_full.log
