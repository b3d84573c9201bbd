#!/bin/bash
# bench + rocprofv3 kernel trace + PMC (HBM traffic) passes; summaries land in gpurun_out/ and are copied to profiles/ afterwards
set -o pipefail
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
timeout -k 10 400 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_full.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_full.log
R=$PWD
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/trace.log 2>&1; echo "trace rc=$?" >> $R/gpurun_out/prof/trace.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/pmc_fetch.log 2>&1; echo "pmc1 rc=$?" >> $R/gpurun_out/prof/pmc_fetch.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/pmc_write.log 2>&1; echo "pmc2 rc=$?" >> $R/gpurun_out/prof/pmc_write.log
cd $R
tail -2 gpurun_out/bench_full.log | cut -c1-600
find gpurun_out/prof -name "*.csv" | head -20
tail -2 gpurun_out/prof/trace.log gpurun_out/prof/pmc_fetch.log gpurun_out/prof/pmc_write.log
