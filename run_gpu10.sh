#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$PWD
timeout -k 10 400 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_final.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_final.log
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/trace3 -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-pipelined > $R/gpurun_out/prof/trace3.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_fetch3 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined > $R/gpurun_out/prof/pmc_fetch3.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_write3 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined > $R/gpurun_out/prof/pmc_write3.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/prof/pmc_sq3 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined > $R/gpurun_out/prof/pmc_sq3.log 2>&1
cd $R
tail -n 2 gpurun_out/bench_final.log | cut -c1-200
find gpurun_out/prof -name "*.csv" -newer run_gpu10.sh | grep -v agent_info
