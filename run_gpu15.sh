#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$PWD
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/t12.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t12.log
grep -E "passed|failed|rc=|^FAILED|^E  " gpurun_out/t12.log | tail -6
timeout -k 10 400 python bench.py > gpurun_out/bench_final.log 2>&1; echo "bench rc=$?"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/trace5 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sequential > $R/gpurun_out/prof/trace5.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_fetch5 -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sequential > $R/gpurun_out/prof/pmc_fetch5.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_write5 -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sequential > $R/gpurun_out/prof/pmc_write5.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/prof/pmc_sq5 -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sequential > $R/gpurun_out/prof/pmc_sq5.log 2>&1
cd $R
grep '^{' gpurun_out/bench_final.log | tail -1 | cut -c1-160
