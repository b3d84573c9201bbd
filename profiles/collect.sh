#!/bin/bash
# Collects the judged profile set of one round on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh r02 [steps]
# 1 kernel trace + stats, 3 separate PMC passes (never combined with trace domains), then the full bench line of the same
# command the driver runs (--steps 20 --warmup 5).
set -e
R=${1:-r02}
export TMPDIR=/tmp
O=gpurun_out/prof_$R
rm -rf $O && mkdir -p $O
STEPS=${2:-20}
CMD="python3 bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $CMD > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $CMD > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/pmc_sq -- $CMD > $O/pmc_sq.log 2>&1
python3 bench.py --steps $STEPS --warmup 5 > $O/bench.log 2>&1
python3 profiles/make_summary.py $R $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/bench.log $STEPS | tee $O/summary.txt
mkdir -p gpurun_out/profiles_$R && cp profiles/$R/* gpurun_out/profiles_$R/
