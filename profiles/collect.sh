#!/bin/bash
# Collects the judged profile set of one round on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh r03 [steps]
# Headline kernel (solve_kernel_w2<srbd13>): 1 kernel trace + stats, 3 separate PMC passes (never combined with trace domains)
# of the command the driver runs (--steps 20 --warmup 5), then its full bench line.  4-wavefront kernel (solve_kernel_mw<srbd37>):
# the same four passes over one cold batch at the reference's own size (ns = 20) and at BASELINE configs[4] (N = 60), two
# workgroups per CU (waves_per_simd = 2); and over the same problem at its code-default contact_model = 4 (srbd61, one per CU).
set -e
R=${1:-r03}
export TMPDIR=/tmp
O=gpurun_out/prof_$R
rm -rf $O && mkdir -p $O
STEPS=${2:-20}
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"
passes() {   # passes <tag> <command...>: trace + FETCH + WRITE + SQ, each its own run
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}trace -- "$@" > $O/${tag}trace.log 2>&1
  echo "$tag trace done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}pmc_fetch -- "$@" > $O/${tag}pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}pmc_write -- "$@" > $O/${tag}pmc_write.log 2>&1
  rocprofv3 --pmc $SQ --output-format csv -d $O/${tag}pmc_sq -- "$@" > $O/${tag}pmc_sq.log 2>&1
  echo "$tag pmc done"
}
passes "" python3 bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-extras
passes mw_srbd37_n20_ python3 profiles/run_mw_batch.py srbd37 20 2048 2
passes mw_srbd37_n60_ python3 profiles/run_mw_batch.py srbd37 60 1024 2
passes mw_srbd61_n20_ python3 profiles/run_mw_batch.py srbd61 20 1024 1
python3 bench.py --steps $STEPS --warmup 5 > $O/bench.log 2> $O/bench.err
echo "bench done"
python3 profiles/make_summary.py $R $O $STEPS | tee $O/summary.txt
mkdir -p gpurun_out/profiles_$R && find profiles/$R -maxdepth 1 -type f -exec cp {} gpurun_out/profiles_$R/ \;
