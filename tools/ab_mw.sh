#!/bin/bash
# A/B of library variants on the 4-wave batches (profiles/run_mw_batch.py):  bash tools/ab_mw.sh base gjr base gjr
for v in "$@"; do
  if [ $v = base ]; then unset SDDP_LIB; else export SDDP_LIB=/root/repo/build/variants/libsddp_$v.so; fi
  for cfg in "srbd37 20 2048 2" "srbd37 60 1024 2" "lip30 20 4096 2" "srbd61 20 512 1"; do
    python profiles/run_mw_batch.py $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$cfg', round(d['kernel_solves_per_s']), round(d['kernel_ms'],3), d['mean_iters'], d['resources']['scratch_bytes_per_lane'])"
  done
done
