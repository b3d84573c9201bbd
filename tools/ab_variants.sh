#!/bin/bash
# A/B of library variants on the driver-shaped run (build/variants/libsddp_<name>.so, built with SDDP_LIB / SDDP_CXXFLAGS):
#   bash tools/ab_variants.sh base dma base dma
for v in "$@"; do
  if [ $v = base ]; then unset SDDP_LIB; else export SDDP_LIB=/root/repo/build/variants/libsddp_$v.so; fi
  python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$v.json").read().strip().splitlines()[-1]); print("$v", round(d["value"]), [round(x) for x in d["value_runs"]], round(d["roofline"]["kernel_ms"],2), d["roofline"]["resources"])
PY
done
