for v in base p12 p24 p40 base; do
  if [ $v = base ]; then unset SDDP_LIB; else export SDDP_LIB=/root/repo/build/variants/libsddp_$v.so; fi
  python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r5_prio_$v.json 2> gpurun_out/r5_prio_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r5_prio_$v.json").read().strip().splitlines()[-1]); print("$v", round(d["value"]), [round(x) for x in d["value_runs"]], d.get("drain"))
PY
done
