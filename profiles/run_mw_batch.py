"""One cold batch through the 4-wavefront kernel, for rocprofv3 (profiles/collect.sh): python3 profiles/run_mw_batch.py srbd37 20 1024 [waves_per_simd]"""
import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from srbd_horizon_amd import workload
from srbd_horizon_amd.engine import DdpEngine
model, N, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
opts = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3, waves_per_simd=int(sys.argv[4]) if len(sys.argv) > 4 else 1)
print(json.dumps(bench.mw_batch(model, N, B, opts, workload, DdpEngine, reps=2)))
