"""B = 1 latency of one model on the library SDDP_LIB points at (A/B of kernel mappings):  python tools/latency_b1.py srbd13 30
cold solve of 8 seeds (host-pointer call) and the receding-horizon loop's solve time."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from srbd_horizon_amd import workload
from srbd_horizon_amd.engine import DdpEngine
from srbd_horizon_amd.mpc import MpcLoop

model, N = sys.argv[1], int(sys.argv[2])
opts = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)
e1 = DdpEngine(model, N, 1, opts=opts)
cold = []
for seed in range(8):
    b1 = workload.make_batch(model, N, [seed])
    t = []
    for _ in range(12):
        e1.set_initial_state(b1["x0"]); e1.set_x_warmstart(b1["xs"]); e1.set_u_warmstart(b1["us"])
        t1 = time.perf_counter()
        e1.solve(b1["params"])
        t.append(1e3 * (time.perf_counter() - t1))
    cold.append((seed, int(e1.stats["iters"][0]), float(np.median(t[2:]))))
loop = MpcLoop(model, N, warm_start="device")
tick, its = [], []
for i in range(220):
    t1 = time.perf_counter()
    loop.tick("walking", (1.0, 0.0))
    tick.append(1e3 * (time.perf_counter() - t1))
    its.append(int(loop.solver.stats["iters"]))
print(json.dumps({"lib": os.environ.get("SDDP_LIB", "default"), "kernel": e1.kernel_info() if hasattr(e1, "kernel_info") else None,
                  "cold": cold, "cold_ms_per_iter": float(np.median([c[2] / max(c[1], 1) for c in cold])),
                  "tick_median": float(np.median(tick[20:])), "solve_median": float(np.median(loop.solve_ms[20:])),
                  "mean_iters": float(np.mean(its[20:]))}))
