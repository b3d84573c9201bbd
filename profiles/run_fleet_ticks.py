"""The warm-started fleet ticks of bench.py's `ms_per_fleet_tick_srbd37` alone (512 srbd37 robots, 28 ticks), printing every tick's
wall time and kernel time: for `rocprofv3 --hip-trace` (which HIP API call does the one-time 20-45 ms host-side stall sit in?)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from srbd_horizon_amd import workload
from srbd_horizon_amd.engine import DdpEngine
model, N, B, ticks = "srbd37", 20, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 28
b = workload.make_batch(model, N, np.arange(B) + 11000)
e = DdpEngine(model, N, B, opts=dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3, waves_per_simd=2))
e.enable_timing(True)
e.set_initial_state(b["x0"]); e.set_x_warmstart(b["xs"]); e.set_u_warmstart(b["us"]); e.set_params(b["params"])
x, u = e.solve_resident()
P = b["params"].copy()
rows = []
for t in range(ticks):
    p_last, x0 = P[:, -1].copy(), x[:, 1].copy()
    P = np.concatenate([P[:, 1:], p_last[:, None]], axis=1)
    t0 = time.perf_counter_ns()
    e.advance(p_last, x0)
    t1 = time.perf_counter_ns()
    e.solve_resident_first()
    t2 = time.perf_counter_ns()
    rows.append(dict(tick=t, advance_ms=(t1 - t0) / 1e6, solve_first_ms=(t2 - t1) / 1e6, kernel_ms=e.last_kernel_ms(), t0_ns=t0, t2_ns=t2))
    x, u, st = e.fetch()
print(json.dumps(rows))
