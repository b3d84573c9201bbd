"""Iteration / rollout distribution over the bench batch (which instances set the batch time)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srbd_horizon_amd import workload
from srbd_horizon_amd.engine import DdpEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
MODEL = sys.argv[2] if len(sys.argv) > 2 else "srbd13"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 30
batch = workload.make_batch(MODEL, N, np.arange(B))
eng = DdpEngine(MODEL, N, B, opts=dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3))
eng.set_initial_state(batch["x0"]); eng.set_x_warmstart(batch["xs"]); eng.set_u_warmstart(batch["us"])
eng.enable_timing(True)
eng.solve(batch["params"])
it, ro = eng.stats["iters"], eng.stats["rollouts"]
print("kernel ms", eng.last_kernel_ms(), "converged", eng.stats["converged"].mean(), "status", np.bincount(eng.stats["status"]))
print("iters: mean %.2f median %d p90 %d p99 %d max %d" % (it.mean(), np.median(it), np.percentile(it, 90), np.percentile(it, 99), it.max()))
print("rollouts: mean %.2f max %d" % (ro.mean(), ro.max()))
print("hist iters (bins of 5):", np.bincount(it // 5))
worst = np.argsort(-it)[:10]
print("worst seeds:", worst, it[worst], ro[worst], "phase", worst % 20)
