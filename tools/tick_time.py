"""Per-tick solve time of the receding-horizon loop (MpcLoop) for one model: median / mean ms and iterations per tick."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from srbd_horizon_amd.mpc import MpcLoop

model = sys.argv[1] if len(sys.argv) > 1 else "srbd37"
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 120
loop = MpcLoop(model=model, ns=int(sys.argv[3]) if len(sys.argv) > 3 else 20, warm_start=sys.argv[4] if len(sys.argv) > 4 else "shift")
for _ in range(10):
    loop.tick("walking", (1.0, 0.0))
loop.solve_ms.clear()
it, full = [], []
for _ in range(ticks):
    t0 = time.perf_counter()
    loop.tick("walking", (1.0, 0.0))
    full.append(1e3 * (time.perf_counter() - t0))
    it.append(int(loop.solver.stats["iters"]))
ms = np.array(loop.solve_ms)
print(f"{model} [{loop.warm_start}]: whole tick median {np.median(full):.3f} | solve call: ms/tick median {np.median(ms):.3f} mean {ms.mean():.3f} max {ms.max():.3f} iters mean {np.mean(it):.2f}")
