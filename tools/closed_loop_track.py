import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from srbd_horizon_amd.mpc import MpcLoop
for model, ns in (("srbd13", 30), ("srbd37", 20), ("lip30", 20)):
    lp = MpcLoop(model, ns, warm_start="device")
    out = []
    for t in range(201):
        ok, sol = lp.tick("walking", (1.0, 0.0))
        if t in (0, 10, 20, 40, 80, 120, 200):
            out.append((t, np.round(lp.state[:3], 3).tolist(), int(lp.solver.stats["iters"]), bool(ok)))
    print(model, out)
