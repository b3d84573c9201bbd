import numpy as np, sys
sys.path.insert(0, '.')
from srbd_horizon_amd import workload
from srbd_horizon_amd.engine import DdpEngine
N=30; seeds=[0,5,13]
batch = workload.make_batch("srbd13", N, seeds)
rng = np.random.default_rng(1)
xs = batch["xs"] + 0.01 * rng.standard_normal(batch["xs"].shape)
us = batch["us"] + 0.01 * rng.standard_normal(batch["us"].shape)
xs[:, 0] = batch["x0"]
eng = DdpEngine("srbd13", N, len(seeds), opts=dict(max_iters=100))
eng.set_initial_state(batch["x0"]); eng.set_x_warmstart(xs); eng.set_u_warmstart(us)
kff, K, scal = eng.backward(batch["params"], mu=0.0)
xg, ug, Jg = eng.forward(batch["params"], 0.5)
print("J", Jg)
print("nan x rows", np.argwhere(np.isnan(xg))[:20].tolist())
print("nan u rows", np.argwhere(np.isnan(ug))[:20].tolist())
