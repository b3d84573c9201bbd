#!/usr/bin/env python3
"""Adding a residual of your own to the reference's SRBD problem (what `prb.createResidual(...)` with a CasADi expression is
upstream, python/prb.py:184-204; the reference's stage costs sum whatever the container holds, python/ddp.py:183-196).

    python examples/user_residual.py

The analytic HIP models take up to 8 user-declared LINEAR residual rows `sqrt(gain) * (A z - ref)` (problem.LinearTerm;
include/sddp.h `extra_*`): here the left-upper contact point is asked to follow a reference in x / y while its foot swings.
Needs a GPU: the engine has no CPU fallback.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srbd_horizon_amd.ddp import DDPSolver  # noqa: E402
from srbd_horizon_amd.prb import SRBDProblem  # noqa: E402
from srbd_horizon_amd.problem import LinearTerm  # noqa: E402


def main():
    ns, T = 20, 1.0
    pb = SRBDProblem()
    prb = pb.createSRBDProblem(ns, T)                                     # prb.py:16-246, contact_model = 2
    ref = prb.createParameter("c0_xy_ref", 2)                             # a parameter of your own: the term's per-node reference
    start = pb.initial_foot_position[0][0:2]
    for k in range(ns + 1):
        ref.assign(start + np.array([0.04, 0.0]) * k / ns, nodes=[k])     # move the point 4 cm forward over the horizon
    prb.createResidual("c0_xy_tracking", LinearTerm({pb.c[0]: [[1, 0, 0], [0, 1, 0]]}, gain=1e5, ref=ref), nodes=range(1, ns + 1))
    for i in (0, 1):
        pb.cdot_switch[i].assign(0.0)                                     # left foot in swing: its contact points may move
    solver = DDPSolver(prb, dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3))     # dsrbd_example.py:55-58
    solver.setInitialState(pb.getInitialState())
    solver.set_u_warmstart(np.repeat(pb.getStaticInput()[:, None], ns, axis=1))
    ok = solver.solve()
    c0 = solver.getSolutionDict()["c0"]
    print(f"converged {ok} in {int(solver.stats['iters'])} iterations, cost {float(solver.stats['cost']):.4f}")
    print("c0_x over the horizon:", np.round(c0[0, ::4], 4), " reference:", np.round(ref.values[0, ::4], 4))


if __name__ == "__main__":
    main()
