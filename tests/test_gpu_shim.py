"""The pyddp-shaped module (srbd_horizon_amd/pyddp_hip.py) called exactly as reference python/ddp.py calls pyddp (:14, :93-94,
:101, :106, :113-123), and costs changed through the problem façade (function container -> model constants), on the GPU."""
import numpy as np
import pytest

from oracle import ddp as oddp, models as omodels
from srbd_horizon_amd import pyddp_hip as pyddp, workload
from srbd_horizon_amd.ddp import DDPSolver
from srbd_horizon_amd.engine import DdpEngine
from srbd_horizon_amd.prb import SRBD13Problem

pytestmark = pytest.mark.gpu

EX = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)          # dsrbd_example.py:55-58


@pytest.mark.parametrize("model,N", [("srbd13", 30), ("srbd37", 20), ("lip30", 20)])
def test_pyddp_call_shapes_give_the_engines_result(model, N):
    batch = workload.make_batch(model, N, [4])
    nx, nu, npar = batch["x0"].shape[1], batch["us"].shape[2], batch["params"].shape[2]
    # ---- the calls of ddp.py, in its order
    ddp_opts = pyddp.DdpSolverOptions()                                      # :14
    ddp_opts.max_iters = EX["max_iters"]                                     # :18-19
    ddp_opts.alpha_converge_threshold = EX["alpha_converge_threshold"]       # :23-25
    ddp_opts.beta = EX["beta"]                                               # :29-31
    f_list, L_list, L_term = pyddp.model_functions(model, N, batch["consts"])     # stands for :83-87
    solver = pyddp.DdpSolver(nx, nu, f_list, L_list, L_term, ddp_opts)      # :93-94
    solver.set_initial_state(batch["x0"][0])                                 # :123
    solver.set_x_warmstart(batch["xs"][0].T)                                 # :117  [nx, N+1]
    solver.set_u_warmstart(batch["us"][0].T)                                 # :114  [nu, N]
    param_values_list = [[float(v) for v in batch["params"][0, node]] for node in range(N + 1)]    # :98-99, :165-177
    x, u = solver.solve(param_values_list)                                   # :101
    assert x.shape == (nx, N + 1) and u.shape == (nu, N)                     # variable-major, node index last (:139, :146)
    assert solver.is_converged() is True                                     # :106
    # ---- the same problem through the batched engine
    eng = DdpEngine(model, N, 1, opts=EX, consts=batch["consts"])
    eng.set_initial_state(batch["x0"]); eng.set_x_warmstart(batch["xs"]); eng.set_u_warmstart(batch["us"])
    xe, ue = eng.solve(batch["params"])
    np.testing.assert_array_equal(x, xe[0].T)
    np.testing.assert_array_equal(u, ue[0].T)
    # the solver object persists across ticks and keeps its solution as the next warm start (dsrbd_example.py:59)
    x2, u2 = solver.solve(param_values_list)
    assert solver.stats["iters"] <= 1 and np.max(np.abs(x2 - x)) <= 1e-6
    with pytest.raises(ValueError):
        solver.solve(param_values_list[:-1])


def test_costs_changed_through_the_facade_reach_the_kernel():
    """A residual removed and a gain changed with the problem surface (prb.removeCostFunction / createResidual): the solve must
    be the oracle's solve with those constants."""
    N = 30
    batch = workload.make_batch("srbd13", N, [2])
    pb = SRBD13Problem(); prb = pb.createSRBD13Problem(N, N * 0.05)
    prb.removeCostFunction("w_tracking")
    prb.createResidual("rz_tracking", pb_term("rz_tracking", "r_tracking_gain", 5e3), nodes=range(1, N + 1))
    for name, par in prb.getParameters().items():                            # the batch's plan into the problem's parameters
        pass
    P = batch["params"][0]
    off = 0
    for par in prb.getParameters().values():
        par.values[:, :] = P[:, off:off + par.getDim()].T
        off += par.getDim()
    solver = DDPSolver(prb, EX)
    solver.setInitialState(batch["x0"][0])
    solver.set_x_warmstart(batch["xs"][0].T); solver.set_u_warmstart(batch["us"][0].T)
    assert solver.solve()
    sol = solver.getSolutionDict()
    cst = omodels.RobotConsts(w_tracking_gain=0.0, r_tracking_gain=5e3)
    r = oddp.solve(omodels.make_model("srbd13", cst), batch["x0"][0], P, batch["xs"][0], batch["us"][0], oddp.DdpOptions(**EX))
    assert r.converged and solver.stats["iters"] == r.iters
    assert np.max(np.abs(sol["x_opt"].T - r.xs)) <= 1e-6 and np.max(np.abs(sol["u_opt"].T - r.us)) <= 1e-6
    # and it is a different problem from the default one
    r0 = oddp.solve(omodels.make_model("srbd13"), batch["x0"][0], P, batch["xs"][0], batch["us"][0], oddp.DdpOptions(**EX))
    assert np.max(np.abs(r0.xs - r.xs)) > 1e-4


def pb_term(key, ckey, gain):
    from srbd_horizon_amd.problem import Term
    return Term(key, ckey, gain)


def test_c_host_program_gets_the_same_solve(tmp_path):
    """examples/c_abi_solve.c (plain C on include/sddp.h, no Python in its process) against the ctypes path and the numpy oracle on
    the same problem: one standing srbd13 robot, cold solve, then warm-started ticks on device-resident data."""
    import json
    import subprocess
    from oracle import ddp as oddp, models as omodels
    from srbd_horizon_amd import _lib
    from tests.test_abi import build_c_host
    out = json.loads(subprocess.run([build_c_host(tmp_path), "12"], check=True, capture_output=True, text=True).stdout.strip().splitlines()[-1])
    N = 30
    c = _lib.default_consts()
    feet = np.array(list(c.feet)).reshape(4, 3)
    x0 = np.zeros(13); x0[0], x0[1], x0[2], x0[6], x0[7] = 0.01, -0.005, c.com[2] + 0.01, 1.0, 0.02
    P = np.zeros((N + 1, 19)); P[:, 6] = 1e2; P[:, 10] = 1.0
    P[:, 11:14] = 0.5 * (feet[0] + feet[1]); P[:, 14:17] = 0.5 * (feet[2] + feet[3]); P[:, 17:19] = 1.0
    xs = np.repeat(x0[None], N + 1, axis=0)
    us = np.zeros((N, 6)); us[:, 2] = us[:, 5] = c.m * 9.81 / c.force_scaling / 2.0
    opts = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)
    r = oddp.solve(omodels.make_model("srbd13"), x0, P, xs, us, oddp.DdpOptions(**opts))
    assert out["cold_iters"] == r.iters and out["cold_converged"] == int(r.converged)
    assert abs(out["cold_cost"] - r.cost) <= 1e-9 * abs(r.cost)
    eng = DdpEngine("srbd13", N, 1, opts=opts)
    eng.set_initial_state(x0[None]); eng.set_x_warmstart(xs[None]); eng.set_u_warmstart(us[None])
    x, u = eng.solve(P[None])
    assert eng.stats["cost"][0] == out["cold_cost"]                       # the same library, the same bits
    eng.set_params(P[None])
    for _ in range(12):
        eng.advance(P[None, -1], x[:, 1])
        x, u = eng.solve_resident()
    assert eng.stats["cost"][0] == out["tick_cost"] and int(eng.stats["iters"][0]) == out["tick_iters"]
    np.testing.assert_array_equal(u[0, 0], np.array(out["u0"]))
    assert out["tick_converged"] == 1 and out["ms_per_tick"] > 0.0        # (the time itself is printed, not asserted)
    print("C host: ms per warm-started tick", out["ms_per_tick"], "iterations", out["tick_iters"])
