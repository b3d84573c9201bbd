"""Pins for the oracle's DDP restatement that need no reference engine: the LIP LQ known answer and
optimality properties of SRBD solves."""
import numpy as np

from oracle import ddp as oddp
from oracle import models as omodels
from srbd_horizon_amd import workload


def _lq_kkt_solution(m, x0, P, N):
    """Dense KKT solve of  min sum ||Jx x_k + Ju u_k + r0||^2  s.t.  x_{k+1} = A x_k + B u_k + b  (LIP is exactly this)."""
    nx, nu = m.nx, m.nu
    fx, fu = m.f_jac(np.zeros(nx), np.zeros(nu), P[0])
    b = m.f(np.zeros(nx), np.zeros(nu), P[0])
    nz = (N + 1) * nx + N * nu
    ix = lambda k: slice(k * nx, (k + 1) * nx)
    iu = lambda k: slice((N + 1) * nx + k * nu, (N + 1) * nx + (k + 1) * nu)
    H = np.zeros((nz, nz)); g = np.zeros(nz)
    for k in range(N + 1):
        u = None if k == N else np.zeros(nu)
        r0, Jx, Ju = m.residual_jac(np.zeros(nx), u, P[k], k)
        if k == N:
            J = np.zeros((r0.size, nz)); J[:, ix(k)] = Jx
        else:
            J = np.zeros((r0.size, nz)); J[:, ix(k)] = Jx; J[:, iu(k)] = Ju
        H += 2 * J.T @ J
        g += 2 * J.T @ r0
    nc = (N + 1) * nx
    Cm = np.zeros((nc, nz)); c = np.zeros(nc)
    Cm[0:nx, ix(0)] = np.eye(nx); c[0:nx] = x0
    for k in range(N):
        rows = slice((k + 1) * nx, (k + 2) * nx)
        Cm[rows, ix(k + 1)] = np.eye(nx); Cm[rows, ix(k)] = -fx; Cm[rows, iu(k)] = -fu; c[rows] = b
    KKT = np.block([[H, Cm.T], [Cm, np.zeros((nc, nc))]])
    sol = np.linalg.solve(KKT, np.concatenate([-g, c]))
    z = sol[:nz]
    return z[:(N + 1) * nx].reshape(N + 1, nx), z[(N + 1) * nx:].reshape(N, nu)


def test_lip_one_full_step_is_the_kkt_solution():
    """prb.py:317-328 + :379-402: linear dynamics, quadratic cost => one alpha = 1 DDP iteration is the exact optimum."""
    N = 20
    batch = workload.make_batch("lip30", N, [5])
    m = omodels.make_model("lip30")
    x0, P = batch["x0"][0], batch["params"][0]
    xk, uk = _lq_kkt_solution(m, x0, P, N)
    r = oddp.solve(m, x0, P, batch["xs"][0], batch["us"][0], oddp.DdpOptions(max_iters=5, cost_reduction_ths=1e-9))
    assert r.trace[0]["alpha"] == 1.0
    assert r.iters <= 2 and r.converged                    # iteration 2 (if any) only confirms stationarity
    np.testing.assert_allclose(r.xs, xk, rtol=0, atol=1e-8)
    np.testing.assert_allclose(r.us, uk, rtol=0, atol=1e-7)
    assert abs(r.cost - oddp.total_cost(m, xk, uk, P)) <= 1e-8 * max(1.0, r.cost)


def test_srbd_solve_properties():
    """Optimality residuals of an SRBD solve: monotone merit, gaps closed, ||Qu|| -> 0, expected reduction -> 0."""
    N = 30
    batch = workload.make_batch("srbd13", N, [0])
    m = omodels.make_model("srbd13")
    r = oddp.solve(m, batch["x0"][0], batch["params"][0], batch["xs"][0], batch["us"][0],
                   oddp.DdpOptions(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3))
    assert r.converged and r.gap == 0.0
    costs = [t["cost"] for t in r.trace]
    assert all(b <= a + 1e-9 * abs(a) for a, b in zip(costs[1:], costs[2:]))      # after the gaps close: monotone
    # with the second-order torque term the tail is (near-)quadratic: the model predicts the last decreases almost exactly
    assert r.trace[-1]["expected"] < 1.0 and r.trace[-1]["qu_inf"] < r.trace[0]["qu_inf"] * 1e-4
    assert abs(r.trace[-1]["dJ"] / r.trace[-1]["expected"] - 1.0) < 0.05
    # ... and it beats plain Gauss-Newton on the same instance
    r_gn = oddp.solve(m, batch["x0"][0], batch["params"][0], batch["xs"][0], batch["us"][0],
                      oddp.DdpOptions(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3, second_order=False))
    assert r_gn.converged and abs(r_gn.cost - r.cost) <= 1e-9 * r.cost and r.iters <= r_gn.iters
    d = oddp.defects(m, r.xs, r.us, batch["params"][0])
    assert np.max(np.abs(d)) < 1e-12
    np.testing.assert_array_equal(r.xs[0], batch["x0"][0])
