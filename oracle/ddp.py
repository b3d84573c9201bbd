"""TEST INFRASTRUCTURE ONLY -- numpy float64 multiple-shooting DDP (the engine the reference delegates to
``pyddp``: call sites ddp.py:93-94, :101, :106).

PARITY UNPINNED: ``pyddp`` is absent and unpinned, so the algorithm below is the documented textbook MS-DDP of
SURVEY.md App. C / DESIGN.md "Algorithm" (Gauss-Newton Hessians; option names and meaning follow the
``DdpSolverOptions`` fields the reference sets at ddp.py:14-35).  The HIP engine implements exactly these
steps; tests compare the two iteration by iteration.

    defects   d_{k+1} = f(x_k,u_k) - x_{k+1}
    backward  v' = Vx+ + Vxx+ d ; Q* = l* + F^T (.) ; Quu += mu I ; k = -Quu^-1 Qu ; K = -Quu^-1 Qux
              second_order 1: Qux += theta * (v'.f_ux restricted to the bilinear torque (c-r) x f); 2: Q += theta * (v'.f_zz +
              exact - Gauss-Newton cost Hessian), all blocks; theta = 1 after a full step (alpha == alpha_0) was accepted,
              else 0; a failed sweep or line search with theta = 1 is redone with 0
              Vx = Qx + Qux^T k ; Vxx = Qxx + Qux^T K (symmetrised)
              dV1 = sum k^T Qu ; dV2 = 1/2 sum k^T Quu k ; G1 = sum d^T Vx+ ; G2 = 1/2 sum d^T Vxx+ d
    forward   xh_0 = x0 ; uh_k = u_k + a k_k + K_k (xh_k - x_k) ; xh_{k+1} = f(xh_k,uh_k) - (1-a) d_{k+1}
    accept    phi = J + rho*||d||_1 ;  phi(a) - phi(0) <= beta * (a (dV1+G1) + a^2 (dV2+G2) - a rho ||d||_1) + slack
    stop      expected reduction -(dV1+dV2) < cost_reduction_ths (and gaps closed)  -> converged
              a < alpha_converge_threshold  -> stop, status 4; converged only with closed gaps and
                                               expected <= cost_reduction_ths * max(1, |J|)
              |J_old - J_new| < cost_reduction_ths (and gaps closed)               -> converged
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class DdpOptions:
    max_iters: int = 100                      # ddp.py:17-19
    alpha_0: float = 1.0                      # ddp.py:20-22
    alpha_converge_threshold: float = 1e-1    # ddp.py:23-25
    line_search_decrease_factor: float = 0.5  # ddp.py:26-28
    beta: float = 1e-4                        # ddp.py:29-31
    cost_reduction_ths: float = 1e-6          # ddp.py:32-33 (engine default unpinned)
    mu0: float = 0.0                          # ddp.py:34-35 (engine default unpinned)
    initial_rollout: bool = False             # True: single shooting (x warm start ignored)
    second_order: int = 1                     # 1: exact bilinear-torque term v'.f_ux once full steps are accepted; 2: full second
                                              # order (v'.f_zz and the exact cost Hessian, Model.second_order_full); 0: Gauss-Newton
    gap_tol: float = 1e-9
    mu_min: float = 1e-6
    mu_max: float = 1e12


@dataclass
class DdpResult:
    xs: np.ndarray
    us: np.ndarray
    cost: float
    iters: int
    converged: bool
    alpha: float
    gap: float
    mu: float
    status: int          # 0 ok, 1 max_iters, 2 regularisation overflow, 3 non-finite, 4 line search exhausted
    trace: list


def total_cost(model, xs, us, P):
    N = us.shape[0]
    J = 0.0
    for k in range(N):
        J += model.cost(xs[k], us[k], P[k], k)
    return J + model.cost(xs[N], None, P[N], N)


def defects(model, xs, us, P):
    N = us.shape[0]
    return np.array([model.f(xs[k], us[k], P[k]) - xs[k + 1] for k in range(N)])


def rollout_open_loop(model, x0, us, P):
    N = us.shape[0]
    xs = np.zeros((N + 1, model.nx))
    xs[0] = x0
    for k in range(N):
        xs[k + 1] = model.f(xs[k], us[k], P[k])
    return xs


def backward_pass(model, xs, us, P, d, mu, theta=0.0, mode=1):
    """-> ok, K [N,nu,nx], kff [N,nu], dV1, dV2, G1, G2, Vx0, Vxx0, qu_inf"""
    N = us.shape[0]
    nx, nu = model.nx, model.nu
    K = np.zeros((N, nu, nx))
    kff = np.zeros((N, nu))
    _, Vx, _, Vxx, _, _ = model.cost_derivs(xs[N], None, P[N], N)
    dV1 = dV2 = G1 = G2 = 0.0
    qu_inf = 0.0
    for k in range(N - 1, -1, -1):
        fx, fu = model.f_jac(xs[k], us[k], P[k])
        _, lx, lu, lxx, lux, luu = model.cost_derivs(xs[k], us[k], P[k], k)
        G1 += d[k] @ Vx
        G2 += 0.5 * d[k] @ Vxx @ d[k]
        vp = Vx + Vxx @ d[k]
        Qx = lx + fx.T @ vp
        Qu = lu + fu.T @ vp
        Qxx = lxx + fx.T @ Vxx @ fx
        Qux = lux + fu.T @ Vxx @ fx
        Quu = luu + fu.T @ Vxx @ fu + mu * np.eye(nu)
        if theta and mode == 2:
            S = theta * model.second_order_full(xs[k], us[k], P[k], k, vp)
            Qxx = Qxx + S[:nx, :nx]
            Qux = Qux + S[nx:, :nx]
            Quu = Quu + S[nx:, nx:]
        elif theta:
            Qux = Qux + theta * model.second_order_ux(xs[k], us[k], P[k], vp)
        try:
            L = np.linalg.cholesky(Quu)
        except np.linalg.LinAlgError:
            return False, K, kff, 0, 0, 0, 0, None, None, 0
        sol = -np.linalg.solve(L.T, np.linalg.solve(L, np.column_stack([Qu, Qux])))
        kff[k] = sol[:, 0]
        K[k] = sol[:, 1:]
        dV1 += kff[k] @ Qu
        dV2 += 0.5 * kff[k] @ Quu @ kff[k]
        Vx = Qx + Qux.T @ kff[k]
        Vxx = Qxx + Qux.T @ K[k]
        Vxx = 0.5 * (Vxx + Vxx.T)
        qu_inf = max(qu_inf, float(np.max(np.abs(Qu))))
    return True, K, kff, dV1, dV2, G1, G2, Vx, Vxx, qu_inf


def forward_pass(model, x0, xs, us, P, d, K, kff, alpha):
    N = us.shape[0]
    xn = np.zeros_like(xs)
    un = np.zeros_like(us)
    xn[0] = x0
    J = 0.0
    for k in range(N):
        un[k] = us[k] + alpha * kff[k] + K[k] @ (xn[k] - xs[k])
        J += model.cost(xn[k], un[k], P[k], k)
        xn[k + 1] = model.f(xn[k], un[k], P[k]) - (1.0 - alpha) * d[k]
    J += model.cost(xn[N], None, P[N], N)
    return xn, un, J


def solve(model, x0, P, xs_ws, us_ws, opt: DdpOptions | None = None) -> DdpResult:
    opt = opt or DdpOptions()
    us = np.array(us_ws, dtype=float)
    N = us.shape[0]
    if opt.initial_rollout:
        xs = rollout_open_loop(model, x0, us, P)
        d = np.zeros((N, model.nx))
    else:
        xs = np.array(xs_ws, dtype=float)
        xs[0] = x0
        d = defects(model, xs, us, P)
    J = total_cost(model, xs, us, P)
    gap = float(np.sum(np.abs(d)))
    mu = opt.mu0
    rho = 0.0
    alpha = 0.0
    theta = 0.0
    iters = 0
    converged = False
    status = 1
    trace = []
    if not np.isfinite(J):
        return DdpResult(xs, us, J, 0, False, 0.0, gap, mu, 3, trace)
    while iters < opt.max_iters:
        # ---- backward sweep (regularisation bump on a non-PD Quu: this is what mu0 is for, ddp.py:34-35)
        while True:
            ok, K, kff, dV1, dV2, G1, G2, Vx0, Vxx0, qu_inf = backward_pass(model, xs, us, P, d, mu, theta, int(opt.second_order))
            if ok:
                break
            if theta:
                theta = 0.0                       # second-order term made Quu indefinite: plain Gauss-Newton sweep
                continue
            mu = max(mu, 0.0) * 10.0 + opt.mu_min
            if mu > opt.mu_max:
                return DdpResult(xs, us, J, iters, False, alpha, gap, mu, 2, trace)
        expected = -(dV1 + dV2)
        if expected < opt.cost_reduction_ths and gap <= opt.gap_tol:
            converged, status = True, 0
            break
        A1 = dV1 + G1
        B2 = dV2 + G2
        if gap > 0.0:
            rho = max(rho, 2.0 * max(A1, A1 + B2, 0.0) / gap)
        # ---- backtracking line search
        a = opt.alpha_0
        accepted = False
        slack = 1e-13 * (abs(J) + rho * gap)
        while a >= opt.alpha_converge_threshold:
            xn, un, Jn = forward_pass(model, x0, xs, us, P, d, K, kff, a)
            pred = a * A1 + a * a * B2 - a * rho * gap
            dphi = (Jn + rho * (1.0 - a) * gap) - (J + rho * gap)
            if np.isfinite(Jn) and dphi <= opt.beta * pred + slack:
                accepted = True
                break
            a *= opt.line_search_decrease_factor
        if not accepted:
            if theta:
                theta = 0.0                       # redo this iteration with the plain Gauss-Newton step
                continue
            # alpha fell below alpha_converge_threshold (App. C: "stop").  No step length decreases the merit function any
            # more; that is an optimum only if the multiple-shooting gaps are closed and the model predicts (next to) no
            # decrease either: expected <= cost_reduction_ths RELATIVE to the cost (status 0).  Otherwise: stalled, status 4.
            converged = bool(gap <= opt.gap_tol and expected <= opt.cost_reduction_ths * max(1.0, abs(J)))
            status = 0 if converged else 4
            alpha = 0.0
            break
        alpha = a
        theta = 1.0 if (opt.second_order and a == opt.alpha_0) else 0.0
        dJ = J - Jn
        xs, us, J = xn, un, Jn
        d = (1.0 - a) * d
        gap = (1.0 - a) * gap
        iters += 1
        if mu > opt.mu0:
            mu = max(opt.mu0, mu * 0.1)
        trace.append(dict(it=iters, cost=J, alpha=a, expected=expected, gap=gap, qu_inf=qu_inf, mu=mu, dJ=dJ, theta=theta))
        if abs(dJ) < opt.cost_reduction_ths and gap <= opt.gap_tol:
            converged, status = True, 0
            break
    return DdpResult(xs, us, J, iters, converged, alpha, gap, mu, status, trace)
