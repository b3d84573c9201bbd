"""TEST INFRASTRUCTURE ONLY -- paper gate for a parallel-in-time Riccati sweep (`latency_mode`).  Not shipped,
not imported by the product; it answers three questions before any kernel is written:

  1. which sweep CAN be written as an associative scan over knots?  Only the Gauss-Newton one (second_order = 0, with or without
     mu): the default sweep adds theta * d2/dudx[(Vx' + Vxx' d) . f] to Qux (oracle/ddp.py:106, the exact bilinear-torque term),
     a stage Hessian that depends on the NEXT knot's value function -- the scan's elements must be known before the scan.
  2. how far do the scan's gains sit from the serial sweep's on the same Gauss-Newton data (same iterate, same mu)?
  3. does a solve that takes its gains from the scan need the same number of iterations?

The scan is the conditional-value-function form (Sarkka & Garcia-Fernandez, "Temporal parallelization of dynamic programming and
linear quadratic control", IEEE TAC 2023), written for this problem's stage  1/2 [x u]^T [Q S^T; S R] [x u] + q.x + r.u,
x+ = Fx x + Fu u + d (d = the multiple-shooting defect):

    V(x, y) = max_l [ zeta + 1/2 x^T J x + eta^T x - 1/2 l^T C l + l^T (A x + b - y) ]
    knot:     A = Fx - Fu R^-1 S   b = d - Fu R^-1 r   C = Fu R^-1 Fu^T   J = Q - S^T R^-1 S   eta = q - S^T R^-1 r
    terminal: A = 0, b = 0, C = 0, J = Vxx_N, eta = Vx_N
    (1) o (2), 1 the earlier one:   M = (I + C1 J2)^-1
              A = A2 M A1            b = A2 M (b1 - C1 eta2) + b2       C = A2 M C1 A2^T + C2
              J = A1^T M^T J2 A1 + J1                                   eta = A1^T M^T (eta2 + J2 b1) + eta1
    suffix product e_k o ... o e_{N-1} o e_N  ->  (J, eta) = (Vxx_k, Vx_k); the gains of knot k follow from (Vxx_{k+1}, Vx_{k+1})
    as in the serial sweep, all knots at once.

    python oracle/pit_riccati.py [instances] [whole-solve instances]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ddp as oddp, models as omodels  # noqa: E402


def knot_data(model, xs, us, P, d, mu):
    N = us.shape[0]
    el = []
    for k in range(N):
        fx, fu = model.f_jac(xs[k], us[k], P[k])
        _, lx, lu, lxx, lux, luu = model.cost_derivs(xs[k], us[k], P[k], k)
        el.append((fx, fu, d[k], lx, lu, lxx, lux, luu + mu * np.eye(model.nu)))
    _, Vx, _, Vxx, _, _ = model.cost_derivs(xs[N], None, P[N], N)
    return el, Vx, Vxx


def element(fx, fu, d, q, r, Q, S, R):
    Ri = np.linalg.inv(R)
    A = fx - fu @ Ri @ S
    return [A, d - fu @ Ri @ r, fu @ Ri @ fu.T, Q - S.T @ Ri @ S, q - S.T @ Ri @ r]


def combine(e1, e2):
    A1, b1, C1, J1, n1 = e1
    A2, b2, C2, J2, n2 = e2
    M = np.linalg.inv(np.eye(A1.shape[0]) + C1 @ J2)
    A2M = A2 @ M
    A1tMt = A1.T @ M.T
    J = A1tMt @ J2 @ A1 + J1
    C = A2M @ C1 @ A2.T + C2
    return [A2M @ A1, A2M @ (b1 - C1 @ n2) + b2, 0.5 * (C + C.T), 0.5 * (J + J.T), A1tMt @ (n2 + J2 @ b1) + n1]


def suffix_scan(el):
    """Hillis-Steele suffix scan: ceil(log2 n) levels, every level combines element k with element k + stride (all k at once)"""
    n = len(el)
    cur = list(el)
    stride, levels, combines = 1, 0, 0
    while stride < n:
        nxt = list(cur)
        for k in range(n - stride):
            nxt[k] = combine(cur[k], cur[k + stride])
            combines += 1
        cur = nxt
        stride *= 2
        levels += 1
    return cur, levels, combines


def backward_pass_scan(model, xs, us, P, d, mu):
    """the Gauss-Newton sweep of oracle/ddp.py:83 (theta = 0) with the value functions from the scan"""
    N = us.shape[0]
    nx, nu = model.nx, model.nu
    kd, VxN, VxxN = knot_data(model, xs, us, P, d, mu)
    for e in kd:
        try:
            np.linalg.cholesky(e[7])
        except np.linalg.LinAlgError:
            return (False,) + (None,) * 9
    el = [element(*e) for e in kd]
    z = np.zeros((nx, nx))
    el.append([z, np.zeros(nx), z, VxxN, VxN])
    sc, levels, combines = suffix_scan(el)
    K = np.zeros((N, nu, nx)); kff = np.zeros((N, nu))
    dV1 = dV2 = G1 = G2 = 0.0
    qu_inf = 0.0
    for k in range(N):                       # independent per knot
        fx, fu, dk, lx, lu, lxx, lux, luu = kd[k]
        Vxx, Vx = sc[k + 1][3], sc[k + 1][4]
        G1 += dk @ Vx
        G2 += 0.5 * dk @ Vxx @ dk
        vp = Vx + Vxx @ dk
        Qu = lu + fu.T @ vp
        Qux = lux + fu.T @ Vxx @ fx
        Quu = luu + fu.T @ Vxx @ fu
        try:
            L = np.linalg.cholesky(Quu)
        except np.linalg.LinAlgError:
            return (False,) + (None,) * 9
        sol = -np.linalg.solve(L.T, np.linalg.solve(L, np.column_stack([Qu, Qux])))
        kff[k] = sol[:, 0]; K[k] = sol[:, 1:]
        dV1 += kff[k] @ Qu
        dV2 += 0.5 * kff[k] @ Quu @ kff[k]
        qu_inf = max(qu_inf, float(np.max(np.abs(Qu))))
    return True, K, kff, dV1, dV2, G1, G2, sc[0][4], sc[0][3], qu_inf


def solve_with(model, x0, P, xs_ws, us_ws, opt, sweep):
    """oracle/ddp.py:140 with second_order = 0 and the sweep swapped (same globalisation, same line search)"""
    us = np.array(us_ws, float); N = us.shape[0]
    xs = np.array(xs_ws, float); xs[0] = x0
    d = oddp.defects(model, xs, us, P); J = oddp.total_cost(model, xs, us, P); gap = float(np.sum(np.abs(d)))
    mu = opt.mu0; rho = 0.0; iters = 0; status = 1; conv = False
    alphas = []
    while iters < opt.max_iters:
        while True:
            ok, K, kff, dV1, dV2, G1, G2, _, _, _ = sweep(model, xs, us, P, d, mu)
            if ok:
                break
            mu = max(mu, 0.0) * 10 + opt.mu_min
            if mu > opt.mu_max:
                return iters, False, 2, J, xs, us, alphas
        expected = -(dV1 + dV2)
        if expected < opt.cost_reduction_ths and gap <= opt.gap_tol:
            conv, status = True, 0
            break
        A1 = dV1 + G1; B2 = dV2 + G2
        if gap > 0:
            rho = max(rho, 2 * max(A1, A1 + B2, 0.0) / gap)
        a = opt.alpha_0; acc = False; slack = 1e-13 * (abs(J) + rho * gap)
        while a >= opt.alpha_converge_threshold:
            xn, un, Jn = oddp.forward_pass(model, x0, xs, us, P, d, K, kff, a)
            pred = a * A1 + a * a * B2 - a * rho * gap
            dphi = (Jn + rho * (1 - a) * gap) - (J + rho * gap)
            if np.isfinite(Jn) and dphi <= opt.beta * pred + slack:
                acc = True
                break
            a *= opt.line_search_decrease_factor
        if not acc:
            conv = bool(gap <= opt.gap_tol and expected <= opt.cost_reduction_ths * max(1, abs(J))); status = 0 if conv else 4
            break
        alphas.append(a)
        dJ = J - Jn; xs, us, J = xn, un, Jn; d = (1 - a) * d; gap = (1 - a) * gap; iters += 1
        if mu > opt.mu0:
            mu = max(opt.mu0, mu * 0.1)
        if abs(dJ) < opt.cost_reduction_ths and gap <= opt.gap_tol:
            conv, status = True, 0
            break
    return iters, conv, status, J, xs, us, alphas


def serial_gn(model, xs, us, P, d, mu):
    return oddp.backward_pass(model, xs, us, P, d, mu, 0.0, 0)


def main(n_inst, n_solve):
    from srbd_horizon_amd import workload
    N = 30
    batch = workload.make_batch("srbd13", N, np.arange(max(n_inst, n_solve)))
    cst = omodels.RobotConsts(**batch["consts"])
    m = omodels.make_model("srbd13", cst)
    opt = oddp.DdpOptions(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3, second_order=0)
    # ---- 2. gains of the scan against the serial sweep: at the warm start and along the first Gauss-Newton iterates
    rel_K, rel_k, conds, condR = [], [], [], []
    for b in range(n_inst):
        x0, P = batch["x0"][b], batch["params"][b]
        xs = np.array(batch["xs"][b]); xs[0] = x0
        us = np.array(batch["us"][b])
        d = oddp.defects(m, xs, us, P)
        for it in range(4):
            ok, K, kff, *_ = serial_gn(m, xs, us, P, d, opt.mu0)
            ok2, K2, kff2, *_ = backward_pass_scan(m, xs, us, P, d, opt.mu0)
            if not (ok and ok2 and np.all(np.isfinite(K)) and np.all(np.isfinite(K2))):
                break
            rel_K.append(np.max(np.abs(K - K2)) / np.max(np.abs(K)))
            rel_k.append(np.max(np.abs(kff - kff2)) / max(np.max(np.abs(kff)), 1e-300))
            kd, VxN, VxxN = knot_data(m, xs, us, P, d, opt.mu0)
            condR.append(max(np.linalg.cond(e[7]) for e in kd))
            el = [element(*e) for e in kd]
            conds.append(max(np.linalg.cond(np.eye(m.nx) + el[k][2] @ el[k + 1][3]) for k in range(N - 1)))
            xn, un, Jn = oddp.forward_pass(m, x0, xs, us, P, d, K, kff, 0.25)      # a fixed short step: other iterates, not a solve
            if not (np.isfinite(Jn) and np.all(np.isfinite(xn))):
                break
            xs, us, d = xn, un, 0.75 * d
    rel_K, rel_k = np.array(rel_K), np.array(rel_k)
    print(f"gains, scan against serial Gauss-Newton sweep, {len(rel_K)} sweeps over {n_inst} instances (srbd13, N = {N}):")
    for name, v in (("K   ", rel_K), ("kff ", rel_k)):
        print(f"  rel linf {name}: median {np.median(v):.2e}  p90 {np.quantile(v, 0.9):.2e}  max {v.max():.2e}")
    print(f"  cond(R = luu + mu I): max {max(condR):.2e};  cond(I + C1 J2) at the first level: max {max(conds):.2e}")
    _, levels, combines = suffix_scan([element(*e) for e in knot_data(m, xs, us, P, d, opt.mu0)[0]] + [[np.zeros((13, 13)), np.zeros(13), np.zeros((13, 13)), np.eye(13), np.zeros(13)]])
    print(f"  scan shape: {levels} levels, {combines} combines (Hillis-Steele) for {N + 1} elements")
    # ---- 3. whole solves in Gauss-Newton mode: serial sweep against scan sweep
    t = time.time()
    its = []
    for b in range(n_solve):
        a = solve_with(m, batch["x0"][b], batch["params"][b], batch["xs"][b], batch["us"][b], opt, serial_gn)
        s = solve_with(m, batch["x0"][b], batch["params"][b], batch["xs"][b], batch["us"][b], opt, backward_pass_scan)
        its.append((a[0], s[0], a[2], s[2], np.max(np.abs(a[4] - s[4])), a[6] == s[6]))
    its = np.array(its, dtype=object)
    ia, isc = its[:, 0].astype(int), its[:, 1].astype(int)
    print(f"whole solves (Gauss-Newton mode), {n_solve} instances, {time.time() - t:.0f} s:")
    print(f"  iterations serial mean {ia.mean():.2f} max {ia.max()};  scan mean {isc.mean():.2f} max {isc.max()};  differ {int(np.sum(ia != isc))}"
          f" (same alpha sequence: {int(np.sum(its[:, 5].astype(bool)))});  status differ {int(np.sum(its[:, 2] != its[:, 3]))}")
    print(f"  linf(x) between the two optima: median {np.median(its[:, 4].astype(float)):.2e} max {np.max(its[:, 4].astype(float)):.2e}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 32)
