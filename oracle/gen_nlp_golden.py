"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/nlp_*.npz: the optimum of the discrete problem the DDP solves, found WITHOUT
any DDP code: the stand-in for north_star's "CasADi solve" (casadi / pyddp are absent).

  problem   the sympy restatement of SURVEY App. A in tests/sym_models.py (written from the equations of prb.py:92-204, not from
            oracle/models.py), assembled as ddp.py:179-230 does: stage cost of node k = sum ||residual||^2 (+ 1e6 ||g||^2 inside
            the residual rows), input residuals on nodes 0..N-1, state residuals on nodes 1..N, no constraints at the terminal
            node; Euler step x+ = x + dt xdot(x, u).  Every derivative is symbolic (lambdified).
  method    direct transcription: unknowns z = (x_1..x_N, u_0..u_{N-1}), equality constraints = the N Euler defects, solved as a
            plain NLP by scipy.optimize.minimize(method="trust-constr") with the exact Lagrangian Hessian, then polished by full
            Newton steps on the KKT system (sparse LU) until the stationarity and feasibility residuals are at rounding level.
            No Riccati recursion, no rollout, no line search over the ladder of ddp.py:20-28, none of oracle/ddp.py.
  output    x [N+1, nx], u [N, nu], cost, KKT residuals, and the inputs (x0, params, warm start, constants) so that the tests
            need neither sympy nor the workload generator to stay as they are.

Run in the build container (sympy + scipy; minutes):   python oracle/gen_nlp_golden.py
"""
import json
import os
import sys
import time

import numpy as np
import scipy.optimize as sopt
import scipy.sparse as sps
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from tests import sym_models  # noqa: E402


class Transcription:
    """The NLP of one instance.  z = [x_1 .. x_N | u_0 .. u_{N-1}]; x_0 is data."""

    def __init__(self, name, N, x0, P, rel_vel=True, rows_key=None):
        self.sym, self.cst = sym_models.symbolic(name, 0, 1.0, rel_vel, rows_key)
        self.so = sym_models.second_order_symbolic(name, 0, 1.0, rel_vel, rows_key)      # Hessian_z[v.f + L_k] - 2 J^T J, symbolic
        self.N, self.x0, self.P = N, np.asarray(x0, float), np.asarray(P, float)
        self.nx = self.sym["nx"]
        self.nu = len(self.sym["_sym"][1])
        self.nz = N * (self.nx + self.nu)

    # ---- unpack
    def xu(self, z):
        N, nx, nu = self.N, self.nx, self.nu
        X = np.vstack([self.x0[None], z[:N * nx].reshape(N, nx)])
        U = z[N * nx:].reshape(N, nu)
        return X, U

    def ix(self, k):          # columns of x_k in z (k >= 1)
        return slice((k - 1) * self.nx, k * self.nx)

    def iu(self, k):
        o = self.N * self.nx
        return slice(o + k * self.nu, o + (k + 1) * self.nu)

    def _res(self, k, x, u):
        """stacked residual of node k and its Jacobian wrt [x u] (terminal: state residuals, x only)"""
        s, p, nx = self.sym, self.P[k], self.nx
        if k == self.N:
            u0 = np.zeros(self.nu)
            return np.asarray(s["sres"](x, u0, p), float).reshape(-1), np.asarray(s["Js"](x, u0, p), float)[:, :nx]
        ri = np.asarray(s["ires"](x, u, p), float).reshape(-1)
        Ji = np.asarray(s["Ji"](x, u, p), float)
        if k == 0:
            return ri, Ji
        return (np.concatenate([ri, np.asarray(s["sres"](x, u, p), float).reshape(-1)]),
                np.vstack([Ji, np.asarray(s["Js"](x, u, p), float)]))

    # ---- objective
    def cost(self, z):
        X, U = self.xu(z)
        return sum(float(r @ r) for r in (self._res(k, X[k], U[k] if k < self.N else None)[0] for k in range(self.N + 1)))

    def grad(self, z):
        X, U = self.xu(z)
        g = np.zeros(self.nz)
        for k in range(self.N + 1):
            r, J = self._res(k, X[k], U[k] if k < self.N else None)
            gk = 2.0 * J.T @ r
            if k >= 1:
                g[self.ix(k)] += gk[:self.nx]
            if k < self.N:
                g[self.iu(k)] += gk[self.nx:]
        return g

    # ---- constraints c_{k+1} = f(x_k, u_k) - x_{k+1}, k = 0..N-1
    def cons(self, z):
        X, U = self.xu(z)
        return np.concatenate([np.asarray(self.sym["f"](X[k], U[k], self.P[k]), float).reshape(-1) - X[k + 1] for k in range(self.N)])

    def jac(self, z):
        X, U = self.xu(z)
        nx = self.nx
        A = sps.lil_matrix((self.N * nx, self.nz))
        for k in range(self.N):
            F = np.asarray(self.sym["F"](X[k], U[k], self.P[k]), float)
            rows = slice(k * nx, (k + 1) * nx)
            if k >= 1:
                A[rows, self.ix(k)] = F[:, :nx]
            A[rows, self.iu(k)] = F[:, nx:]
            A[rows, self.ix(k + 1)] = -np.eye(nx)
        return A.tocsr()

    def hess(self, z, lam):
        """exact Hessian of cost + lam . c"""
        X, U = self.xu(z)
        nx, nu = self.nx, self.nu
        H = sps.lil_matrix((self.nz, self.nz))
        for k in range(self.N + 1):
            r, J = self._res(k, X[k], U[k] if k < self.N else None)
            Hk = 2.0 * J.T @ J
            if k < self.N:
                Hk = Hk + np.asarray(self.so(X[k], U[k], self.P[k], lam[k * nx:(k + 1) * nx]), float)
                if k >= 1:
                    H[self.ix(k), self.ix(k)] += Hk[:nx, :nx]
                    H[self.ix(k), self.iu(k)] += Hk[:nx, nx:]
                    H[self.iu(k), self.ix(k)] += Hk[nx:, :nx]
                H[self.iu(k), self.iu(k)] += Hk[nx:, nx:]
            else:
                H[self.ix(k), self.ix(k)] += Hk
        return H.tocsr()

    def kkt(self, z, lam):
        A = self.jac(z)
        return self.grad(z) + A.T @ lam, self.cons(z), A


def solve(name, N, x0, P, xs0, us0, verbose=True, rel_vel=True, rows_key=None):
    T = Transcription(name, N, x0, P, rel_vel, rows_key)
    z0 = np.concatenate([np.asarray(xs0, float)[1:].reshape(-1), np.asarray(us0, float).reshape(-1)])
    t0 = time.time()
    con = sopt.NonlinearConstraint(T.cons, 0.0, 0.0, jac=T.jac, hess=lambda z, v: T.hess(z, v) - T.hess(z, 0 * v))
    res = sopt.minimize(T.cost, z0, jac=T.grad, hess=lambda z: T.hess(z, np.zeros(N * T.nx)), constraints=[con], method="trust-constr",
                        options=dict(gtol=1e-6, xtol=1e-12, maxiter=3000, initial_tr_radius=1.0, verbose=0))
    z = res.x
    lam = np.asarray(res.v[0], float)
    if verbose:
        print(f"  trust-constr: {res.nit} iterations, {time.time() - t0:.1f} s, cost {res.fun:.9e}, constraint violation {res.constr_violation:.2e}")
    # ---- Newton polish on the KKT system (exact Hessian): quadratic convergence to rounding level
    hist = []
    for it in range(12):
        gL, c, A = T.kkt(z, lam)
        # multipliers by least squares at the current point first (trust-constr's are approximate)
        if it == 0:
            lam = spla.lsqr(A.T.tocsr(), -T.grad(z), atol=1e-14, btol=1e-14, iter_lim=20000)[0]
            gL = T.grad(z) + A.T @ lam
        scale = max(1.0, np.max(np.abs(T.grad(z))))
        hist.append((float(np.max(np.abs(gL)) / scale), float(np.max(np.abs(c)))))
        if verbose:
            print(f"  newton {it}: stationarity {hist[-1][0]:.2e} (rel), feasibility {hist[-1][1]:.2e}, cost {T.cost(z):.12e}")
        if hist[-1][0] <= 1e-11 and hist[-1][1] <= 1e-12:
            break
        H = T.hess(z, lam)
        K = sps.bmat([[H, A.T], [A, None]], format="csc")
        d = spla.spsolve(K, -np.concatenate([gL, c]))
        z = z + d[:T.nz]
        lam = lam + d[T.nz:]
    gL, c, A = T.kkt(z, lam)
    # second-order sufficiency on the null space of A is not computed; the reduced cost at a perturbed feasible point is checked
    # by the tests instead (DDP lands on the same point from the same start)
    X, U = T.xu(z)
    return dict(x=X, u=U, cost=T.cost(z), kkt_stationarity_rel=float(np.max(np.abs(gL)) / max(1.0, np.max(np.abs(T.grad(z))))),
                kkt_feasibility=float(np.max(np.abs(c))), nit_trust_constr=int(res.nit), seconds=time.time() - t0)


def main(which=None):
    from srbd_horizon_amd import workload
    os.makedirs(OUT, exist_ok=True)
    jobs = [("srbd13", 30, [0, 1, 5, 12]),     # seeds 5, 12: commanded-velocity instances (rdot_ref at the last node != 0)
            ("srbd37", 20, [0, 3]),
            ("lip30", 20, [5]),
            ("srbd37", 60, [2]),               # BASELINE configs[4]: N = 60, every defect open at the start
            ("srbd61", 20, [1])]               # the code-default contact model (prb.py:39-41: 2 legs x 4 sole corners)
    for name, N, seeds in jobs:
        if which and name not in which and f"{name}_n{N}" not in which:
            continue
        if which and f"{name}_n{N}" not in which and any(w.startswith(name + "_n") for w in which):
            continue
        batch = workload.make_batch(name, N, seeds)
        for j, seed in enumerate(seeds):
            print(f"{name} N={N} seed {seed}: rdot_ref(N) = {batch['params'][j, N, 0:3]}")
            r = solve(name, N, batch["x0"][j], batch["params"][j], batch["xs"][j], batch["us"][j])
            path = os.path.join(OUT, f"nlp_{name}_n{N}_seed{seed}.npz")
            np.savez_compressed(path, model=name, N=N, seed=seed, x0=batch["x0"][j], params=batch["params"][j], xs0=batch["xs"][j],
                                us0=batch["us"][j],
                                consts_json=np.array(json.dumps({k: np.asarray(v, float).reshape(-1).tolist()
                                                                 for k, v in batch["consts"].items()}, sort_keys=True)), **r)
            print(f"  -> {path}: cost {r['cost']:.12e}, KKT stationarity {r['kkt_stationarity_rel']:.1e}, feasibility {r['kkt_feasibility']:.1e}")


def variant_srbd37_point_feet_with_user_rows():
    """srbd37 N = 20 seed 1 as number_of_legs = 4 x contact_model = 1 (no relative-velocity constraints, prb.py:166) with three
    user-declared linear rows (problem.LinearTerm): c0_xy tracking of a per-knot reference (two state rows) and a balance term on
    two vertical forces (a stage row).  -> tests/golden/nlp_srbd37x_n20_seed1.npz"""
    from srbd_horizon_amd import workload
    name, N, seed = "srbd37", 20, 1
    batch = workload.make_batch(name, N, [seed])
    nx, nu = 37, 24
    a0 = np.zeros(nx + nu); a0[7] = 1.0
    a1 = np.zeros(nx + nu); a1[8] = 1.0
    a2 = np.zeros(nx + nu); a2[nx + 5] = 1.0; a2[nx + 17] = -1.0
    rows = (dict(a=a0.tolist(), w=250.0, kind="state", const=0.0), dict(a=a1.tolist(), w=250.0, kind="state", const=0.0),
            dict(a=a2.tolist(), w=4.0, kind="stage", const=0.5))
    sym_models.EXTRA_ROWS["srbd37x"] = rows
    P = np.concatenate([batch["params"][0], np.zeros((N + 1, 8))], axis=1)
    foot = batch["x0"][0, 7:9]
    P[:, 19] = foot[0] + 0.02 * np.sin(np.arange(N + 1) / 4.0)          # references of the two tracking rows
    P[:, 20] = foot[1] - 0.01 * np.arange(N + 1) / N
    print(f"srbd37x N={N} seed {seed}: point feet + 3 user rows")
    r = solve(name, N, batch["x0"][0], P, batch["xs"][0], batch["us"][0], rel_vel=False, rows_key="srbd37x")
    consts = {k: np.asarray(v, float).reshape(-1).tolist() for k, v in batch["consts"].items()}
    consts["relative_velocity_constraints"] = [0.0]
    path = os.path.join(OUT, f"nlp_srbd37x_n{N}_seed{seed}.npz")
    np.savez_compressed(path, model=name, N=N, seed=seed, x0=batch["x0"][0], params=P, xs0=batch["xs"][0], us0=batch["us"][0],
                        consts_json=np.array(json.dumps(consts, sort_keys=True)), extra_rows_json=np.array(json.dumps(rows)), **r)
    print(f"  -> {path}: cost {r['cost']:.12e}, KKT stationarity {r['kkt_stationarity_rel']:.1e}, feasibility {r['kkt_feasibility']:.1e}")


if __name__ == "__main__":
    if sys.argv[1:] == ["srbd37x"]:
        variant_srbd37_point_feet_with_user_rows()
        sys.exit(0)
    main(sys.argv[1:] or None)
