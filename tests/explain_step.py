"""Diagnostic for ONE step-length decision on which the HIP engine and the oracle differ from the same iterate (the cases
tests/soak_parity.py reports as unexplained): is it the GAINS (an ill-conditioned Quu: the elimination order decides what the
near-null direction of the gains looks like) or the ROLLOUT?  Not collected by pytest; on the GPU box:

    python tests/explain_step.py <seed> <k> <alpha>      # srbd13, N = 30: the engine's iterate after k accepted steps

prints the rollout cost at `alpha` for the four combinations {engine gains, oracle gains} x {engine rollout, oracle rollout}, the
largest cond(Quu) of that sweep and how far the two sets of gains are apart."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ddp as oddp, models as omodels  # noqa: E402
from srbd_horizon_amd import workload  # noqa: E402
from srbd_horizon_amd.engine import DdpEngine  # noqa: E402

OPTS = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)


def main(seed, k, alpha, N=30):
    b = workload.make_srbd13_batch(N, [seed])
    cst = omodels.RobotConsts(**b["consts"])
    m = omodels.make_model("srbd13", cst)
    eng = DdpEngine("srbd13", N, 1, opts=dict(OPTS, max_iters=k), consts=b["consts"])
    eng.set_initial_state(b["x0"]); eng.set_x_warmstart(b["xs"]); eng.set_u_warmstart(b["us"])
    x, u = eng.solve(b["params"])
    st = eng.stats
    print(f"engine after {int(st['iters'][0])} steps: cost {st['cost'][0]:.12e} alpha {st['alpha'][0]} gap {st['gap'][0]:.3e} mu {st['mu'][0]} rho {st['rho'][0]:.3e}")
    xs, us, P, x0 = x[0], u[0], b["params"][0], b["x0"][0]
    mu = float(st["mu"][0])
    # the engine's gains and rollout from this iterate (theta = 0: the step before was not a full one on these crawls)
    eng.set_x_warmstart(x); eng.set_u_warmstart(u)
    kff_g, K_g, _ = eng.backward(b["params"], mu=mu)
    xg, ug, Jg = eng.forward(b["params"], alpha)
    d = oddp.defects(m, xs, us, P)
    ok, K_o, kff_o, *_ = oddp.backward_pass(m, xs, us, P, d, mu, 0.0, 1)
    J = oddp.total_cost(m, xs, us, P)
    _, _, J_oo = oddp.forward_pass(m, x0, xs, us, P, d, K_o, kff_o, alpha)
    _, _, J_go = oddp.forward_pass(m, x0, xs, us, P, d, K_g[0], kff_g[0], alpha)
    print(f"J = {J:.12e}; rollout cost at alpha = {alpha}:")
    print(f"  engine gains, engine rollout : {Jg[0]:.12e}   ({(Jg[0] - J) / J:+.3e} of J)")
    print(f"  engine gains, oracle rollout : {J_go:.12e}   ({(J_go - J) / J:+.3e})")
    print(f"  oracle gains, oracle rollout : {J_oo:.12e}   ({(J_oo - J) / J:+.3e})")
    # conditioning of the sweep and distance of the gains
    conds = []
    _, Vx, _, Vxx, _, _ = m.cost_derivs(xs[N], None, P[N], N)
    for kk in range(N - 1, -1, -1):
        fx, fu = m.f_jac(xs[kk], us[kk], P[kk])
        _, lx, lu, lxx, lux, luu = m.cost_derivs(xs[kk], us[kk], P[kk], kk)
        vp = Vx + Vxx @ d[kk]
        Quu = luu + fu.T @ Vxx @ fu + mu * np.eye(m.nu)
        Qux = lux + fu.T @ Vxx @ fx
        conds.append((np.linalg.cond(Quu), kk))
        Kk = -np.linalg.solve(Quu, Qux); kv = -np.linalg.solve(Quu, lu + fu.T @ vp)
        Vx = lx + fx.T @ vp + Qux.T @ kv
        Vxx = lxx + fx.T @ Vxx @ fx + Qux.T @ Kk
        Vxx = 0.5 * (Vxx + Vxx.T)
    # backward stability of a sweep: with V propagated by ITS OWN gains, every knot's gains must solve that knot's system
    # [Quu | Qu Qux] to a small relative RESIDUAL -- unless Quu is singular to working precision and the right-hand side has a
    # (rounding-sized) component along its null direction: then no solver can, and the oracle's Cholesky cannot either
    def worst_residual(Kall, kall):
        _, Vx, _, Vxx, _, _ = m.cost_derivs(xs[N], None, P[N], N)
        worst, at = 0.0, -1
        for kk in range(N - 1, -1, -1):
            fx, fu = m.f_jac(xs[kk], us[kk], P[kk])
            _, lx, lu, lxx, lux, luu = m.cost_derivs(xs[kk], us[kk], P[kk], kk)
            vp = Vx + Vxx @ d[kk]
            Quu = luu + fu.T @ Vxx @ fu + mu * np.eye(m.nu)
            Qux = lux + fu.T @ Vxx @ fx
            Qu = lu + fu.T @ vp
            Kk, kv = Kall[kk], kall[kk]
            nq = np.abs(Quu).max()
            r = max(np.abs(Quu @ Kk + Qux).max() / (nq * np.abs(Kk).max() + np.abs(Qux).max()),
                    np.abs(Quu @ kv + Qu).max() / (nq * np.abs(kv).max() + np.abs(Qu).max()))
            if kk == N - 1:
                first = r            # the first knot of the sweep: V' is the terminal cost's, the same input for both sweeps
            if r > worst:
                worst, at = r, kk
            Vx = lx + fx.T @ vp + Qux.T @ kv
            Vxx = lxx + fx.T @ Vxx @ fx + Qux.T @ Kk
            Vxx = 0.5 * (Vxx + Vxx.T)
        return worst, at, first
    for name, (Ka, ka) in (("engine", (K_g[0], kff_g[0])), ("oracle", (K_o, kff_o))):
        r, at, first = worst_residual(Ka, ka)
        print(f"{name} gains: relative residual of knot N - 1 (same inputs on both sides) {first:.2e}; largest along the recursion "
              f"re-done in numpy from these gains {r:.2e} (knot {at}; an ill-conditioned recursion does not reproduce across "
              "implementations, so this one is NOT a check)")
    c, kk = max(conds)
    dK = np.abs(K_g[0] - K_o).max(axis=(1, 2)) / np.maximum(np.abs(K_o).max(axis=(1, 2)), 1e-300)
    print(f"cond(Quu): max {c:.2e} at knot {kk}, median {np.median([v for v, _ in conds]):.2e}")
    print(f"gains engine vs oracle, relative l-inf per knot: max {dK.max():.2e} at knot {int(dK.argmax())}, median {np.median(dK):.2e}")


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]))
