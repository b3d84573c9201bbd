"""User-declared linear residual rows (include/sddp.h extra_*; reference: ddp.py:183-196 / :216-226 sum whatever residual the
function container holds): the "_x" builds of all four models against both oracles -- per-knot evaluation, one sweep,
converged solves from the C-ABI level, and through the builder surface with problem.LinearTerm."""
import numpy as np
import pytest

from oracle import cport, ddp as oddp, models as omodels
from srbd_horizon_amd import workload
from srbd_horizon_amd.ddp import DDPSolver
from srbd_horizon_amd.engine import DdpEngine, eval_knots
from srbd_horizon_amd.prb import SRBDProblem
from srbd_horizon_amd.problem import LinearTerm

pytestmark = pytest.mark.gpu
OPTS = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)


def _rows(model):
    nx, nu, _ = cport.DIMS[model]
    nz = nx + nu
    rng = np.random.default_rng(5)
    a0 = np.zeros(nz); a0[0] = 1.0                                  # r_x tracking (state row)
    a1 = np.zeros(nz); a1[1] = 1.0; a1[0] = -0.5                    # r_y - r_x / 2 (state row)
    a2 = np.zeros(nz); a2[nx + 2] = 1.0; a2[nx + (5 if model != "lip30" else 1)] = -1.0       # a difference of inputs (stage row)
    a3 = np.zeros(nz); a3[:nx] = 0.05 * rng.standard_normal(nx); a3[nx:] = 0.05 * rng.standard_normal(nu)   # dense (stage row)
    return (dict(a=a0, w=2e3, kind="state", const=0.0), dict(a=a1, w=5e2, kind="state", const=0.01),
            dict(a=a2, w=3.0, kind="stage", const=0.0), dict(a=a3, w=40.0, kind="stage", const=-0.2))


def _problem(model, N, seeds):
    batch = workload.make_batch(model, N, seeds)
    rows = _rows(model)
    npb = cport.DIMS[model][2]
    B = len(seeds)
    P = np.concatenate([batch["params"], np.zeros((B, N + 1, 8))], axis=2)
    P[:, :, npb + 0] = 0.02 * np.sin(np.arange(N + 1) / 5.0)[None]              # per-knot reference of row 0
    P[:, :, npb + 3] = 0.1 * np.cos(np.arange(N + 1) / 3.0)[None]               # ... and of row 3
    consts = dict(batch["consts"], extra_rows=rows)
    return batch, P, consts


@pytest.mark.parametrize("model", ["srbd13", "srbd37", "lip30", "srbd61"])
def test_knots_with_extra_rows(model):
    N = 20
    batch, P, consts = _problem(model, N, [0])
    m = omodels.make_model(model, omodels.RobotConsts(**consts))
    rng = np.random.default_rng(2)
    ks = np.array([0, 1, 7, N - 1, N], dtype=np.int32)
    X = np.stack([m.initial_state() + 0.05 * rng.standard_normal(m.nx) for _ in ks])
    U = np.stack([m.static_input() + 0.05 * rng.standard_normal(m.nu) for _ in ks])
    Pk = np.stack([P[0, k] + 0.01 * rng.standard_normal(m.np_) for k in ks])
    f, F, H, g, L = eval_knots(model, N, ks, X, U, Pk, consts=consts)
    for i, k in enumerate(ks):
        term = k == N
        Lo, lx, lu, lxx, lux, luu = m.cost_derivs(X[i], None if term else U[i], Pk[i], int(k))
        assert abs(L[i] - Lo) <= 1e-11 * max(1.0, abs(Lo))
        if term:
            np.testing.assert_allclose(g[i, :m.nx], lx, rtol=1e-11, atol=1e-9 * max(1, np.max(np.abs(lx))))
            np.testing.assert_allclose(H[i, :m.nx, :m.nx], lxx, rtol=1e-11, atol=1e-7)
        else:
            np.testing.assert_allclose(f[i], m.f(X[i], U[i], Pk[i]), rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(g[i], np.concatenate([lx, lu]), rtol=1e-11, atol=1e-9 * max(1, np.max(np.abs(lx))))
            np.testing.assert_allclose(H[i], np.block([[lxx, lux.T], [lux, luu]]), rtol=1e-11, atol=1e-7)


@pytest.mark.parametrize("model,N,B,wps", [("srbd13", 30, 48, 1), ("srbd13", 30, 48, 2), ("srbd37", 20, 16, 2), ("srbd37", 60, 4, 1), ("lip30", 20, 12, 1), ("srbd61", 20, 6, 1)])
def test_solves_with_extra_rows_match_the_c_oracle(model, N, B, wps):
    batch, P, consts = _problem(model, N, np.arange(B) + 1)
    eng = DdpEngine(model, N, B, opts=dict(OPTS, waves_per_simd=wps), consts=consts)
    assert eng.np_ == cport.DIMS[model][2] + 8
    eng.set_initial_state(batch["x0"]); eng.set_x_warmstart(batch["xs"]); eng.set_u_warmstart(batch["us"])
    x, u = eng.solve(P)
    xo, uo, so = cport.solve_batch(omodels.RobotConsts(**consts), oddp.DdpOptions(**OPTS), batch["x0"], P, batch["xs"], batch["us"],
                                   threads=4, model=model)
    np.testing.assert_array_equal(eng.stats["iters"], so[:, 1].astype(int))
    assert eng.stats["converged"].all()
    assert np.max(np.abs(x - xo)) <= 1e-6 and np.max(np.abs(u - uo)) <= 1e-6
    np.testing.assert_allclose(eng.stats["cost"], so[:, 0], rtol=1e-9)
    # the rows act: the solution differs from the plain model's
    e0 = DdpEngine(model, N, B, opts=OPTS, consts=batch["consts"])
    e0.set_initial_state(batch["x0"]); e0.set_x_warmstart(batch["xs"]); e0.set_u_warmstart(batch["us"])
    x0, u0 = e0.solve(batch["params"])
    assert np.max(np.abs(x - x0)) > 1e-4


def test_one_sweep_and_one_pass_with_extra_rows():
    """gains of one backward sweep and one forward pass of the srbd13 "_x" build against the numpy oracle"""
    model, N = "srbd13", 30
    batch, P, consts = _problem(model, N, [4])
    m = omodels.make_model(model, omodels.RobotConsts(**consts))
    eng = DdpEngine(model, N, 1, opts=OPTS, consts=consts)
    eng.set_initial_state(batch["x0"]); eng.set_x_warmstart(batch["xs"]); eng.set_u_warmstart(batch["us"])
    kff, K, scal = eng.backward(P, mu=0.0)
    xs = batch["xs"][0].copy(); xs[0] = batch["x0"][0]
    d = oddp.defects(m, xs, batch["us"][0], P[0])
    ok, Ko, ko = oddp.backward_pass(m, xs, batch["us"][0], P[0], d, 0.0)[:3]
    assert ok
    np.testing.assert_allclose(kff[0], ko, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(K[0], Ko, rtol=1e-7, atol=1e-8)
    xg, ug, Jg = eng.forward(P, 0.125)                      # (0.5 blows up from this cold start, in the oracle too)
    xo, uo, Jo = oddp.forward_pass(m, batch["x0"][0], xs, batch["us"][0], P[0], d, Ko, ko, 0.125)
    assert np.max(np.abs(xg[0] - xo)) <= 1e-9 and abs(Jg[0] - Jo) <= 1e-9 * abs(Jo)


@pytest.mark.parametrize("contact_model", [2, 4])
def test_builder_surface_with_a_user_tracking_term(contact_model):
    """prb.createResidual("c0_xy_tracking", ...) on the reference's problem: solves instead of raising (VERDICT r04 missing #4);
    contact_model 2 = the launch file's (srbd37), 4 = the default in prb.py:39 (srbd61)."""
    ns, T = 20, 1.0
    model, nx, npar = ("srbd37", 37, 19) if contact_model == 2 else ("srbd61", 61, 27)
    pb = SRBDProblem(); prb = pb.createSRBDProblem(ns, T, params=dict(contact_model=contact_model))
    ref = prb.createParameter("c0_xy_ref", 2)
    tgt = pb.initial_foot_position[0][0:2] + np.array([0.03, -0.02])
    ref.assign(tgt)
    prb.createResidual("c0_xy_tracking", LinearTerm({pb.c[0]: [[1, 0, 0], [0, 1, 0]]}, gain=1e5, ref=ref), nodes=range(1, ns + 1))
    # the foot must be free to move: swing phase for every contact point of that foot (cdot_switch = 0 releases cdotxy_tracking)
    for i in range(contact_model):
        pb.cdot_switch[i].assign(0.0)
    solver = DDPSolver(prb, OPTS)
    x0 = pb.getInitialState()
    solver.setInitialState(x0)
    solver.set_u_warmstart(np.repeat(pb.getStaticInput()[:, None], ns, axis=1))
    assert solver.solve()
    sol = solver.getSolutionDict()
    P = solver._parameter_matrix()
    consts = solver.ddp_solver.consts
    assert consts.n_extra == 2 and P.shape == (ns + 1, npar + 8)
    rows = tuple(dict(a=np.array(consts.extra_a[128 * j:128 * j + 128][:nx + (24 if contact_model == 2 else 48)]), w=consts.extra_weight[j], kind="state", const=0.0)
                 for j in range(2))
    m = omodels.make_model(model, omodels.RobotConsts(extra_rows=rows))
    r = oddp.solve(m, x0, P, np.repeat(x0[None], ns + 1, axis=0), np.repeat(pb.getStaticInput()[None], ns, axis=0), oddp.DdpOptions(**OPTS))
    assert r.converged and solver.stats["iters"] == r.iters
    assert np.max(np.abs(sol["x_opt"].T - r.xs)) <= 1e-6 and np.max(np.abs(sol["u_opt"].T - r.us)) <= 1e-6
    moved = sol["c0"][0:2, -1] - pb.initial_foot_position[0][0:2]
    assert np.linalg.norm(moved - np.array([0.03, -0.02])) < 0.01            # the term pulls the foot to its target
