"""The reference's SRBD problem at the contact configuration its code defaults to -- contact_model = 4, number_of_legs = 2
(prb.py:39-41): nc = 8 contact points, nx = 61, nu = 48, np = 27 ("srbd61") -- HIP engine (through the C ABI) vs the oracles.
Same tolerances as tests/test_gpu_parity.py.  The kernel is the 4-wavefront one in its W-free layout (three 3x3 blocks of Q and two
2x2 blocks of Vxx per thread, two augmented columns per lane in the Gauss-Jordan, 12 rows per wavefront)."""
import numpy as np
import pytest

from oracle import cport
from oracle import ddp as oddp
from oracle import models as omodels
from srbd_horizon_amd import workload
from srbd_horizon_amd.engine import DdpEngine, eval_knots

pytestmark = pytest.mark.gpu
NAME = "srbd61"


def _opts(**over):
    o = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)      # dsrbd_example.py:55-58
    o.update(over)
    return o


def test_default_constants_are_the_eight_point_robot():
    """sddp_create(consts = NULL) / DdpEngine("srbd61") without constants must solve the problem its own workload and oracle
    describe: contact points 0..3 = the first four sole corners of the eight-point model (sddp_default_consts_for, ABI v9), not the
    line feet of srbd37 (the rel_pos targets d_initial_1/2 silently differed)."""
    N, seeds = 20, np.arange(6)
    batch = workload.make_batch(NAME, N, seeds)
    eng = DdpEngine(NAME, N, len(seeds), opts=_opts())                       # no consts
    eng.set_initial_state(batch["x0"]); eng.set_x_warmstart(batch["xs"]); eng.set_u_warmstart(batch["us"])
    x, u = eng.solve(batch["params"])
    xo, uo, so = cport.solve_batch(omodels.make_model(NAME).cst, oddp.DdpOptions(**_opts()), batch["x0"], batch["params"], batch["xs"],
                                   batch["us"], threads=4, model=NAME)
    np.testing.assert_array_equal(eng.stats["iters"], so[:, 1].astype(int))
    assert eng.stats["converged"].all()
    assert np.max(np.abs(x - xo)) <= 1e-6 and np.max(np.abs(u - uo)) <= 1e-6
    np.testing.assert_allclose(eng.stats["cost"], so[:, 0], rtol=1e-9)
    ex = DdpEngine(NAME, N, len(seeds), opts=_opts(), consts=batch["consts"])      # the builder's explicit constants: the same problem
    ex.set_initial_state(batch["x0"]); ex.set_x_warmstart(batch["xs"]); ex.set_u_warmstart(batch["us"])
    x2, u2 = ex.solve(batch["params"])
    np.testing.assert_array_equal(x2, x); np.testing.assert_array_equal(u2, u)


@pytest.mark.parametrize("imode,lever", [(0, 1.0), (1, -1.0)])
def test_eval_knots_matches_oracle(imode, lever):
    N = 20
    m = omodels.make_model(NAME, omodels.RobotConsts(inertia_mode=imode, lever_sign=lever))
    rng = np.random.default_rng(5)
    ks = np.array([0, 1, 7, N - 1, N, 3, N, 0], dtype=np.int32)
    nk = len(ks)
    X = np.tile(m.initial_state(), (nk, 1)) + 0.05 * rng.standard_normal((nk, m.nx))
    U = np.tile(m.static_input(), (nk, 1)) + 0.05 * rng.standard_normal((nk, m.nu))
    P = np.tile(m.default_params(N)[3], (nk, 1)) + 0.05 * rng.standard_normal((nk, m.np_))
    f, F, H, g, L = eval_knots(NAME, N, ks, X, U, P, consts=dict(inertia_mode=imode, lever_sign=lever, feet=m.cst.feet8[:4]))
    for t, k in enumerate(ks):
        if k < N:
            np.testing.assert_allclose(f[t], m.f(X[t], U[t], P[t]), rtol=1e-12, atol=1e-13)
            fx, fu = m.f_jac(X[t], U[t], P[t])
            np.testing.assert_allclose(F[t], np.hstack([fx, fu]), rtol=1e-11, atol=1e-12)
            Lo, lx, lu, lxx, lux, luu = m.cost_derivs(X[t], U[t], P[t], int(k))
            Ho = np.block([[lxx, lux.T], [lux, luu]])
            go = np.concatenate([lx, lu])
        else:
            Lo, lx, _, lxx, _, _ = m.cost_derivs(X[t], None, P[t], int(k))
            Ho = np.zeros((m.nx + m.nu,) * 2); Ho[:m.nx, :m.nx] = lxx
            go = np.concatenate([lx, np.zeros(m.nu)])
        assert abs(L[t] - Lo) <= 1e-12 * max(1.0, abs(Lo))
        np.testing.assert_allclose(g[t], go, rtol=1e-11, atol=1e-11 * max(1.0, np.max(np.abs(go))))
        np.testing.assert_allclose(H[t], Ho, rtol=1e-11, atol=1e-11 * max(1.0, np.max(np.abs(Ho))))


def test_backward_and_forward_pass_match_oracle():
    N, seeds = 20, [0, 5, 13]
    batch = workload.make_batch(NAME, N, seeds)
    m = omodels.make_model(NAME)
    rng = np.random.default_rng(1)
    xs = batch["xs"] + 1e-3 * rng.standard_normal(batch["xs"].shape)       # open gaps: multiple shooting
    us = batch["us"] + 1e-3 * rng.standard_normal(batch["us"].shape)
    xs[:, 0] = batch["x0"]
    eng = DdpEngine(NAME, N, len(seeds), opts=_opts(), consts=batch["consts"])
    eng.set_initial_state(batch["x0"]); eng.set_x_warmstart(xs); eng.set_u_warmstart(us)
    kff, K, scal = eng.backward(batch["params"], mu=0.0)
    xg, ug, Jg = eng.forward(batch["params"], 0.25)
    for b in range(len(seeds)):
        P = batch["params"][b]
        d = oddp.defects(m, xs[b], us[b], P)
        ok, Ko, ko, dV1, dV2, G1, G2, Vx0, Vxx0, qu = oddp.backward_pass(m, xs[b], us[b], P, d, 0.0)
        assert ok and scal[b, 4] == 1.0
        np.testing.assert_allclose(K[b], Ko, rtol=1e-7, atol=1e-8 * max(1.0, np.max(np.abs(Ko))))
        np.testing.assert_allclose(kff[b], ko, rtol=1e-7, atol=1e-8 * max(1.0, np.max(np.abs(ko))))
        for got, ref in ((scal[b, 0], dV1), (scal[b, 1], dV2), (scal[b, 2], G1), (scal[b, 3], G2)):
            assert abs(got - ref) <= 1e-8 * max(1.0, abs(ref), abs(dV1))
        assert abs(scal[b, 7] - oddp.total_cost(m, xs[b], us[b], P)) <= 1e-11 * abs(scal[b, 7])
        xo, uo, Jo = oddp.forward_pass(m, batch["x0"][b], xs[b], us[b], P, d, Ko, ko, 0.25)
        np.testing.assert_allclose(xg[b], xo, rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(ug[b], uo, rtol=1e-8, atol=1e-8)
        assert abs(Jg[b] - Jo) <= 1e-9 * abs(Jo)


def test_converged_solve_matches_the_numpy_oracle():
    N, seeds = 20, [0, 1, 6, 16]
    batch = workload.make_batch(NAME, N, seeds)
    m = omodels.make_model(NAME)
    eng = DdpEngine(NAME, N, len(seeds), opts=_opts(), consts=batch["consts"])
    eng.set_initial_state(batch["x0"]); eng.set_x_warmstart(batch["xs"]); eng.set_u_warmstart(batch["us"])
    x, u = eng.solve(batch["params"])
    st = eng.stats
    for b in range(len(seeds)):
        r = oddp.solve(m, batch["x0"][b], batch["params"][b], batch["xs"][b], batch["us"][b], oddp.DdpOptions(**_opts()))
        assert r.converged and st["converged"][b] and st["status"][b] == 0
        assert st["iters"][b] == r.iters, (st["iters"][b], r.iters)
        assert np.max(np.abs(x[b] - r.xs)) <= 1e-6 and np.max(np.abs(u[b] - r.us)) <= 1e-6
        assert abs(st["cost"][b] - r.cost) <= 1e-9 * abs(r.cost)
        assert st["gap"][b] <= 1e-9


@pytest.mark.parametrize("N,nseeds", [(60, 8), (20, 40)])
def test_solved_to_convergence_matches_the_c_oracle(N, nseeds):
    """whole solves against the plain-C oracle: N = 60 with every node's defect open (configs[4]'s shape on this model), and
    N = 20 (the horizon dsrbd_example.py runs) over 40 seeds."""
    seeds = np.arange(nseeds) + 20
    batch = workload.make_batch(NAME, N, seeds)
    rng = np.random.default_rng(9)
    xs = batch["xs"] + (1e-3 * rng.standard_normal(batch["xs"].shape) if N == 60 else 0.0)
    xs[:, 0] = batch["x0"]
    eng = DdpEngine(NAME, N, len(seeds), opts=_opts(), consts=batch["consts"])
    eng.set_initial_state(batch["x0"]); eng.set_x_warmstart(xs); eng.set_u_warmstart(batch["us"])
    x, u = eng.solve(batch["params"])
    st = eng.stats.copy()
    xo, uo, so = cport.solve_batch(omodels.make_model(NAME).cst, oddp.DdpOptions(**_opts()), batch["x0"], batch["params"], xs, batch["us"],
                                   threads=4, model=NAME)
    print(f"srbd61 N={N}: iterations GPU", st["iters"].tolist(), "oracle", so[:, 1].astype(int).tolist(),
          "linf x", float(np.max(np.abs(x - xo))), "linf u", float(np.max(np.abs(u - uo))))
    np.testing.assert_array_equal(st["iters"], so[:, 1].astype(int))
    np.testing.assert_array_equal(st["status"], so[:, 6].astype(int))
    np.testing.assert_array_equal(st["converged"], so[:, 2].astype(int))
    assert st["converged"].all() and np.all(st["gap"] <= 1e-9) and st["iters"].min() >= 3
    assert np.max(np.abs(x - xo)) <= 1e-6 and np.max(np.abs(u - uo)) <= 1e-6
    np.testing.assert_allclose(st["cost"], so[:, 0], rtol=1e-9)


def test_regularisation_bump_on_indefinite_quu():
    """mu0 < 0 large makes Quu indefinite: the engine must bump mu (ddp.py:34-35) exactly like the oracle; a failed sweep leaves
    tiles that alias each other half-written (W-free layout: the gains over GC | WC), the retry must not see them"""
    N, seeds = 20, [3]
    batch = workload.make_batch(NAME, N, seeds)
    m = omodels.make_model(NAME)
    over = dict(mu0=-1e9, max_iters=3)
    eng = DdpEngine(NAME, N, 1, opts=_opts(**over), consts=batch["consts"])
    eng.set_initial_state(batch["x0"]); eng.set_x_warmstart(batch["xs"]); eng.set_u_warmstart(batch["us"])
    x, u = eng.solve(batch["params"])
    r = oddp.solve(m, batch["x0"][0], batch["params"][0], batch["xs"][0], batch["us"][0], oddp.DdpOptions(**_opts(**over)))
    assert eng.stats["mu"][0] == pytest.approx(r.mu, rel=1e-12)
    assert eng.stats["iters"][0] == r.iters
    assert np.max(np.abs(x[0] - r.xs)) <= 1e-7


@pytest.mark.parametrize("N,B", [(1, 2), (2, 3), (70, 1)])
def test_extreme_horizons(N, B):
    seeds = list(range(B))
    batch = workload.make_batch(NAME, N, seeds)
    eng = DdpEngine(NAME, N, B, opts=_opts(), consts=batch["consts"])
    eng.set_initial_state(batch["x0"]); eng.set_x_warmstart(batch["xs"]); eng.set_u_warmstart(batch["us"])
    x, u = eng.solve(batch["params"])
    xo, uo, so = cport.solve_batch(omodels.make_model(NAME).cst, oddp.DdpOptions(**_opts()), batch["x0"], batch["params"], batch["xs"],
                                   batch["us"], model=NAME)
    np.testing.assert_array_equal(eng.stats["iters"], so[:, 1].astype(int))
    assert np.max(np.abs(x - xo)) <= 1e-6 and np.max(np.abs(u - uo)) <= 1e-6


def test_queue_of_more_instances_than_slots_is_bit_identical():
    """20 instances through 3 queue slots, in cost order, against one workgroup per instance"""
    N, seeds = 20, np.arange(20)
    batch = workload.make_batch(NAME, N, seeds)
    res = []
    for over in (dict(), dict(max_slots=3, queue_order=2)):
        eng = DdpEngine(NAME, N, len(seeds), opts=_opts(**over), consts=batch["consts"])
        eng.set_initial_state(batch["x0"]); eng.set_x_warmstart(batch["xs"]); eng.set_u_warmstart(batch["us"])
        x, u = eng.solve(batch["params"])
        res.append((x.copy(), u.copy(), eng.stats.copy()))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][2]["iters"], res[1][2]["iters"])
    np.testing.assert_array_equal(res[0][2]["cost"], res[1][2]["cost"])


def test_reference_surface_at_the_default_contact_model():
    """SRBDProblem with the defaults of prb.py:39-40 through DDPSolver (ddp.py:10-230), then a short closed loop"""
    from srbd_horizon_amd.mpc import MpcLoop
    loop = MpcLoop(NAME, ns=20)
    assert loop.srbd.nc == 8 and loop.srbd.prb.getState().getVars()[-1].getName() == "cdot7"
    assert loop.solver.state_size == 61 and loop.solver.input_size == 48
    for t in range(6):
        ok, sol = loop.tick("walking", axes=(1.0, 0.0))
        assert ok
    assert sol["x_opt"].shape == (61, 21) and sol["u_opt"].shape == (48, 20) and sol["c7"].shape == (3, 21) and sol["f7"].shape == (3, 20)
    assert loop.state.shape == (61,) and abs(np.linalg.norm(loop.state[3:7]) - 1.0) < 1e-12
