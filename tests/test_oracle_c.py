"""The plain-C oracle (CPU baseline / large-batch checker) vs the numpy oracle: per-knot evaluation and whole solves."""
import numpy as np
import pytest

from oracle import cport, ddp as oddp, models as omodels
from srbd_horizon_amd import workload


@pytest.mark.parametrize("imode,lever,barrier", [(0, 1.0, 0.0), (1, -1.0, 0.0), (0, 1.0, 6.0)])
def test_c_knot_evaluation_matches_numpy(imode, lever, barrier):
    cst = omodels.RobotConsts(inertia_mode=imode, lever_sign=lever, friction_barrier_weight=barrier, friction_barrier_sharpness=4.0)
    m = omodels.make_model("srbd13", cst)
    rng = np.random.default_rng(11)
    for k, term in ((0, False), (4, False), (20, True)):
        x = m.initial_state() + 0.05 * rng.standard_normal(13)
        u = m.static_input() + 0.05 * rng.standard_normal(6)
        p = m.default_params(20)[3] + 0.05 * rng.standard_normal(19)
        f, F, H, g, L = cport.eval_knot(cst, x, u, p, k, term)
        Lo, lx, lu, lxx, lux, luu = m.cost_derivs(x, None if term else u, p, k)
        assert abs(L - Lo) <= 1e-12 * max(1, abs(Lo))
        if term:
            np.testing.assert_allclose(g[:13], lx, rtol=1e-11, atol=1e-9)
            np.testing.assert_allclose(H[:13, :13], lxx, rtol=1e-11, atol=1e-9)
        else:
            np.testing.assert_allclose(f, m.f(x, u, p), rtol=1e-12, atol=1e-13)
            fx, fu = m.f_jac(x, u, p)
            np.testing.assert_allclose(F, np.hstack([fx, fu]), rtol=1e-11, atol=1e-12)
            np.testing.assert_allclose(g, np.concatenate([lx, lu]), rtol=1e-11, atol=1e-9 * max(1, np.max(np.abs(lx))))
            np.testing.assert_allclose(H, np.block([[lxx, lux.T], [lux, luu]]), rtol=1e-11, atol=1e-7)


@pytest.mark.parametrize("barrier", [0.0, 6.0])
def test_c_solve_matches_numpy_solve(barrier):
    N, seeds = 30, [0, 1, 3, 6]
    batch = workload.make_batch("srbd13", N, seeds)
    cst = omodels.RobotConsts(friction_barrier_weight=barrier, friction_barrier_sharpness=4.0)
    m = omodels.make_model("srbd13", cst)
    opts = oddp.DdpOptions(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)
    xs, us, st = cport.solve_batch(cst, opts, batch["x0"], batch["params"], batch["xs"], batch["us"], threads=2)
    for b in range(len(seeds)):
        r = oddp.solve(m, batch["x0"][b], batch["params"][b], batch["xs"][b], batch["us"][b], opts)
        assert int(st[b, 1]) == r.iters and bool(st[b, 2]) == r.converged and st[b, 3] == r.alpha
        assert np.max(np.abs(xs[b] - r.xs)) <= 1e-8 and np.max(np.abs(us[b] - r.us)) <= 1e-8
        assert abs(st[b, 0] - r.cost) <= 1e-10 * abs(r.cost)


def test_exhausted_line_search_is_a_stall_not_convergence_in_both_oracles():
    """alpha below alpha_converge_threshold stops the solve; with open multiple-shooting gaps that is status 4
    and NOT converged.  With closed gaps and no predicted decrease left it still counts as converged."""
    N = 30
    batch = workload.make_batch("srbd13", N, [7])
    cst = omodels.RobotConsts()
    m = omodels.make_model("srbd13", cst)
    opts = oddp.DdpOptions(max_iters=5, alpha_converge_threshold=0.5, beta=1e-3)       # seed 7 needs alpha = 0.125 in iteration 1
    r = oddp.solve(m, batch["x0"][0], batch["params"][0], batch["xs"][0], batch["us"][0], opts)
    _, _, st = cport.solve_batch(cst, opts, batch["x0"], batch["params"], batch["xs"], batch["us"])
    assert r.status == 4 and not r.converged and r.iters == 0 and r.gap > opts.gap_tol
    assert int(st[0, 6]) == 4 and int(st[0, 2]) == 0 and int(st[0, 1]) == 0
    # restart from the optimum with an impossible Armijo fraction: nothing is accepted, but the point is optimal
    full = oddp.DdpOptions(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)
    opt = oddp.solve(m, batch["x0"][0], batch["params"][0], batch["xs"][0], batch["us"][0], full)
    assert opt.converged and opt.status == 0
    hard = oddp.DdpOptions(max_iters=5, alpha_converge_threshold=0.5, beta=1e6, cost_reduction_ths=1e-12)
    r2 = oddp.solve(m, batch["x0"][0], batch["params"][0], opt.xs, opt.us, hard)
    _, _, st2 = cport.solve_batch(cst, hard, batch["x0"], batch["params"], opt.xs[None], opt.us[None])
    assert r2.status == 0 and r2.converged and r2.iters == 0 and int(st2[0, 2]) == 1 and int(st2[0, 6]) == 0      # converged <=> status 0
    # the same restart with a threshold no sweep can meet: the line search is exhausted and the point does NOT pass the
    # (relative) optimality test any more -> status 4, not converged
    harder = oddp.DdpOptions(max_iters=5, alpha_converge_threshold=0.5, beta=1e6, cost_reduction_ths=1e-30)
    r3 = oddp.solve(m, batch["x0"][0], batch["params"][0], opt.xs, opt.us, harder)
    _, _, st3 = cport.solve_batch(cst, harder, batch["x0"], batch["params"], opt.xs[None], opt.us[None])
    assert r3.status == 4 and not r3.converged and int(st3[0, 6]) == 4 and int(st3[0, 2]) == 0


@pytest.mark.parametrize("name,barrier", [("srbd37", 0.0), ("srbd37", 6.0), ("lip30", 0.0), ("srbd61", 0.0), ("srbd61", 6.0)])
def test_c_knot_evaluation_matches_numpy_reference_models(name, barrier):
    cst = omodels.RobotConsts(friction_barrier_weight=barrier, friction_barrier_sharpness=4.0)
    m = omodels.make_model(name, cst)
    rng = np.random.default_rng(5)
    for k, term in ((0, False), (3, False), (20, True)):
        x = m.initial_state() + 0.05 * rng.standard_normal(m.nx)
        u = m.static_input() + 0.05 * rng.standard_normal(m.nu)
        p = m.default_params(20)[3] + 0.05 * rng.standard_normal(m.np_)
        f, F, H, g, L = cport.eval_knot(cst, x, u, p, k, term, model=name)
        Lo, lx, lu, lxx, lux, luu = m.cost_derivs(x, None if term else u, p, k)
        assert abs(L - Lo) <= 1e-12 * max(1, abs(Lo))
        nx = m.nx
        if term:
            np.testing.assert_allclose(g[:nx], lx, rtol=1e-11, atol=1e-9 * max(1, np.max(np.abs(lx))))
            np.testing.assert_allclose(H[:nx, :nx], lxx, rtol=1e-11, atol=1e-7)
        else:
            np.testing.assert_allclose(f, m.f(x, u, p), rtol=1e-12, atol=1e-13)
            fx, fu = m.f_jac(x, u, p)
            np.testing.assert_allclose(F, np.hstack([fx, fu]), rtol=1e-11, atol=1e-12)
            np.testing.assert_allclose(g, np.concatenate([lx, lu]), rtol=1e-11, atol=1e-9 * max(1, np.max(np.abs(g))))
            np.testing.assert_allclose(H, np.block([[lxx, lux.T], [lux, luu]]), rtol=1e-11, atol=1e-9 * np.max(np.abs(H)))


@pytest.mark.parametrize("name,N,seeds", [("srbd37", 20, [0, 3]), ("lip30", 20, [1, 2]), ("srbd37", 8, [5]), ("srbd61", 20, [0, 2]), ("srbd61", 6, [7])])
def test_c_solve_matches_numpy_solve_reference_models(name, N, seeds):
    batch = workload.make_batch(name, N, seeds)
    cst = omodels.RobotConsts()
    m = omodels.make_model(name, cst)
    opts = oddp.DdpOptions(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)
    xs, us, st = cport.solve_batch(cst, opts, batch["x0"], batch["params"], batch["xs"], batch["us"], threads=2, model=name)
    for b in range(len(seeds)):
        r = oddp.solve(m, batch["x0"][b], batch["params"][b], batch["xs"][b], batch["us"][b], opts)
        assert int(st[b, 1]) == r.iters and bool(st[b, 2]) == r.converged and st[b, 3] == r.alpha and int(st[b, 6]) == r.status
        assert np.max(np.abs(xs[b] - r.xs)) <= 1e-7 and np.max(np.abs(us[b] - r.us)) <= 1e-7
        assert abs(st[b, 0] - r.cost) <= 1e-9 * abs(r.cost)


@pytest.mark.parametrize("name,N,seeds,bar", [("srbd13", 30, [1, 2, 4], 0.0), ("srbd37", 12, [3], 0.0), ("srbd13", 30, [2], 2.0), ("srbd61", 10, [4], 0.0)])
def test_c_full_second_order_solve_matches_numpy(name, N, seeds, bar):
    """second_order = 2 (v'.f_zz + exact cost Hessian after full steps): the C port takes the same path as the numpy oracle."""
    batch = workload.make_batch(name, N, seeds)
    cst = omodels.RobotConsts(friction_barrier_weight=bar, friction_barrier_sharpness=4.0)
    m = omodels.make_model(name, cst)
    opts = oddp.DdpOptions(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3, second_order=2)
    xs, us, st = cport.solve_batch(cst, opts, batch["x0"], batch["params"], batch["xs"], batch["us"], threads=2, model=name)
    o1 = oddp.DdpOptions(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3, second_order=1)
    _, _, st1 = cport.solve_batch(cst, o1, batch["x0"], batch["params"], batch["xs"], batch["us"], threads=2, model=name)
    for b in range(len(seeds)):
        r = oddp.solve(m, batch["x0"][b], batch["params"][b], batch["xs"][b], batch["us"][b], opts)
        assert int(st[b, 1]) == r.iters and bool(st[b, 2]) == r.converged and st[b, 3] == r.alpha
        assert np.max(np.abs(xs[b] - r.xs)) <= 1e-7 and np.max(np.abs(us[b] - r.us)) <= 1e-7
        assert abs(st[b, 0] - r.cost) <= 1e-9 * abs(r.cost)
        assert abs(st[b, 0] - st1[b, 0]) <= 1e-6 * abs(st1[b, 0])          # the same optimum as the default mode


def test_two_cpu_builds_of_the_oracle_shadow_each_other_step_by_step():
    """oracle/Makefile builds the same C sources twice: -ffp-contract=off (the oracle) and =fast (gcc fuses multiply-adds, as hipcc
    does on the device).  On the hard instances of the workload (seeds found by scanning configs[3]'s 8192: long crawls through
    an ill-conditioned region) the two builds drift apart with identical step lengths and end at different iteration counts --
    the restatement scatters against itself under a legal change of rounding, by about as much as the GPU scatters against it
    (tests/test_gpu_divergence.py reports both counts).  What does hold, and is the parity statement the GPU tests make for such
    instances: every accepted step of one build is the other build's step from the same iterate (tests/shadow.py)."""
    from tests import shadow
    N, seeds = 30, [1897, 5393, 59571]       # 59571: a step on which the two builds pick 2^-10 and 2^-1 from the same iterate
    opts = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)
    batch = workload.make_batch("srbd13", N, seeds)
    cst = omodels.RobotConsts(**batch["consts"])
    o = oddp.DdpOptions(**opts)
    for b in range(len(seeds)):
        a = (batch["x0"][b], batch["params"][b], batch["xs"][b], batch["us"][b])
        states = shadow.engine_states_from_oracle(cst, opts, *a, variant="fast")           # the "engine under test": build 2
        steps = shadow.shadow_one_instance(cst, opts, a[0], a[1], states)                  # checked by build 1, one step at a time
        s = shadow.summarize(steps)
        assert s["steps"] == states[-1]["iters"] >= 30                                     # a long crawl, every step checked
        shadow.assert_shadowed(dict(shadow=s, gpu_iters=states[-1]["iters"]))
        # end to end the two builds part ways (same steps until the drift is visible): reported, not asserted -- which
        # multiply-adds gcc fuses depends on the host CPU the library is built for
        _, _, _, t0 = cport.solve_trace(cst, o, *a)
        _, _, _, t1 = cport.solve_trace(cst, o, *a, variant="fast")
        print(f"seed {seeds[b]}: iterations off {sum(r['alpha'] > 0 for r in t0)} / fast {sum(r['alpha'] > 0 for r in t1)}, first split "
              f"{shadow.first_split(t1, t0)}; worst stable step {s['max_rel_cost']:.1e} = {s['max_ratio']:.1f} x the oracle's own sensitivity; "
              f"{s['unstable_steps']} unstable steps")


def test_resumed_oracle_solve_continues_the_same_path():
    """The carried state (rho, mu, theta, closed gaps) is all an iteration needs: a solve cut at k and continued lands where the
    uncut solve lands (to rounding: the continued solve recomputes defects and cost from the stored iterate)."""
    N = 30
    opts = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)
    batch = workload.make_batch("srbd13", N, [3])
    cst = omodels.RobotConsts(**batch["consts"])
    a = (batch["x0"], batch["params"], batch["xs"], batch["us"])
    xf, uf, sf = cport.solve_batch(cst, oddp.DdpOptions(**opts), *a)
    k = 3
    assert sf[0, 1] > k + 1
    xk, uk, sk = cport.solve_batch(cst, oddp.DdpOptions(**dict(opts, max_iters=k)), *a)
    resume = dict(rho=sk[0, 7], theta=1.0 if sk[0, 3] == 1.0 else 0.0, closed=sk[0, 4] == 0.0, mu=sk[0, 5])
    xr, ur, sr, _ = cport.solve_trace(cst, oddp.DdpOptions(**opts), batch["x0"][0], batch["params"][0], xk[0], uk[0], resume=resume)
    assert int(sr[1]) + k == int(sf[0, 1]) and int(sr[2]) == 1
    assert np.max(np.abs(xr - xf[0])) <= 1e-9 and np.max(np.abs(ur - uf[0])) <= 1e-9


@pytest.mark.parametrize("model,N", [("srbd37", 20), ("lip30", 20)])
def test_c_oracle_without_relative_velocity_constraints_matches_numpy(model, N):
    """number_of_legs = 4 x contact_model = 1 (prb.py:166: no relative_vel_* constraints): the flag in both oracles."""
    batch = workload.make_batch(model, N, [2, 9])
    cst = omodels.RobotConsts(**dict(batch["consts"], relative_velocity_constraints=False))
    m = omodels.make_model(model, cst)
    opts = oddp.DdpOptions(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)
    xs, us, st = cport.solve_batch(cst, opts, batch["x0"], batch["params"], batch["xs"], batch["us"], model=model)
    xs1, us1, st1 = cport.solve_batch(omodels.RobotConsts(**batch["consts"]), opts, batch["x0"], batch["params"], batch["xs"], batch["us"], model=model)
    for b in range(2):
        r = oddp.solve(m, batch["x0"][b], batch["params"][b], batch["xs"][b], batch["us"][b], opts)
        assert int(st[b, 1]) == r.iters and bool(st[b, 2]) == r.converged
        assert np.max(np.abs(xs[b] - r.xs)) <= 1e-8 and np.max(np.abs(us[b] - r.us)) <= 1e-8
    if model == "srbd37":
        assert np.max(np.abs(xs - xs1)) > 1e-6          # and it is another problem than the one with the constraints


@pytest.mark.parametrize("model,N", [("srbd13", 12), ("srbd37", 8), ("lip30", 10), ("srbd61", 8)])
def test_c_oracle_with_user_rows_matches_numpy(model, N):
    """user-declared linear rows (sddp.h extra_*; ddp.py:183-196 sums whatever residual the container holds): the C restatement
    (xr_rows) against the numpy one (WithLinearRows) on every model -- a state row with a per-knot reference and a dense stage row"""
    from srbd_horizon_amd import workload
    nx, nu, npb = cport.DIMS[model]
    batch = workload.make_batch(model, N, [3])
    rng = np.random.default_rng(5)
    a0 = np.zeros(nx + nu); a0[0] = 1.0
    a3 = np.zeros(nx + nu); a3[:nx] = 0.05 * rng.standard_normal(nx); a3[nx:] = 0.05 * rng.standard_normal(nu)
    rows = (dict(a=a0, w=2e3, kind="state", const=0.0), dict(a=a3, w=40.0, kind="stage", const=-0.2))
    P = np.concatenate([batch["params"], np.zeros((1, N + 1, 8))], axis=2)
    P[:, :, npb] = 0.02 * np.sin(np.arange(N + 1) / 5.0)[None]
    cst = omodels.RobotConsts(**dict(batch["consts"], extra_rows=rows))
    o = oddp.DdpOptions(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)
    r = oddp.solve(omodels.make_model(model, cst), batch["x0"][0], P[0], batch["xs"][0], batch["us"][0], o)
    xo, uo, so = cport.solve_batch(cst, o, batch["x0"], P, batch["xs"], batch["us"], threads=1, model=model)
    assert r.converged and int(so[0, 1]) == r.iters and int(so[0, 2]) == 1
    assert np.max(np.abs(r.xs - xo[0])) <= 1e-9 and np.max(np.abs(r.us - uo[0])) <= 1e-9 and abs(r.cost - so[0, 0]) <= 1e-11 * abs(r.cost)
    plain = cport.solve_batch(omodels.RobotConsts(**batch["consts"]), o, batch["x0"], batch["params"], batch["xs"], batch["us"], threads=1, model=model)
    assert np.max(np.abs(plain[0][0] - xo[0])) > 1e-5            # the rows act
