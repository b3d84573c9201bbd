"""Error behaviour and surface details of the drop-in boundary on a real device (C ABI status codes -> RuntimeError /
ValueError / KeyError in the Python mirrors, like the reference's Python exceptions)."""
import os

import numpy as np
import pytest

from oracle import cport, ddp as oddp, models as omodels
from srbd_horizon_amd import workload
from srbd_horizon_amd.ddp import DDPSolver
from srbd_horizon_amd.engine import DdpEngine
from srbd_horizon_amd.prb import LIPProblem, SRBD13Problem, SRBDProblem
from tests import shadow
from tests.test_gpu_divergence import assert_batch

pytestmark = pytest.mark.gpu

OPTS = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)      # dsrbd_example.py:55-58


def test_engine_rejects_bad_calls():
    eng = DdpEngine("srbd13", 30, 2)
    batch = workload.make_batch("srbd13", 30, [0, 1])
    with pytest.raises(RuntimeError, match="sddp_set_initial_state"):
        eng.solve(batch["params"])                                   # nothing set yet
    eng.set_initial_state(batch["x0"])
    with pytest.raises(RuntimeError, match="u_warmstart"):
        eng.solve(batch["params"])
    with pytest.raises(ValueError):
        eng.set_u_warmstart(batch["us"][:, :-1])                     # wrong shape
    with pytest.raises(KeyError):
        eng.set_options(not_an_option=1)
    with pytest.raises(RuntimeError, match="line_search_decrease_factor"):
        eng.set_options(line_search_decrease_factor=1.5)
    with pytest.raises(KeyError):
        DdpEngine("srbd13", 30, 1, opts=dict(bogus=1))
    with pytest.raises(KeyError):
        DdpEngine("no_such_model", 30, 1)
    with pytest.raises(RuntimeError):
        DdpEngine("srbd13", 0, 1)                                    # N < 1


def test_non_finite_start_is_reported_not_propagated():
    eng = DdpEngine("srbd13", 30, 2)
    batch = workload.make_batch("srbd13", 30, [0, 1])
    x0 = batch["x0"].copy()
    x0[1, 3:7] = np.nan
    xs = batch["xs"].copy()
    xs[1, :, 3:7] = np.nan
    eng.set_initial_state(x0); eng.set_x_warmstart(xs); eng.set_u_warmstart(batch["us"])
    eng.solve(batch["params"])
    assert eng.stats["status"][0] == 0 and eng.stats["converged"][0] == 1    # the healthy instance is unaffected
    assert eng.stats["status"][1] == 3 and eng.stats["converged"][1] == 0 and eng.stats["iters"][1] == 0


@pytest.mark.parametrize("builder,name,ns", [(SRBDProblem, "createSRBDProblem", 20), (LIPProblem, "createLIPProblem", 20),
                                              (SRBD13Problem, "createSRBD13Problem", 30)])
def test_ddpsolver_surface_matches_reference_adapter(builder, name, ns):
    """Same surface as reference python/ddp.py: DDPSolver(prb, opts), setInitialState, solve() -> bool, getSolutionDict()
    with one [dim, nodes] entry per variable plus 'x_opt' / 'u_opt' (ddp.py:102-104, :125-151)."""
    pb = builder()
    getattr(pb, name)(ns, ns * 0.05)
    solver = DDPSolver(pb.prb, opts=dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3))
    with pytest.raises(RuntimeError):
        solver.solve()                                               # setInitialState first
    x0 = pb.getInitialState()
    solver.setInitialState(x0)
    u_ws = np.repeat(pb.getStaticInput()[:, None], ns, axis=1)       # what dsrbd_example.py:61-68 computes
    solver.set_u_warmstart(u_ws)
    solver.set_x_warmstart(np.repeat(x0[:, None], ns + 1, axis=1))
    assert solver.solve() is True
    sol = solver.getSolutionDict()
    nx, nu = solver.state_size, solver.input_size
    assert sol["x_opt"].shape == (nx, ns + 1) and sol["u_opt"].shape == (nu, ns)
    np.testing.assert_array_equal(sol["x_opt"][:, 0], x0)
    names = [v.getName() for v in pb.prb.var_container.getVarList(offset=False)]
    assert all(n in sol for n in names)
    assert np.vstack([sol[n] for n in names if sol[n].shape[1] == ns + 1]).shape[0] == nx
    assert np.vstack([sol[n] for n in names if sol[n].shape[1] == ns]).shape[0] == nu
    # standing still at the nominal state is (nearly) optimal: the solution stays there
    assert np.max(np.abs(sol["x_opt"] - x0[:, None])) < 0.25           # standing: the optimum stays near the nominal state
    with pytest.raises(KeyError):
        DDPSolver(pb.prb, opts=dict(max_iterations=5))               # unknown option key


def test_default_warm_start_when_none_is_passed():
    """The reference examples never pass a warm start: documented default x = x0 at every node, u = 0."""
    pb = SRBD13Problem()
    pb.createSRBD13Problem(30, 1.5)
    solver = DDPSolver(pb.prb, opts=dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3))
    solver.setInitialState(pb.getInitialState())
    assert solver.solve() in (True, False)
    assert np.all(np.isfinite(solver.getSolutionDict()["x_opt"]))
    it1 = int(solver.stats["iters"])
    solver.solve()                                                   # the solver object keeps its solution as warm start
    assert int(solver.stats["iters"]) <= max(1, it1 // 2)


def test_throughput_build_returns_the_same_results_as_the_latency_build():
    """sddp_options.waves_per_simd only changes the register allocation of the fused kernel (two instances per SIMD): same
    iterates, same iteration counts.  Also valid (and a no-op) on the models that have a single build."""
    seeds = np.arange(96)
    batch = workload.make_batch("srbd13", 30, seeds)
    opts = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)
    res = []
    for w in (1, 2):
        eng = DdpEngine("srbd13", 30, len(seeds), opts=dict(opts, waves_per_simd=w))
        eng.set_initial_state(batch["x0"]); eng.set_x_warmstart(batch["xs"]); eng.set_u_warmstart(batch["us"])
        x, u = eng.solve(batch["params"])
        res.append((x, u, eng.stats.copy()))
    (x1, u1, s1), (x2, u2, s2) = res
    np.testing.assert_array_equal(s1["iters"], s2["iters"])
    np.testing.assert_array_equal(s1["rollouts"], s2["rollouts"])
    np.testing.assert_allclose(x2, x1, rtol=0, atol=1e-9)
    np.testing.assert_allclose(u2, u1, rtol=0, atol=1e-9)
    np.testing.assert_allclose(s2["cost"], s1["cost"], rtol=1e-12)
    with pytest.raises(RuntimeError, match="waves_per_simd"):
        DdpEngine("srbd13", 30, 1, opts=dict(waves_per_simd=3))
    b37 = workload.make_batch("srbd37", 20, [0, 1])
    eng = DdpEngine("srbd37", 20, 2, opts=dict(opts, waves_per_simd=2))
    eng.set_initial_state(b37["x0"]); eng.set_x_warmstart(b37["xs"]); eng.set_u_warmstart(b37["us"])
    eng.solve(b37["params"])
    assert np.all(eng.stats["converged"] == 1)
    # lip30: the tiles of TWO instances fit the LDS of a CU, so waves_per_simd = 2 selects the half-register-file build of the
    # 4-wave kernel (two workgroups per CU); more instances than its slots: a queue.  Same results as the full-register build
    bl = workload.make_batch("lip30", 20, np.arange(700))
    rl = []
    for w in (1, 2):
        eng = DdpEngine("lip30", 20, 700, opts=dict(opts, waves_per_simd=w))
        eng.set_initial_state(bl["x0"]); eng.set_x_warmstart(bl["xs"]); eng.set_u_warmstart(bl["us"])
        x, u = eng.solve(bl["params"])
        rl.append((x, u, eng.stats.copy(), eng.queue_info()))
    np.testing.assert_array_equal(rl[0][2]["iters"], rl[1][2]["iters"])
    np.testing.assert_allclose(rl[1][0], rl[0][0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(rl[1][1], rl[0][1], rtol=0, atol=1e-9)
    assert rl[1][3][1] == 2 * rl[0][3][1] and rl[0][3][2] == rl[1][3][2] == 700, (rl[0][3], rl[1][3])   # twice the resident workgroups, both queued
    # srbd37 since round 3 (Q without a tile of its own: 77 KB per instance): the same, on a queue longer than its 512 slots
    bs = workload.make_batch("srbd37", 20, np.arange(600))
    rs = []
    for w in (1, 2):
        eng = DdpEngine("srbd37", 20, 600, opts=dict(opts, waves_per_simd=w, queue_order=2))
        eng.set_initial_state(bs["x0"]); eng.set_x_warmstart(bs["xs"]); eng.set_u_warmstart(bs["us"])
        x, u = eng.solve(bs["params"])
        rs.append((x, u, eng.stats.copy(), eng.queue_info()))
    np.testing.assert_array_equal(rs[0][2]["iters"], rs[1][2]["iters"])
    np.testing.assert_array_equal(rs[0][2]["status"], rs[1][2]["status"])
    np.testing.assert_allclose(rs[1][0], rs[0][0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(rs[1][1], rs[0][1], rtol=0, atol=1e-9)
    assert rs[1][3][1] == 2 * rs[0][3][1] == 512 and rs[0][3][2] == rs[1][3][2] == 600, (rs[0][3], rs[1][3])
    # the full second-order build of srbd37 (its own kernel instantiation, larger tables) in both register budgets
    b2 = workload.make_batch("srbd37", 20, np.arange(24))
    r2 = []
    for w in (1, 2):
        eng = DdpEngine("srbd37", 20, 24, opts=dict(opts, waves_per_simd=w, second_order=2))
        eng.set_initial_state(b2["x0"]); eng.set_x_warmstart(b2["xs"]); eng.set_u_warmstart(b2["us"])
        x, u = eng.solve(b2["params"])
        r2.append((x, u, eng.stats.copy()))
    np.testing.assert_array_equal(r2[0][2]["iters"], r2[1][2]["iters"])
    np.testing.assert_allclose(r2[1][0], r2[0][0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(r2[1][1], r2[0][1], rtol=0, atol=1e-9)
    assert np.all(r2[0][2]["converged"] == 1)


def test_batches_in_flight_on_separate_streams_equal_sequential_solves():
    """The bench / fleet-server pattern: several handles, each on its own HIP stream, launched back to back with no host
    synchronisation in between (sddp_set_stream, sddp_set_*_device, sddp_solve_device).  Every handle must return exactly what
    a lone sequential solve of its batch returns: handles share nothing on the device."""
    import torch
    N, B, S = 30, 64, 4
    opts = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3, waves_per_simd=2)
    batches = [workload.make_batch("srbd13", N, 1000 * k + np.arange(B)) for k in range(S)]
    ref = []
    for bt in batches:
        e = DdpEngine("srbd13", N, B, opts=opts)
        e.set_initial_state(bt["x0"]); e.set_x_warmstart(bt["xs"]); e.set_u_warmstart(bt["us"])
        x, u = e.solve(bt["params"])
        ref.append((x.copy(), u.copy(), e.stats.copy()))
    dev = torch.device("cuda", 0)
    engs, streams, bufs = [], [], []
    for bt in batches:
        e = DdpEngine("srbd13", N, B, opts=opts)
        st = torch.cuda.Stream()
        e.use_torch_stream(st)
        engs.append(e); streams.append(st)
        bufs.append({k: torch.from_numpy(bt[k]).to(dev) for k in ("x0", "xs", "us", "params")})
    torch.cuda.synchronize()
    for rep in range(2):                                   # second round: the handles are reused while others still run
        for e, st, d in zip(engs, streams, bufs):
            with torch.cuda.stream(st):
                e.set_initial_state_device(d["x0"]); e.set_x_warmstart_device(d["xs"]); e.set_u_warmstart_device(d["us"])
                e.solve_device(d["params"])
    torch.cuda.synchronize()
    for e, (x0, u0, s0) in zip(engs, ref):
        x, u, st = e.fetch()
        np.testing.assert_array_equal(st["iters"], s0["iters"])
        np.testing.assert_array_equal(x, x0)
        np.testing.assert_array_equal(u, u0)
        np.testing.assert_array_equal(st["cost"], s0["cost"])


@pytest.mark.parametrize("model,N", [("srbd13", 30), ("srbd37", 20), ("lip30", 20)])
def test_model_step_is_the_dynamics_of_the_knot_evaluation(model, N):
    """sddp_model_step (the closed-loop simulator step, dsrbd_example.py:158-159) returns the same x+ as the f of
    sddp_eval_knots, i.e. the solver's own Euler step (bit-exact: same device code), and validates its node argument."""
    from srbd_horizon_amd.engine import eval_knots
    B = 5
    batch = workload.make_batch(model, N, np.arange(B))
    eng = DdpEngine(model, N, B, consts=batch["consts"])
    rng = np.random.default_rng(1)
    x = batch["x0"] + 0.01 * rng.standard_normal(batch["x0"].shape)
    u = batch["us"][:, 0] + 0.01 * rng.standard_normal(batch["us"][:, 0].shape)
    for k in (0, N - 1):
        p = batch["params"][:, k]
        xn = eng.model_step(x, u, p, k)
        f, _, _, _, _ = eval_knots(model, N, np.full(B, k), x, u, p, consts=batch["consts"])
        np.testing.assert_array_equal(xn, f)
    with pytest.raises(RuntimeError, match="stage node"):
        eng.model_step(x, u, batch["params"][:, N], N)


@pytest.mark.parametrize("model,ns", [("srbd13", 30), ("srbd37", 20), ("lip30", 20), ("srbd61", 20)])
def test_device_resident_receding_horizon_equals_the_host_shift(model, ns):
    """sddp_set_params / sddp_advance / sddp_solve_resident: shifting the parameter tensor and the warm
    start on the device gives the same ticks as shifting them on the host and passing them through sddp_solve."""
    from srbd_horizon_amd.mpc import MpcLoop
    host, dev = MpcLoop(model, ns, warm_start="shift"), MpcLoop(model, ns, warm_start="device")
    for t in range(8):
        motion = "walking" if t < 6 else "standing"
        ch, sh = host.tick(motion, (1.0, 0.5))
        cd, sd = dev.tick(motion, (1.0, 0.5))
        assert ch == cd and host.solver.stats["iters"] == dev.solver.stats["iters"]
        np.testing.assert_array_equal(sd["x_opt"], sh["x_opt"])
        np.testing.assert_array_equal(sd["u_opt"], sh["u_opt"])
        np.testing.assert_array_equal(dev.state, host.state)
    assert dev.solver.resyncs == 0                                   # the example loop only shifts and assigns node N
    # a parameter assigned BELOW node N between two ticks (nothing the reference loop does): the host shadow notices and the
    # whole tensor is uploaded again, so the device cannot drift from the problem's parameters
    for lp in (host, dev):
        lp.srbd.rdot_ref.assign([0.3, -0.2, 0.0], nodes=3)
    ch, sh = host.tick("walking", (1.0, 0.5))
    cd, sd = dev.tick("walking", (1.0, 0.5))
    assert dev.solver.resyncs == 1 and ch == cd
    np.testing.assert_array_equal(sd["x_opt"], sh["x_opt"])
    np.testing.assert_array_equal(sd["u_opt"], sh["u_opt"])
    ch, sh = host.tick("walking", (1.0, 0.5))
    cd, sd = dev.tick("walking", (1.0, 0.5))
    assert dev.solver.resyncs == 1                                   # back on the shift-only path
    np.testing.assert_array_equal(sd["x_opt"], sh["x_opt"])
    e = DdpEngine(model, ns, 1)
    with pytest.raises(RuntimeError, match="sddp_set_params"):
        e.advance(np.zeros((1, e.np_)), np.zeros((1, e.nx)))


def test_fleet_tick_of_1024_robots_matches_the_c_oracle_tick_by_tick(record_property):
    """The path behind bench.py's `ms_per_fleet_tick` (VERDICT r02 #4a): B = 1024 robots, each tick = sddp_advance (parameters and
    previous solution shifted by one knot on the device, new last parameter column and measured state uploaded) +
    sddp_solve_resident, against the C oracle solving the same tick from the HOST-side shift of the same data
    (dsrbd_example.py:102-135).  The oracle is fed the GPU's previous solution, so every tick is compared on its own."""
    N, B, ticks = 30, 1024, int(os.environ.get("SDDP_SOAK_TICKS", "6"))      # (a soak run by hand: 200 ticks)
    b = workload.make_batch("srbd13", N, np.arange(B) + 20000)
    eng = DdpEngine("srbd13", N, B, opts=dict(OPTS, waves_per_simd=1))
    eng.set_initial_state(b["x0"]); eng.set_x_warmstart(b["xs"]); eng.set_u_warmstart(b["us"])
    eng.set_params(b["params"])
    x, u = eng.solve_resident()                                   # every robot's first (cold) tick
    P = b["params"].copy()
    cst, o = omodels.RobotConsts(**b["consts"]), oddp.DdpOptions(**OPTS)
    rng = np.random.default_rng(5)
    for t in range(ticks):
        p_last = P[:, -1].copy()
        p_last[:, 0:2] += 0.05 * rng.standard_normal((B, 2))     # the commanded velocity drifts: node N differs from node N-1
        x0 = x[:, 1] + 1e-3 * rng.standard_normal((B, 13))       # the robot is near, not at, where the plan said
        xs_ws = np.concatenate([x[:, 1:], x[:, -1:]], axis=1)
        us_ws = np.concatenate([u[:, 1:], u[:, -1:]], axis=1)
        P = np.concatenate([P[:, 1:], p_last[:, None]], axis=1)
        eng.advance(p_last, x0)
        x, u = eng.solve_resident()
        st = eng.stats.copy()
        # every robot of the tick against the C oracle; robots on another path: every accepted GPU step shadowed (tests/shadow.py)
        res = shadow.compare_batch("srbd13", N, dict(x0=x0, params=P, xs=xs_ws, us=us_ws, consts=b["consts"]), OPTS, dict(waves_per_simd=1),
                                   cst, x, u, st, threads=min(16, os.cpu_count() or 1))
        assert_batch(res, f"fleet_tick_{t}", record_property)
        assert (res["so"][:, 2] == 1).mean() >= 0.98
    assert eng.stats["iters"].mean() < 8                          # warm-started ticks, not cold solves


def test_first_knot_fetch_equals_the_full_fetch():
    """sddp_solve_resident_first (a fleet's closed-loop tick: only u_0, x_1, cost, iterations, status leave the device) returns
    exactly what sddp_solve_resident returns at those positions, over a few ticks, and leaves the trajectories on the device."""
    N, B = 30, 200
    b = workload.make_batch("srbd13", N, np.arange(B) + 300)
    engs = []
    for _ in range(2):
        e = DdpEngine("srbd13", N, B, opts=OPTS)
        e.set_initial_state(b["x0"]); e.set_x_warmstart(b["xs"]); e.set_u_warmstart(b["us"]); e.set_params(b["params"])
        engs.append(e)
    full, first = engs
    p_last = b["params"][:, -1].copy()
    x, u = full.solve_resident()
    u0, x1 = first.solve_resident_first()
    for t in range(4):
        np.testing.assert_array_equal(u0, u[:, 0])
        np.testing.assert_array_equal(x1, x[:, 1])
        np.testing.assert_array_equal(first.first_stats["iters"], full.stats["iters"])
        np.testing.assert_array_equal(first.first_stats["status"], full.stats["status"])
        np.testing.assert_array_equal(first.first_stats["cost"], full.stats["cost"])
        xf, uf, _ = first.fetch()                                       # the whole solution is still there
        np.testing.assert_array_equal(xf, x); np.testing.assert_array_equal(uf, u)
        full.advance(p_last, x[:, 1].copy()); first.advance(p_last, x1.copy())
        x, u = full.solve_resident()
        u0, x1 = first.solve_resident_first()
