"""Synthetic, seeded MPC instances: what the bench, the smoke test and the parity tests solve.

Per instance ``seed``:
  * contact schedule: the walking-pattern scheduler (wpg ``"step"`` action, reference python/wpg.py:80-88) started at
    phase ``seed mod 20`` and advanced ``N + 1 + seed mod 20`` ticks so the horizon is fully populated; the example
    loop's per-tick back-shift of rdot_ref / w_ref / oref / orientation gain (python/dsrbd_example.py:102-106) applied;
  * velocity command: rdot_ref = (0.5 a_x, 0.5 a_y, 0), a ~ U{-1,0,1}^2 (dsrbd_example.py:112, :119-122);
  * footsteps (srbd13 only, contacts are data): each touchdown INSIDE the horizon moves that foot by one stride =
    rdot_ref_xy * 1 s; at node 0 the feet are at their nominal places under the (perturbed) nominal initial state;
  * initial state: nominal + N(0, sigma) (1 cm, 5 cm/s, 0.05 rad/s, small-angle quaternion 0.02), renormalised;
  * warm start: x = x0 at every node, u = static input (what dsrbd_example.py:61-68 computes).

Everything is computed in closed form, vectorised over the batch: node j of the final horizon was written at tick
t = T - (N - j), so its value is ``cycle[(phase + t - 1) mod 20]``.  ``schedule_by_ticking`` does the same thing by
literally ticking ``wpg.steps_phase`` and is used to test the closed form.
"""
from __future__ import annotations

import numpy as np

from . import wpg as _wpg
from .prb import LIPProblem, RobotModel, SRBD13Problem, SRBDProblem

G = 9.81


def _schedule_tables(c_init_z=0.0):
    step_nodes, l_cycle, l_sw, r_cycle, r_sw = _wpg.step_tables(c_init_z)
    return step_nodes, np.stack([l_cycle, r_cycle]), np.stack([l_sw, r_sw])


def _closed_form(N, seeds, init_z, init_sw, init_otg):
    """-> z_ref [B,2,N+1], sw [B,2,N+1], otg [B,N+1], n_td [B,2,N+1] (touchdowns of each foot up to node j)."""
    seeds = np.asarray(seeds, dtype=np.int64)
    B = seeds.shape[0]
    step_nodes, cyc, swt = _schedule_tables(0.0)
    period = 2 * step_nodes
    phase = seeds % period
    T = N + 1 + phase                                    # ticks executed
    j = np.arange(N + 1)[None, :]
    t = T[:, None] - (N - j)                             # tick (1-based) that wrote node j; all >= 1 here
    idx = (phase[:, None] + t - 1) % period
    z = np.stack([cyc[0][idx], cyc[1][idx]], axis=1)
    sw = np.stack([swt[0][idx], swt[1][idx]], axis=1)
    otg = np.full((B, N + 1), 1e2)                       # wpg writes 1e2 at node N each tick (wpg.py:82), then shifted back
    # touchdowns: switch goes 0 -> 1 between consecutive ticks; count along the global tick timeline
    n_td = np.zeros((B, 2, N + 1))
    tick_all = np.arange(1, int(T.max()) + 1)
    for leg in range(2):
        sw_all = swt[leg][(phase[:, None] + tick_all[None, :] - 1) % period]          # [B, Tmax]
        prev = np.concatenate([np.ones((B, 1)), sw_all[:, :-1]], axis=1)
        td = ((prev == 0.0) & (sw_all == 1.0)).astype(float)
        cum = np.cumsum(td, axis=1)
        n_td[:, leg, :] = np.take_along_axis(cum, t - 1, axis=1)
    n_td = n_td - n_td[:, :, :1]                         # count from node 0: the plan is relative to the current state
    return z, sw, otg, n_td


def make_srbd13_batch(N: int, seeds, robot: RobotModel | None = None, x0_draw: int = 0):
    """-> dict(x0 [B,13], params [B,N+1,19], xs [B,N+1,13], us [B,N,6], consts) for the metric model.
    x0_draw != 0: the same robots (schedule, velocity command, footsteps) with ANOTHER draw of the initial-state perturbation:
    a robot's next problem, similar to its last one but not the same (bench.py: what a history-ordered queue sees)."""
    robot = robot or RobotModel()
    seeds = np.asarray(seeds, dtype=np.int64)
    B = seeds.shape[0]
    pb = SRBD13Problem()
    pb.createSRBD13Problem(N, N * 0.05, robot)
    z, sw, otg, n_td = _closed_form(N, seeds, 0.0, 1.0, 1e1)
    a = np.stack([np.random.default_rng(int(s)).integers(-1, 2, size=2) for s in seeds]).astype(float)
    v = 0.5 * a                                                              # alphaX = alphaY = 0.5 when walking
    P = np.zeros((B, N + 1, 19))
    P[:, :, 0:2] = v[:, None, :]                                             # rdot_ref
    P[:, :, 6] = otg
    P[:, :, 7:11] = np.array([-0.0, -0.0, -0.0, 1.0])                        # oref
    for leg in range(2):
        c0 = pb.initial_foot_position[leg]
        P[:, :, 11 + 3 * leg + 0] = c0[0] + v[:, None, 0] * 1.0 * n_td[:, leg, :]
        P[:, :, 11 + 3 * leg + 1] = c0[1] + v[:, None, 1] * 1.0 * n_td[:, leg, :]
        P[:, :, 11 + 3 * leg + 2] = z[:, leg, :]
        P[:, :, 17 + leg] = sw[:, leg, :]
    x0 = np.tile(pb.getInitialState(), (B, 1))
    for b, s in enumerate(seeds):
        rng = np.random.default_rng(int(s) + 1_000_003 + 7_919 * int(x0_draw))
        x0[b, 0:3] += 0.01 * rng.standard_normal(3)
        dq = 0.02 * rng.standard_normal(3)
        q = np.array([dq[0], dq[1], dq[2], 1.0])
        x0[b, 3:7] = q / np.linalg.norm(q)
        x0[b, 7:10] += 0.05 * rng.standard_normal(3)
        x0[b, 10:13] += 0.05 * rng.standard_normal(3)
    xs = np.repeat(x0[:, None, :], N + 1, axis=1)
    us = np.tile(pb.getStaticInput(), (B, N, 1))
    return dict(x0=x0, params=P, xs=xs, us=us, consts=pb.prb.model_consts, problem=pb)


def schedule_by_ticking(N: int, seed: int, robot: RobotModel | None = None):
    """The same srbd13 parameter tensor, produced by literally running the receding-horizon bookkeeping."""
    robot = robot or RobotModel()
    pb = SRBD13Problem()
    pb.createSRBD13Problem(N, N * 0.05, robot)
    gen = _wpg.steps_phase(pb.f, pb.c, pb.cdot, 0.0, pb.c_ref, pb.w_ref, pb.orientation_tracking_gain, pb.cdot_switch,
                           N, number_of_legs=2, contact_model=1)
    phase = seed % 20
    gen.step_counter = phase
    a = np.random.default_rng(int(seed)).integers(-1, 2, size=2).astype(float)
    v = 0.5 * a
    pb.rdot_ref.assign([v[0], v[1], 0.0])
    prev_sw = [1.0, 1.0]
    for _ in range(N + 1 + phase):
        for par in (pb.rdot_ref, pb.w_ref, pb.oref, pb.orientation_tracking_gain):     # dsrbd_example.py:102-106
            par.values[:, :N] = par.values[:, 1:N + 1]
        for i in range(2):                                                              # footstep xy rides with the plan
            pb.c[i].values[0:2, :N] = pb.c[i].values[0:2, 1:N + 1]
        pb.rdot_ref.assign([v[0], v[1], 0.0], nodes=N)
        gen.set("step")
        for i in range(2):
            sw_new = pb.cdot_switch[i].values[0, N]
            xy = pb.c[i].values[0:2, N - 1].copy()
            if prev_sw[i] == 0.0 and sw_new == 1.0:
                xy = xy + v * 1.0
            pb.c[i].values[0:2, N] = xy
            prev_sw[i] = sw_new
    for i in range(2):                                    # the plan is relative to the current state (node 0 = nominal feet)
        off = pb.c[i].values[0:2, 0] - pb.initial_foot_position[i][0:2]
        pb.c[i].values[0:2, :] -= off[:, None]
    return pb.prb.parameter_matrix()


def make_srbd37_batch(N: int, seeds, robot: RobotModel | None = None, contact_model: int = 2):
    """Reference-faithful SRBD (nc = 4 line feet, launch:16-17): the line-foot contact schedule of config 5.
    contact_model = 4: the same schedule on the problem's code-default nc = 8 (srbd61, prb.py:39-41)."""
    robot = robot or RobotModel()
    seeds = np.asarray(seeds, dtype=np.int64)
    B = seeds.shape[0]
    nc = 2 * contact_model
    pb = SRBDProblem()
    pb.createSRBDProblem(N, N * 0.05, robot, params=dict(contact_model=contact_model))
    z, sw, otg, _ = _closed_form(N, seeds, 0.0, 1.0, 1e1)
    a = np.stack([np.random.default_rng(int(s)).integers(-1, 2, size=2) for s in seeds]).astype(float)
    P = np.zeros((B, N + 1, 11 + 2 * nc))
    P[:, :, 0:2] = 0.5 * a[:, None, :]
    P[:, :, 6] = otg
    for i in range(nc):
        leg = 0 if i < contact_model else 1                   # wpg.py:84: contacts i < contact_model belong to the left foot
        P[:, :, 7 + 2 * i] = z[:, leg, :]
        P[:, :, 8 + 2 * i] = sw[:, leg, :]
    P[:, :, 7 + 2 * nc:11 + 2 * nc] = np.array([-0.0, -0.0, -0.0, 1.0])
    x0 = np.tile(pb.getInitialState(), (B, 1))
    rd0 = 7 + 3 * nc
    for b, s in enumerate(seeds):
        rng = np.random.default_rng(int(s) + 1_000_003)
        x0[b, 0:3] += 0.01 * rng.standard_normal(3)
        dq = 0.02 * rng.standard_normal(3)
        q = np.array([dq[0], dq[1], dq[2], 1.0])
        x0[b, 3:7] = q / np.linalg.norm(q)
        x0[b, rd0:rd0 + 3] += 0.05 * rng.standard_normal(3)
        x0[b, rd0 + 3:rd0 + 6] += 0.05 * rng.standard_normal(3)
    xs = np.repeat(x0[:, None, :], N + 1, axis=1)
    us = np.tile(pb.getStaticInput(), (B, N, 1))
    return dict(x0=x0, params=P, xs=xs, us=us, consts=pb.prb.model_consts, problem=pb)


def make_lip30_batch(N: int, seeds, robot: RobotModel | None = None):
    robot = robot or RobotModel()
    seeds = np.asarray(seeds, dtype=np.int64)
    B = seeds.shape[0]
    pb = LIPProblem()
    pb.createLIPProblem(N, N * 0.05, robot)
    z, sw, _, _ = _closed_form(N, seeds, 0.0, 1.0, 1e1)
    a = np.stack([np.random.default_rng(int(s)).integers(-1, 2, size=2) for s in seeds]).astype(float)
    P = np.zeros((B, N + 1, 11))
    P[:, :, 0:2] = 0.5 * a[:, None, :]
    for i in range(4):
        leg = 0 if i < 2 else 1
        P[:, :, 3 + 2 * i] = z[:, leg, :]
        P[:, :, 4 + 2 * i] = sw[:, leg, :]
    x0 = np.tile(pb.getInitialState(), (B, 1))
    for b, s in enumerate(seeds):
        rng = np.random.default_rng(int(s) + 1_000_003)
        x0[b, 0:3] += 0.01 * rng.standard_normal(3)
        x0[b, 15:18] += 0.05 * rng.standard_normal(3)
    xs = np.repeat(x0[:, None, :], N + 1, axis=1)
    us = np.tile(pb.getStaticInput(), (B, N, 1))
    return dict(x0=x0, params=P, xs=xs, us=us, consts=pb.prb.model_consts, problem=pb)


def make_srbd61_batch(N: int, seeds, robot: RobotModel | None = None):
    return make_srbd37_batch(N, seeds, robot, contact_model=4)


MAKERS = {"srbd13": make_srbd13_batch, "srbd37": make_srbd37_batch, "lip30": make_lip30_batch, "srbd61": make_srbd61_batch}


def make_batch(model: str, N: int, seeds, robot: RobotModel | None = None):
    return MAKERS[model](N, seeds, robot)


def srbd13_schedule_classes(params, signed=True):
    """Class label of every instance of an srbd13 batch (sddp.h queue_order 3): what the caller knows about the problem BEFORE it is
    solved -- which feet are in stance at node 0, how many nodes until the first contact switch, and the commanded forward / lateral
    velocity at the end of the horizon as zero / positive / negative (signed=False: zero / non-zero, round 5's first labels; the sign
    decides on which side of the stance foot the plan leads, and the list-scheduling model prices it at 1.15 against 1.23 x the
    ideal makespan: profiles/r05/queue_keys.txt).  params [B, N+1, 19] (prb.py layout: rdot_ref 0:3, cdot_switch 17:19).
    -> (labels [B] int32, n_classes)"""
    P = np.asarray(params)
    N = P.shape[1] - 1
    sw = P[:, :, 17:19] > 0.5
    stance0 = sw[:, 0, 0].astype(np.int64) * 2 + sw[:, 0, 1].astype(np.int64)                 # 0..3
    changed = np.any(sw != sw[:, :1, :], axis=2)                                              # [B, N+1]
    first_change = np.where(changed.any(axis=1), changed.argmax(axis=1), N + 1)               # 1..N, N+1: never
    nv = 3 if signed else 2

    def cmd(v):                                                                               # 0: none, 1: positive, 2: negative
        c = (np.abs(v) > 1e-12).astype(np.int64)
        return c + ((v < -1e-12) & signed) if signed else c
    label = ((stance0 * (N + 2) + first_change) * nv + cmd(P[:, N, 0])) * nv + cmd(P[:, N, 1])
    return label.astype(np.int32), 4 * (N + 2) * nv * nv
