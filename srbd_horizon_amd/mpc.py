"""Receding-horizon MPC loop: the body of reference python/dsrbd_example.py:82-185 (SRBD) and of
python/dlip_example.py:89-160 (LIP, ``model="lip30"``) with ROS stripped.

Per tick (dsrbd_example.py line numbers): setInitialState (:84) -> shift rdot_ref / w_ref / oref / orientation gain back by
one node (:102-106, one slice move per parameter) -> assign the commanded velocity at node ns (:109-124) -> wpg.set(action)
(:126-131) -> solve, timed (:134-136: this is the "ms/MPC-tick" metric) -> simulate one Euler step with the first input
and renormalise the quaternion (:158-160).  The closed-loop simulator step runs through the same HIP model code as the solver
(`sddp_model_step`), so there is no second implementation of the dynamics on the host.
"""
from __future__ import annotations

import time

import numpy as np

from . import wpg as _wpg
from .ddp import DDPSolver
from .problem import Parameter
from .prb import LIPProblem, SRBD13Problem, SRBDProblem

# reference option set (dsrbd_example.py:55-58)
EXAMPLE_OPTS = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)


class MpcLoop:
    def __init__(self, model: str = "srbd37", ns: int = 20, T: float | None = None, opts: dict | None = None, robot=None,
                 warm_start: str = "shift", number_of_legs: int = 2, contact_model: int | None = None):
        """warm_start: "shift" = previous solution advanced by one knot (last knot repeated),
        "device" = the same, with the parameter tensor and the warm start resident on the GPU and shifted there (only the new
        last parameter column and the state cross PCIe), "previous" = previous solution as is (what a stateful pyddp object
        would keep), "reset" = x0 repeated / static input."""
        T = ns * 0.05 if T is None else T                              # wpg hard-codes dt = 0.05 (wpg.py:20)
        legs = 2
        if model in ("srbd37", "srbd61"):
            # rosparams number_of_legs / contact_model (prb.py:39-40): srbd37 = 2 x 2 (the launch file's) or 4 x 1 (four point feet)
            cm = contact_model if contact_model is not None else (2 if model == "srbd37" else 4)
            legs = number_of_legs
            self.srbd = SRBDProblem()
            self.srbd.createSRBDProblem(ns, T, robot, params=dict(contact_model=cm, number_of_legs=legs))
            if self.srbd.prb.model != model:
                raise ValueError(f"number_of_legs = {legs}, contact_model = {cm} is model {self.srbd.prb.model}, not {model}")
            contact_model = self.srbd.contact_model
        elif model == "srbd13":
            self.srbd = SRBD13Problem()
            self.srbd.createSRBD13Problem(ns, T, robot)
            contact_model = 1
        elif model == "lip30":
            # dlip_example.py:49-52 builds an SRBD problem as well and hands ITS f / w_ref / orientation gain to the scheduler
            # (:86-87); the LIP problem has no such parameters, so the scheduler writes into stand-ins that nothing reads
            self.srbd = LIPProblem()
            self.srbd.createLIPProblem(ns, T, robot)
            contact_model = self.srbd.contact_model
            self.srbd.f = None
            self.srbd.w_ref = Parameter("w_ref", 3, ns + 1)
            self.srbd.orientation_tracking_gain = Parameter("orientation_tracking_gain", 1, ns + 1)
        else:
            raise ValueError(model)
        self.model, self.ns = model, ns
        self.warm_start = warm_start
        self.solver = DDPSolver(self.srbd.prb, opts=dict(EXAMPLE_OPTS if opts is None else opts))
        self.state = self.srbd.getInitialState().astype(float)
        c_init_z = float(self.srbd.initial_foot_position[0][2])
        self.wpg = _wpg.steps_phase(self.srbd.f, self.srbd.c, self.srbd.cdot, c_init_z, self.srbd.c_ref, self.srbd.w_ref,
                                    self.srbd.orientation_tracking_gain, self.srbd.cdot_switch, ns, number_of_legs=legs,
                                    contact_model=contact_model)
        # warm start the first solve like dsrbd_example.py:61-68 computes it (x = state at every node, u = static input)
        self.solver.set_x_warmstart(np.repeat(self.state[:, None], ns + 1, axis=1))
        self.solver.set_u_warmstart(np.repeat(self.srbd.getStaticInput()[:, None], ns, axis=1))
        self._prev_sw = [1.0, 1.0]     # srbd13: contact switch of each foot at node ns after the previous tick (touchdown detection)
        self.solve_ms = []
        self.trace = None          # set to a list to record every tick's solver inputs (x0, params, warm start): bench.py replays them
        self._last = None

    def tick(self, motion: str = "standing", axes=(0.0, 0.0)):
        s, ns = self.srbd, self.ns
        if self.warm_start != "device":
            self.solver.setInitialState(self.state)                                    # :84
        shifted = (s.rdot_ref,) if self.model == "lip30" else (s.rdot_ref, s.w_ref, s.oref, s.orientation_tracking_gain)
        for par in shifted:                                                            # :102-106 / dlip_example.py:108-109
            par.values[:, :ns] = par.values[:, 1:ns + 1]
        a = 0.1 if motion == "standing" else 0.5                                       # :109-112
        s.rdot_ref.assign([a * axes[0], a * axes[1], 0.0], nodes=ns)                   # :119-122
        self.wpg.set({"walking": "step", "jumping": "jump"}.get(motion, "standing"))   # :126-131
        if self.model == "srbd13":
            # The metric model's contacts are per-knot DATA, so the footstep plan has to ride with the horizon
            # like every other parameter (the reference's contacts are states and move by themselves): shift the xy rows, repeat the
            # last node, and at a touchdown (switch 0 -> 1 at node ns) move that foot one stride = commanded velocity x 1 s step
            # cycle ahead -- the plan of workload.schedule_by_ticking, which the bench batch is built from
            stride = np.array([a * axes[0], a * axes[1]]) * 1.0
            for i in range(2):
                cxy = s.c[i].values
                cxy[0:2, :ns] = cxy[0:2, 1:ns + 1]
                sw_new = float(s.cdot_switch[i].values[0, ns])
                if self._prev_sw[i] == 0.0 and sw_new == 1.0:
                    cxy[0:2, ns] = cxy[0:2, ns - 1] + stride
                self._prev_sw[i] = sw_new
        if self.trace is not None:
            self.trace.append(self._solver_inputs())
        t0 = time.perf_counter()                                                       # :134 tic()
        if self.warm_start == "device":
            converged = self.solver.solve_receding(self.state)                         # :84 + :135, device-resident data
        else:
            converged = self.solver.solve()                                            # :135
        self.solve_ms.append(1e3 * (time.perf_counter() - t0))                         # :136 toc()
        sol = self.solver.getSolutionDict()                                            # :137
        self._last = sol
        u0 = sol["u_opt"][:, 0]                                                        # :158
        p0 = s.prb.parameter_matrix()[0]
        self.state = self.solver.ddp_solver.model_step(self.state[None], u0[None], p0[None], 0)[0]   # :159 Euler step (same HIP model)
        if self.model != "lip30":
            self.state[3:7] /= np.linalg.norm(self.state[3:7])                         # :160 (the LIP state has no quaternion)
        if self.warm_start == "shift":
            x, u = sol["x_opt"], sol["u_opt"]
            self.solver.set_x_warmstart(np.concatenate([x[:, 1:], x[:, -1:]], axis=1))
            self.solver.set_u_warmstart(np.concatenate([u[:, 1:], u[:, -1:]], axis=1))
        elif self.warm_start == "reset":
            self.solver.set_x_warmstart(np.repeat(self.state[:, None], ns + 1, axis=1))
            self.solver.set_u_warmstart(np.repeat(s.getStaticInput()[:, None], ns, axis=1))
        return converged, sol

    def _solver_inputs(self):
        """What this tick's solve starts from, in the C ABI's knot-major layout (warm_start "device" / "shift": the previous
        solution advanced by one knot, last knot repeated; first tick: state at every node, static input)."""
        ns = self.ns
        if self._last is None or self.warm_start == "reset":
            xs = np.repeat(self.state[None, :], ns + 1, axis=0)
            us = np.repeat(self.srbd.getStaticInput()[None, :], ns, axis=0)
        elif self.warm_start == "previous":
            xs, us = self._last["x_opt"].T.copy(), self._last["u_opt"].T.copy()
        else:
            x, u = self._last["x_opt"], self._last["u_opt"]
            xs = np.concatenate([x[:, 1:], x[:, -1:]], axis=1).T.copy()
            us = np.concatenate([u[:, 1:], u[:, -1:]], axis=1).T.copy()
        return dict(x0=self.state.copy(), params=self.srbd.prb.parameter_matrix().copy(), xs=xs, us=us)

    def reference_record(self, sol, node: int = 1, foot_frames=("left_sole_link", "right_sole_link")):
        """ROS-free form of what the reference hands to CartesIO every tick (cartesio.py:58-79, called at
        dsrbd_example.py:179-181): com position, base_link orientation (quaternion x,y,z,w) and one position per foot frame
        -- the midpoint of a line foot's two contact points (cartesio.py:68-72), the contact itself for a point foot -- all
        taken from the solution at ``node`` (the reference publishes node 1, the next tick's target)."""
        o = sol["o"][:, node] if "o" in sol else np.array([0.0, 0.0, 0.0, 1.0])          # dlip_example.py:145 publishes identity
        rec = {"com": np.array(sol["r"][:, node]), "base_link": np.array(o), "contacts": {}}
        if self.model in ("srbd37", "srbd61", "lip30"):
            cm = self.srbd.contact_model
            for leg, frame in enumerate(foot_frames):
                pts = [sol["c" + str(leg * cm + j)][:, node] for j in range(cm)]
                rec["contacts"][frame] = np.mean(pts, axis=0)
        else:                                   # srbd13: contacts are parameters of the plan, not decision variables
            for leg, frame in enumerate(foot_frames):
                rec["contacts"][frame] = np.array(self.srbd.c[leg].values[:, node])
        return rec

    def run(self, ticks: int, motion="walking", axes=(1.0, 0.0)):
        out = []
        for _ in range(ticks):
            out.append(self.tick(motion, axes)[0])
        return out
