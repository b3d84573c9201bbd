"""Problem builders: the surface of reference python/prb.py (``SRBDProblem`` / ``LIPProblem``) over registered analytic
HIP models instead of CasADi graphs.

The reference pulls mass / inertia / CoM / foot positions out of a URDF through casadi_kin_dyn and its gains out of a
ROS parameter server (prb.py:23-28, :92-95, :130-139, :142-150).  Neither exists here: a ``RobotModel`` carries the
constants (synthetic Kangaroo-like defaults) and ``params`` is a plain dict with the rosparam
names and defaults.  Variables and parameters are created in the reference's order so that the state / input / parameter
layouts (prb.py:224-246; ddp.py:173-177) are identical.

``SRBD13Problem`` is the reduced model BASELINE.json's metric is quoted on (nx=13, nu=6): one point
contact per foot, contact positions are per-knot parameters.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .problem import Aggregate, ParameterRow, Problem, Term


@dataclass
class RobotModel:
    """What ``cas_kin_dyn.CasadiKinDyn(urdf)`` provides at ``joint_init`` in the reference (prb.py:92-95, :130-139)."""
    m: float = 40.0
    I: np.ndarray = field(default_factory=lambda: np.array([[2.0, 0.03, -0.02], [0.03, 1.8, 0.04], [-0.02, 0.04, 0.6]]))
    com: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, 0.88]))
    # left_foot_upper, left_foot_lower, right_foot_upper, right_foot_lower (launch:24-25)
    feet: np.ndarray = field(default_factory=lambda: np.array(
        [[0.08, 0.1, 0.0], [-0.08, 0.1, 0.0], [0.08, -0.1, 0.0], [-0.08, -0.1, 0.0]]))
    # contact_model = 4 (the default in the code, prb.py:39): the four corners of the left sole, then of the right one
    feet8: np.ndarray = field(default_factory=lambda: np.array(
        [[0.08, 0.13, 0.0], [-0.08, 0.13, 0.0], [0.08, 0.07, 0.0], [-0.08, 0.07, 0.0],
         [0.08, -0.07, 0.0], [-0.08, -0.07, 0.0], [0.08, -0.13, 0.0], [-0.08, -0.13, 0.0]]))


# rosparam names and defaults (prb.py:39-40, :142-150, :358-362)
DEFAULT_PARAMS = dict(contact_model=2, number_of_legs=2, r_tracking_gain=1e3, rdot_tracking_gain=1e4, w_tracking_gain=1e4,
                      rel_position_gain=1e4, force_switch_weight=1e2, min_qddot_gain=1e0, min_f_gain=1e-2,
                      zmp_tracking_gain=1e3, inertia_mode=0, lever_sign=1.0,
                      # prb.py:174 rosparam; the barrier is this build's opt-in for the inequality handling the reference disables
                      friction_cone_coefficient=0.8, friction_barrier_weight=0.0, friction_barrier_sharpness=1.0)


def quat_inverse(q):
    """reference python/utilities.py:34-37 (in place, aliases its input)."""
    p = q
    p[0:3] = -p[0:3]
    return p


def _consts(robot: RobotModel, prm: dict, dt: float, feet) -> dict:
    return dict(relative_velocity_constraints=int(prm["contact_model"] > 1),      # prb.py:166: `if contact_model > 1`
                m=robot.m, I=np.asarray(robot.I), com=np.asarray(robot.com), feet=np.asarray(feet), dt=dt,
                force_scaling=1000.0, r_tracking_gain=prm["r_tracking_gain"], rdot_tracking_gain=prm["rdot_tracking_gain"],
                w_tracking_gain=prm["w_tracking_gain"], rel_pos_gain=prm["rel_position_gain"],
                force_switch_weight=prm["force_switch_weight"], min_qddot_gain=prm["min_qddot_gain"],
                min_f_gain=prm["min_f_gain"], zmp_tracking_gain=prm["zmp_tracking_gain"], lip_height=0.88,
                inertia_mode=int(prm["inertia_mode"]), lever_sign=float(prm["lever_sign"]),
                friction_cone_coefficient=float(prm["friction_cone_coefficient"]),
                friction_barrier_weight=float(prm["friction_barrier_weight"]),
                friction_barrier_sharpness=float(prm["friction_barrier_sharpness"]))


def _declare_srbd_terms(prb, prm, ns, nc, contact_model, with_contact_states: bool):
    """The cost terms and constraints of prb.py:166-204, declared in the reference's order with its names and node ranges.
    `with_contact_states`: the reference problem (contacts are states); False: the metric model (contacts are parameters, so
    the contact penalties and rel_pos terms do not exist)."""
    if with_contact_states:
        if contact_model > 1:                                                             # prb.py:166-170
            for i in range(1, contact_model):
                prb.createConstraint("relative_vel_left_" + str(i), Term("relative_vel", dim=2))
            for i in range(contact_model + 1, 2 * contact_model):
                prb.createConstraint("relative_vel_right_" + str(i), Term("relative_vel", dim=2))
    if prm["friction_barrier_weight"] > 0.0:
        # prb.py:172-177 builds the linearised friction cone per contact force and leaves its createIntermediateConstraint
        # commented out, so the reference's container never holds it.  It is declared (as the inequality A f <= 0 it is) only when
        # this build's opt-in barrier is switched on; the adapter then finds it among the inequality constraints (ddp.py:46-48)
        for i in range(nc):
            prb.createIntermediateConstraint(f"f{i}_friction_cone", Term("friction_cone", "friction_cone_coefficient",
                                                                        prm["friction_cone_coefficient"], dim=5),
                                             bounds=dict(lb=-np.inf, ub=0.0))
    if with_contact_states:
        for i in range(nc):                                                               # prb.py:179-181
            prb.createConstraint("cz_tracking" + str(i), Term("cz_tracking"))
            prb.createConstraint("cdotxy_tracking" + str(i), Term("cdotxy_tracking", dim=2))
    st = range(1, ns + 1)
    prb.createResidual("rz_tracking", Term("rz_tracking", "r_tracking_gain", prm["r_tracking_gain"]), nodes=st)          # :184
    prb.createResidual("o_tracking_xyz", Term("o_tracking", dim=3), nodes=st)                                              # :188
    prb.createResidual("o_tracking_w", Term("o_tracking"), nodes=st)                                                       # :189
    prb.createResidual("rdot_tracking", Term("rdot_tracking", "rdot_tracking_gain", prm["rdot_tracking_gain"], 3), nodes=st)   # :190
    prb.createResidual("w_tracking", Term("w_tracking", "w_tracking_gain", prm["w_tracking_gain"], 3), nodes=st)          # :191
    if with_contact_states:
        for nm in ("rel_pos_y_1_4", "rel_pos_x_1_4", "rel_pos_y_3_6", "rel_pos_x_3_6"):                                   # :192-199
            prb.createResidual(nm, Term("rel_pos", "rel_pos_gain", prm["rel_position_gain"]), nodes=st)
    prb.createResidual("min_qddot", Term("min_qddot", "min_qddot_gain", prm["min_qddot_gain"], 6 + (3 * nc if with_contact_states else 0)),
                       nodes=range(0, ns))                                                                                 # :200
    for i in range(nc):                                                                                                    # :201-204
        prb.createResidual("min_f" + str(i), Term("min_f", "min_f_gain", prm["min_f_gain"], 3), nodes=range(0, ns))
        prb.createResidual("f" + str(i) + "_active", Term("f_active", "force_switch_weight", prm["force_switch_weight"], 3),
                           nodes=range(0, ns))


class SRBDProblem:
    """prb.py:16-246 -- nx = 37, nu = 24, np = 19 with the launch file's contact_model=2, number_of_legs=2 (model "srbd37");
    nx = 61, nu = 48, np = 27 with the defaults in the code, contact_model=4, number_of_legs=2 (prb.py:39-40; model "srbd61",
    pass params={"contact_model": 4}).  number_of_legs=4 with contact_model=1 (four point feet, nc = 4) is the srbd37 layout
    without the relative-velocity constraints, which prb.py:166 only adds `if contact_model > 1`."""

    def createSRBDProblem(self, ns, T, robot: RobotModel | None = None, params: dict | None = None):
        robot = robot or RobotModel()
        prm = {**DEFAULT_PARAMS, **(params or {})}
        prb = Problem(ns)
        contact_model, number_of_legs = prm["contact_model"], prm["number_of_legs"]
        nc = number_of_legs * contact_model
        if (number_of_legs, contact_model) not in ((2, 2), (2, 4), (4, 1)):
            raise ValueError("analytic models exist for number_of_legs=2 with contact_model=2 (srbd37, the launch file's) or "
                             "contact_model=4 (srbd61, the default in prb.py:39), and for number_of_legs=4 with contact_model=1 (four "
                             "point feet: the srbd37 layout without the relative-velocity constraints of prb.py:166-170); the "
                             "problem indexes feet 0..3 (prb.py:153-154)")
        r = prb.createStateVariable("r", 3)                                   # prb.py:32
        o = prb.createStateVariable("o", 4)                                   # prb.py:33
        q = Aggregate(); q.addVariable(r); q.addVariable(o)
        c = {i: prb.createStateVariable("c" + str(i), 3) for i in range(nc)}   # prb.py:43-46
        rdot = prb.createStateVariable("rdot", 3)                             # prb.py:49
        w = prb.createStateVariable("w", 3)                                   # prb.py:50
        cdot = {i: prb.createStateVariable("cdot" + str(i), 3) for i in range(nc)}   # prb.py:56-59
        cddot, f = {}, {}
        for i in range(nc):                                                   # prb.py:66-68 (interleaved)
            cddot[i] = prb.createInputVariable("cddot" + str(i), 3)
            f[i] = prb.createInputVariable("f" + str(i), 3)
        rdot_ref = prb.createParameter("rdot_ref", 3)                         # prb.py:71
        w_ref = prb.createParameter("w_ref", 3)                               # prb.py:72
        prb.setDt(T / ns)                                                     # prb.py:110
        feet = np.asarray(robot.feet if nc == 4 else robot.feet8, dtype=float)
        otg = prb.createParameter("orientation_tracking_gain", 1)             # prb.py:143
        otg.assign(1e1)                                                       # prb.py:144
        c_ref, cdot_switch = {}, {}
        for i in range(nc):                                                   # prb.py:159-163
            c_ref[i] = prb.createParameter("c_ref" + str(i), 1)
            c_ref[i].assign(feet[i][2], nodes=range(0, ns + 1))
            cdot_switch[i] = prb.createParameter("cdot_switch" + str(i), 1)
            cdot_switch[i].assign(1.0, nodes=range(0, ns + 1))
        oref = prb.createParameter("oref", 4)                                 # prb.py:185
        oref.assign(quat_inverse(np.array([0.0, 0.0, 0.0, 1.0])))             # prb.py:186
        _declare_srbd_terms(prb, prm, ns, nc, contact_model, True)            # prb.py:166-204
        # the graphs only use contact points 0..3 (d_initial_1 / d_initial_2, prb.py:153-154), whatever nc is
        prb.setModel("srbd37" if nc == 4 else "srbd61", _consts(robot, prm, T / ns, feet[:4]))
        self.prb = prb
        self.initial_foot_position = {i: feet[i].copy() for i in range(nc)}
        self.com = np.asarray(robot.com, dtype=float)
        self.force_scaling = 1000.0
        self.m, self.I = robot.m, np.asarray(robot.I, dtype=float)
        self.f, self.c, self.cdot = f, c, cdot
        self.c_ref, self.w_ref, self.rdot_ref, self.oref = c_ref, w_ref, rdot_ref, oref
        self.orientation_tracking_gain, self.cdot_switch = otg, cdot_switch
        self.contact_model, self.nc = contact_model, nc
        return prb

    def getInitialState(self):                                                # prb.py:224-240
        return np.concatenate([self.com, [0.0, 0.0, 0.0, 1.0]] + [self.initial_foot_position[i] for i in range(self.nc)]
                              + [np.zeros(6 + 3 * self.nc)])

    def getStaticInput(self):
        # prb.py:242-246 writes four blocks with m g / force_scaling / 4 (its nc = 4): the weight shared by the nc contact points
        fz = self.m * 9.81 / self.force_scaling / self.nc
        return np.tile([0.0, 0.0, 0.0, 0.0, 0.0, fz], self.nc)


class SRBD13Problem:
    """Metric model: x = r|o|rdot|w, u = f_L|f_R,
    p = rdot_ref | w_ref | orientation_tracking_gain | oref | c0(3) | c1(3) | cdot_switch0 | cdot_switch1."""

    def createSRBD13Problem(self, ns, T, robot: RobotModel | None = None, params: dict | None = None):
        robot = robot or RobotModel()
        prm = {**DEFAULT_PARAMS, **(params or {}), "contact_model": 1, "number_of_legs": 2}
        prb = Problem(ns)
        r = prb.createStateVariable("r", 3)
        o = prb.createStateVariable("o", 4)
        rdot = prb.createStateVariable("rdot", 3)
        w = prb.createStateVariable("w", 3)
        f = {i: prb.createInputVariable("f" + str(i), 3) for i in range(2)}
        rdot_ref = prb.createParameter("rdot_ref", 3)
        w_ref = prb.createParameter("w_ref", 3)
        otg = prb.createParameter("orientation_tracking_gain", 1)
        otg.assign(1e1)
        oref = prb.createParameter("oref", 4)
        oref.assign(quat_inverse(np.array([0.0, 0.0, 0.0, 1.0])))
        feet4 = np.asarray(robot.feet, dtype=float)
        centers = [0.5 * (feet4[0] + feet4[1]), 0.5 * (feet4[2] + feet4[3])]   # one point contact per foot
        c = {}
        for i in range(2):
            c[i] = prb.createParameter("c" + str(i), 3)
            c[i].assign(centers[i])
        cdot_switch = {}
        for i in range(2):
            cdot_switch[i] = prb.createParameter("cdot_switch" + str(i), 1)
            cdot_switch[i].assign(1.0)
        prb.setDt(T / ns)
        _declare_srbd_terms(prb, prm, ns, 2, 1, False)
        prb.setModel("srbd13", _consts(robot, prm, T / ns, feet4))
        self.prb = prb
        self.initial_foot_position = {i: centers[i].copy() for i in range(2)}
        self.com = np.asarray(robot.com, dtype=float)
        self.force_scaling = 1000.0
        self.m, self.I = robot.m, np.asarray(robot.I, dtype=float)
        self.f, self.c, self.cdot = f, c, {0: None, 1: None}
        self.c_ref = {i: ParameterRow(c[i], 2) for i in range(2)}            # z row of the contact position
        self.w_ref, self.rdot_ref, self.oref = w_ref, rdot_ref, oref
        self.orientation_tracking_gain, self.cdot_switch = otg, cdot_switch
        self.contact_model, self.nc = 1, 2
        return prb

    def getInitialState(self):
        return np.concatenate([self.com, [0.0, 0.0, 0.0, 1.0], np.zeros(6)])

    def getStaticInput(self):
        fz = self.m * 9.81 / self.force_scaling / 2
        return np.array([0.0, 0.0, fz, 0.0, 0.0, fz])


class LIPProblem:
    """prb.py:248-441 -- nx = 30, nu = 15, np = 11."""

    def createLIPProblem(self, ns, T, robot: RobotModel | None = None, params: dict | None = None):
        robot = robot or RobotModel()
        prm = {**DEFAULT_PARAMS, **(params or {})}
        prb = Problem(ns)
        contact_model, number_of_legs = prm["contact_model"], prm["number_of_legs"]
        nc = number_of_legs * contact_model
        if nc != 4 or contact_model != 2:
            raise ValueError("the reference's LIP problem indexes feet 0..3 (prb.py:364-365)")
        r = prb.createStateVariable("r", 3)                                   # prb.py:264
        c = {i: prb.createStateVariable("c" + str(i), 3) for i in range(nc)}   # prb.py:273-276
        rdot = prb.createStateVariable("rdot", 3)                             # prb.py:279
        cdot = {i: prb.createStateVariable("cdot" + str(i), 3) for i in range(nc)}   # prb.py:284-287
        z = prb.createInputVariable("z", 3)                                   # prb.py:292
        cddot = {i: prb.createInputVariable("cddot" + str(i), 3) for i in range(nc)}   # prb.py:293-295
        rdot_ref = prb.createParameter("rdot_ref", 3)                         # prb.py:298
        prb.setDt(T / ns)                                                     # prb.py:329
        feet = np.asarray(robot.feet, dtype=float)
        c_ref, cdot_switch = {}, {}
        for i in range(nc):                                                   # prb.py:370-376
            c_ref[i] = prb.createParameter("c_ref" + str(i), 1)
            c_ref[i].assign(feet[i][2], nodes=range(0, ns + 1))
            cdot_switch[i] = prb.createParameter("cdot_switch" + str(i), 1)
            cdot_switch[i].assign(1.0, nodes=range(0, ns + 1))
        # prb.py:379-402, in the reference's order
        for i in range(1, contact_model):
            prb.createConstraint("relative_vel_left_" + str(i), Term("relative_vel", dim=2))
        for i in range(contact_model + 1, 2 * contact_model):
            prb.createConstraint("relative_vel_right_" + str(i), Term("relative_vel", dim=2))
        for i in range(nc):
            prb.createConstraint("cz_tracking" + str(i), Term("cz_tracking"))
            prb.createConstraint("cdotxy_tracking" + str(i), Term("cdotxy_tracking", dim=2))
        st = range(1, ns + 1)
        prb.createResidual("rz_tracking", Term("rz_tracking", "r_tracking_gain", prm["r_tracking_gain"]), nodes=st)          # :390
        prb.createResidual("rxy_tracking", Term("rxy_tracking", "r_tracking_gain", prm["r_tracking_gain"], 2), nodes=st)     # :391
        prb.createResidual("rdot_tracking", Term("rdot_tracking", "rdot_tracking_gain", prm["rdot_tracking_gain"], 3), nodes=st)   # :392
        prb.createResidual("zmp_tracking", Term("zmp_tracking", "zmp_tracking_gain", prm["zmp_tracking_gain"], 3), nodes=range(0, ns))   # :393
        for nm in ("rel_pos_y_1_4", "rel_pos_x_1_4", "rel_pos_y_3_6", "rel_pos_x_3_6"):                                      # :394-401
            prb.createResidual(nm, Term("rel_pos", "rel_pos_gain", prm["rel_position_gain"]), nodes=st)
        prb.createResidual("min_qddot", Term("min_qddot", "min_qddot_gain", prm["min_qddot_gain"], 15), nodes=range(0, ns))  # :402
        prb.setModel("lip30", _consts(robot, prm, T / ns, feet))
        self.prb = prb
        self.initial_foot_position = {i: feet[i].copy() for i in range(nc)}
        self.com = np.asarray(robot.com, dtype=float)
        self.force_scaling = 1000.0
        self.m = robot.m
        self.c, self.cdot, self.c_ref, self.cdot_switch = c, cdot, c_ref, cdot_switch
        self.contact_model, self.rdot_ref, self.nc = contact_model, rdot_ref, nc
        return prb

    def getInitialState(self):                                                # prb.py:420-433
        return np.concatenate([self.com] + [self.initial_foot_position[i] for i in range(self.nc)] + [np.zeros(15)])

    def getStaticInput(self):                                                 # prb.py:435-441
        return np.concatenate([[self.com[0], self.com[1], 0.0], np.zeros(12)])
