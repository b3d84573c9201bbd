"""TEST INFRASTRUCTURE ONLY -- numpy float64 restatement of the reference problem definitions.

PARITY UNPINNED upstream (see ``oracle/__init__.py``).  Every function cites the reference lines it follows.
Upstream formulas that live in absent third-party packages (Horizon ``utils.toRot``, ``kin_dyn.fSRBD``,
``utils.double_integrator_with_floating_base``, ``utils.quaterion_product``) are restated from their
published definitions and tagged UPSTREAM-UNVERIFIED.

Four models:
  * ``srbd13``  nx=13 nu=6  np=19 -- the BASELINE.json metric model
  * ``srbd37``  nx=37 nu=24 np=19 -- the reference-faithful SRBD problem (prb.py:16-246) at the launch file's contact_model = 2
  * ``srbd61``  nx=61 nu=48 np=27 -- the same problem at the code's default contact_model = 4 (prb.py:39-41)
  * ``lip30``   nx=30 nu=15 np=11 -- the reference LIP problem (prb.py:248-441)

Costs follow ddp.py:179-226: stage L_k = sum ||residual||^2 + 1e6 * sum ||eq-constraint||^2, terminal
L_N = sum ||residual||^2 (no constraints).  Everything is expressed as ONE stacked residual vector per node
(penalties enter as sqrt(1e6) * g) so cost = ||res||^2, gradient = 2 J^T res, Gauss-Newton Hessian = 2 J^T J.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

GRAVITY = 9.81                      # prb.py:243 (static input balances m*9.81), prb.py:317
CONSTRAINT_WEIGHT = 1e6             # ddp.py:181


# ----------------------------------------------------------------------------------------------------------
# constants (the reference reads these from a URDF + rosparam server that are absent: synthetic, fixed)
# ----------------------------------------------------------------------------------------------------------
@dataclass
class RobotConsts:
    """Model constants.  Gains are the rosparam defaults of prb.py:142-150 / prb.py:358-362."""
    m: float = 40.0                                                  # kindyn.mass()            prb.py:92
    I: np.ndarray = field(default_factory=lambda: np.array(          # CRBA(q)[3:6,3:6]         prb.py:94-95
        [[2.0, 0.03, -0.02], [0.03, 1.8, 0.04], [-0.02, 0.04, 0.6]]))
    com: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, 0.88]))   # prb.py:138-139
    # nc=4 line feet: left upper/lower, right upper/lower (launch:24-25)
    feet: np.ndarray = field(default_factory=lambda: np.array(
        [[0.08, 0.1, 0.0], [-0.08, 0.1, 0.0], [0.08, -0.1, 0.0], [-0.08, -0.1, 0.0]]))
    # nc=8 (contact_model = 4): the four corners of the left sole, then of the right one
    feet8: np.ndarray = field(default_factory=lambda: np.array(
        [[0.08, 0.13, 0.0], [-0.08, 0.13, 0.0], [0.08, 0.07, 0.0], [-0.08, 0.07, 0.0],
         [0.08, -0.07, 0.0], [-0.08, -0.07, 0.0], [0.08, -0.13, 0.0], [-0.08, -0.13, 0.0]]))
    dt: float = 0.05                                                 # T/ns, prb.py:110 ; wpg.py:20
    force_scaling: float = 1000.0                                    # prb.py:98
    r_tracking_gain: float = 1e3                                     # prb.py:142
    rdot_tracking_gain: float = 1e4                                  # prb.py:145
    w_tracking_gain: float = 1e4                                     # prb.py:146
    rel_pos_gain: float = 1e4                                        # prb.py:147
    force_switch_weight: float = 1e2                                 # prb.py:148
    min_qddot_gain: float = 1e0                                      # prb.py:149
    min_f_gain: float = 1e-2                                         # prb.py:150
    zmp_tracking_gain: float = 1e3                                   # prb.py:360 (LIP)
    lip_height: float = 0.88                                         # prb.py:317
    inertia_mode: int = 0       # 0: R o I o R^T element-wise (reference-faithful); 1: R I R^T
    lever_sign: float = 1.0     # +1: (c_i - r) x f_i (physical, default); -1: (r - c_i) x f_i (App. A.3)
    # Inequality handling the reference collects and then ignores (prb.py:172-177 friction cone, ddp.py:197-202 exponential
    # barrier, both commented out upstream): OFF by default (= reference behaviour).  weight > 0 adds, per contact force and
    # stage node, weight * sum_j exp(sharpness * a_j . f) over the 5 rows of the linearised cone A f <= 0
    friction_cone_coefficient: float = 0.8      # rosparam default, prb.py:174 ; linearised as mu / sqrt(2) (Horizon, unverified)
    friction_barrier_weight: float = 0.0        # ddp.py:182 exp_parameter would be 6.0
    friction_barrier_sharpness: float = 1.0
    # Variable bounds the reference turns into exponential barriers and then comments out (ddp.py:203-208): for every state and
    # input variable, at the stage nodes, exp(6 (v - upper)) + exp(6 (lower - v)).  OFF by default (weight 0; and prb.py sets no
    # bounds at all).  lower / upper: one value per entry of z = [x u] (None or +-inf: unbounded)
    bound_barrier_weight: float = 0.0
    bound_barrier_sharpness: float = 6.0        # exp_parameter, ddp.py:182
    lower: np.ndarray | None = None
    upper: np.ndarray | None = None
    # prb.py:166-170: the relative-velocity constraints inside a foot exist only `if contact_model > 1`.  False = the same state
    # layout with number_of_legs = 4 point feet (contact_model = 1, nc = 4): no such rows
    relative_velocity_constraints: bool = True
    # User-declared LINEAR residual rows (ddp.py:183-196 sums whatever residual the function container holds; the analytic models
    # take up to NXR = 8 extra rows of the form sqrt(w) (a . z - ref), z = [x u]): tuple of dict(a [nz], w, kind "state" (nodes
    # 1..N) | "stage" (nodes 0..N-1), const).  Row j's per-knot reference is parameter column np + j of a parameter vector that is
    # NXR columns wider than the model's own; ref = p[np + j] + const.
    extra_rows: tuple | None = None


# ----------------------------------------------------------------------------------------------------------
# quaternion helpers (x, y, z, w order; identity = 0,0,0,1 -- prb.py:226)
# ----------------------------------------------------------------------------------------------------------
def skew(a):
    return np.array([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]], dtype=np.result_type(a, float))


def quat_to_rot(q):
    """Horizon utils.toRot (prb.py:97).  UPSTREAM-UNVERIFIED: standard xyzw -> matrix, no normalisation."""
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.result_type(q, float))


def quat_to_rot_jac(q):
    """dR/dq_a for a in x,y,z,w (list of four 3x3)."""
    x, y, z, w = q
    t = np.result_type(q, float)
    return [
        np.array([[0, 2 * y, 2 * z], [2 * y, -4 * x, -2 * w], [2 * z, 2 * w, -4 * x]], dtype=t),
        np.array([[-4 * y, 2 * x, 2 * w], [2 * x, 0, 2 * z], [-2 * w, 2 * z, -4 * y]], dtype=t),
        np.array([[-4 * z, -2 * w, 2 * x], [2 * w, -4 * z, 2 * y], [2 * x, 2 * y, 0]], dtype=t),
        np.array([[0, -2 * z, 2 * y], [2 * z, 0, -2 * x], [-2 * y, 2 * x, 0]], dtype=t)]


def quat_mul(q, p):
    """Horizon utils.quaterion_product (prb.py:187).  UPSTREAM-UNVERIFIED: Hamilton product, xyzw."""
    qv, qw = q[0:3], q[3]
    pv, pw = p[0:3], p[3]
    return np.concatenate([qw * pv + pw * qv + np.cross(qv, pv), [qw * pw - qv @ pv]])


def quat_rate(o, w):
    """odot = 1/2 [w;0] (x) o  (LOCAL_WORLD_ALIGNED, prb.py:107-108).  UPSTREAM-UNVERIFIED."""
    ov, ow = o[0:3], o[3]
    return 0.5 * np.concatenate([ow * w + np.cross(w, ov), [-(w @ ov)]])


def quat_rate_jac(o, w):
    """(d odot/d o [4x4], d odot/d w [4x3])."""
    ov, ow = o[0:3], o[3]
    Jo = np.zeros((4, 4))
    Jo[0:3, 0:3] = 0.5 * skew(w)
    Jo[0:3, 3] = 0.5 * w
    Jo[3, 0:3] = -0.5 * w
    Jw = np.zeros((4, 3))
    Jw[0:3, :] = 0.5 * (ow * np.eye(3) - skew(ov))
    Jw[3, :] = -0.5 * ov
    return Jo, Jw


# ----------------------------------------------------------------------------------------------------------
# SRBD accelerations (Horizon kin_dyn.fSRBD, prb.py:99).  UPSTREAM-UNVERIFIED lever-arm sign -> flag.
# ----------------------------------------------------------------------------------------------------------
def world_inertia(cst: RobotConsts, o):
    """I_w and its derivative wrt the quaternion.  prb.py:99 writes ``w_R_b * (I / force_scaling) * w_R_b.T``
    where CasADi ``*`` is ELEMENT-WISE; inertia_mode=1 gives the physical R I R^T."""
    R = quat_to_rot(o)
    dR = quat_to_rot_jac(o)
    Is = np.asarray(cst.I) / cst.force_scaling
    if cst.inertia_mode == 0:
        M = R * Is * R.T
        dM = [Is * (dR[a] * R.T + R * dR[a].T) for a in range(4)]
    else:
        M = R @ Is @ R.T
        dM = [dR[a] @ Is @ R.T + R @ Is @ dR[a].T for a in range(4)]
    return M, dM


def world_inertia_hess(cst: RobotConsts, o):
    """d2 I_w / do_p do_q (4 x 4 list of 3x3).  R is quadratic in the quaternion, so dR/dq is linear and its derivative constant."""
    R = quat_to_rot(o)
    dR = quat_to_rot_jac(o)
    E = np.eye(4)
    d2R = [[quat_to_rot_jac(E[q])[p] for q in range(4)] for p in range(4)]        # d(dR/dq_p)/dq_q = dR/dq_p evaluated at e_q
    Is = np.asarray(cst.I) / cst.force_scaling
    if cst.inertia_mode == 0:
        return [[Is * (d2R[p][q] * R.T + dR[p] * dR[q].T + dR[q] * dR[p].T + R * d2R[p][q].T) for q in range(4)] for p in range(4)]
    return [[d2R[p][q] @ Is @ R.T + dR[p] @ Is @ dR[q].T + dR[q] @ Is @ dR[p].T + R @ Is @ d2R[p][q].T for q in range(4)]
            for p in range(4)]


def srbd_wdot_hess(cst: RobotConsts, r, o, w, cs, fs):
    """Second derivatives of wdot = I_w(o)^-1 (sum s (c_i - r) x f_i - w x I_w(o) w)  (prb.py:99; north_star "analytic second
    derivatives").  -> T [3, n, n] over the local variable order z = r(3) | o(4) | w(3) | c_0..(3 each) | f_0..(3 each), from
    differentiating I_w wdot = n twice:  d_a d_b wdot = I_w^-1 (d_a d_b n - d_a d_b I_w wdot - d_a I_w d_b wdot - d_b I_w d_a wdot)."""
    nc = len(cs)
    n = 10 + 6 * nc
    iO, iW = 3, 7
    iC = lambda i: 10 + 3 * i
    iF = lambda i: 10 + 3 * nc + 3 * i
    M, dM = world_inertia(cst, o)
    d2M = world_inertia_hess(cst, o)
    Minv = np.linalg.inv(M)
    J = srbd_acc_jac(cst, r, o, w, cs, fs)
    Jw = np.zeros((3, n))
    Jw[:, 0:3], Jw[:, iO:iO + 4], Jw[:, iW:iW + 3] = J["wdot_r"], J["wdot_o"], J["wdot_w"]
    for i in range(nc):
        Jw[:, iC(i):iC(i) + 3] = J["wdot_c"][i]
        Jw[:, iF(i):iF(i) + 3] = J["wdot_f"][i]
    _, wdot = srbd_acc(cst, r, o, w, cs, fs)
    E = np.eye(3)
    s = cst.lever_sign
    d2n = np.zeros((n, n, 3))
    for i in range(nc):
        for a in range(3):
            for b in range(3):
                v = s * np.cross(E[a], E[b])                       # d2 / dc_a df_b of s (c - r) x f ; d2 / dr_a df_b is its negative
                d2n[iC(i) + a, iF(i) + b] += v; d2n[iF(i) + b, iC(i) + a] += v
                d2n[a, iF(i) + b] -= v; d2n[iF(i) + b, a] -= v
    for a in range(3):
        for b in range(3):
            d2n[iW + a, iW + b] = -(np.cross(E[a], M @ E[b]) + np.cross(E[b], M @ E[a]))
        for q in range(4):
            v = -(np.cross(E[a], dM[q] @ w) + np.cross(w, dM[q] @ E[a]))
            d2n[iW + a, iO + q] = v; d2n[iO + q, iW + a] = v
    for p_ in range(4):
        for q in range(4):
            d2n[iO + p_, iO + q] = -np.cross(w, d2M[p_][q] @ w)
    T = np.zeros((3, n, n))
    for a in range(n):
        for b in range(n):
            v = d2n[a, b].copy()
            if iO <= a < iO + 4 and iO <= b < iO + 4:
                v -= d2M[a - iO][b - iO] @ wdot
            if iO <= a < iO + 4:
                v -= dM[a - iO] @ Jw[:, b]
            if iO <= b < iO + 4:
                v -= dM[b - iO] @ Jw[:, a]
            T[:, a, b] = Minv @ v
    return T, wdot


def quat_rate_hess_contract(vo):
    """sum_a vo_a d2 odot_a / do_b dw_c  [4 x 3]: odot is bilinear in (o, w), so d(d odot/d o)/dw_c = Jo evaluated at w = e_c."""
    S = np.zeros((4, 3))
    z4 = np.zeros(4)
    for c in range(3):
        Jo, _ = quat_rate_jac(z4, np.eye(3)[c])
        S[:, c] = vo @ Jo
    return S


def srbd_acc(cst: RobotConsts, r, o, w, cs, fs):
    """rddot, wdot for contact points ``cs`` and (scaled) forces ``fs`` (prb.py:92-99, App. A.3)."""
    ms = cst.m / cst.force_scaling
    rddot = np.array([0.0, 0.0, -GRAVITY]) + sum(fs) / ms
    M, _ = world_inertia(cst, o)
    tau = sum(cst.lever_sign * np.cross(c - r, f) for c, f in zip(cs, fs)) - np.cross(w, M @ w)
    try:
        wdot = np.linalg.solve(M, tau)
    except np.linalg.LinAlgError:            # a diverged line-search candidate (inf / nan state): its cost must come out non-finite,
        wdot = np.full(3, np.nan)            # not as an exception (the C oracle and the kernels divide and carry on)
    return rddot, wdot


def srbd_acc_jac(cst: RobotConsts, r, o, w, cs, fs):
    """Jacobians of (rddot, wdot): dict with wdot_r [3x3], wdot_o [3x4], wdot_w [3x3], wdot_c[i], wdot_f[i],
    rddot_f (scalar 1/ms: d rddot / d f_i = I/ms)."""
    ms = cst.m / cst.force_scaling
    M, dM = world_inertia(cst, o)
    try:
        Minv = np.linalg.inv(M)
    except np.linalg.LinAlgError:            # diverged line-search candidate: non-finite results, no exception (see srbd_acc)
        Minv = np.full((3, 3), np.nan)
    tau = sum(cst.lever_sign * np.cross(c - r, f) for c, f in zip(cs, fs)) - np.cross(w, M @ w)
    wdot = Minv @ tau
    s = cst.lever_sign
    out = {
        "rddot_f": 1.0 / ms,
        "wdot_r": Minv @ (s * sum(skew(f) for f in fs)),
        "wdot_w": Minv @ (skew(M @ w) - skew(w) @ M),
        "wdot_c": [Minv @ (-s * skew(f)) for f in fs],
        "wdot_f": [Minv @ (s * skew(c - r)) for c in cs],
    }
    Jo = np.zeros((3, 4))
    for a in range(4):
        Jo[:, a] = -Minv @ (dM[a] @ wdot + np.cross(w, dM[a] @ w))
    out["wdot_o"] = Jo
    return out


# ----------------------------------------------------------------------------------------------------------
# generic model interface
# ----------------------------------------------------------------------------------------------------------
class _Rows:
    """Accumulates residual rows and their Jacobians."""

    def __init__(self, nx, nu):
        self.nx, self.nu = nx, nu
        self.r, self.Jx, self.Ju = [], [], []

    def add(self, val, Jx=None, Ju=None):
        val = np.atleast_1d(np.asarray(val, dtype=float))
        n = val.shape[0]
        self.r.append(val)
        self.Jx.append(np.zeros((n, self.nx)) if Jx is None else np.asarray(Jx, dtype=float).reshape(n, self.nx))
        self.Ju.append(np.zeros((n, self.nu)) if Ju is None else np.asarray(Ju, dtype=float).reshape(n, self.nu))

    def stack(self):
        return np.concatenate(self.r), np.vstack(self.Jx), np.vstack(self.Ju)


class Model:
    """f: x+ = x + dt*xdot(x,u[,p])  (explicit Euler, ddp.py:228-230);
    residual(x,u,p,k): stacked residual vector of node k (u=None -> terminal node, ddp.py:216-226)."""
    name = "?"
    nx = nu = np_ = 0

    def __init__(self, cst: RobotConsts | None = None):
        self.cst = cst if cst is not None else RobotConsts()

    # --- to be provided -------------------------------------------------------------------------------
    def f(self, x, u, p):
        raise NotImplementedError

    def f_jac(self, x, u, p):
        raise NotImplementedError

    def residual_jac(self, x, u, p, k):
        """-> (res, Jx, Ju).  ``u is None`` marks the terminal node."""
        raise NotImplementedError

    # --- derived ------------------------------------------------------------------------------------------
    def residual(self, x, u, p, k):
        return self.residual_jac(x, u, p, k)[0]

    def cost(self, x, u, p, k):
        r = self.residual(x, u, p, k)
        return float(r @ r)

    def cost_derivs(self, x, u, p, k):
        """-> L, lx, lu, lxx, lux, luu (Gauss-Newton Hessian 2 J^T J)."""
        r, Jx, Ju = self.residual_jac(x, u, p, k)
        return float(r @ r), 2 * Jx.T @ r, 2 * Ju.T @ r, 2 * Jx.T @ Jx, 2 * Ju.T @ Jx, 2 * Ju.T @ Ju

    def second_order_ux(self, x, u, p, vp):
        """Exact second-order dynamics term used by the DDP sweep: the (u, x) block of sum_i vp_i d2 f_i / du dx restricted to
        the bilinear torque (c - r) x f of prb.py:99 (constant tensor).  Zero for linear models."""
        return np.zeros((self.nu, self.nx))

    def second_order_full(self, x, u, p, k, vp):
        """Full second-order correction of a stage node, [nz x nz] over z = [x u] (second_order = 2, "full DDP"):
        sum_i vp_i d2 f_i / dz dz   (the dynamics tensor contracted with v' = Vx+ + Vxx+ d)
        + (exact Hessian of L_k - its Gauss-Newton part 2 J^T J) = sum_j 2 res_j d2 res_j / dz dz.
        Zero for linear-quadratic models."""
        return np.zeros((self.nx + self.nu, self.nx + self.nu))

    def initial_state(self):
        raise NotImplementedError

    def static_input(self):
        raise NotImplementedError

    def default_params(self, N):
        raise NotImplementedError


# ----------------------------------------------------------------------------------------------------------
# srbd13 -- metric model: x = r|o|rdot|w, u = f_L|f_R, contacts are per-knot parameters
# ----------------------------------------------------------------------------------------------------------
class SRBD13(Model):
    name = "srbd13"
    nx, nu, np_ = 13, 6, 19
    R_, O_, RD_, W_ = slice(0, 3), slice(3, 7), slice(7, 10), slice(10, 13)
    # p = rdot_ref(3) | w_ref(3) | otg(1) | oref(4) | c_L(3) | c_R(3) | sw_L | sw_R
    P_RDREF, P_WREF, P_OTG, P_OREF = slice(0, 3), slice(3, 6), 6, slice(7, 11)
    P_C = (slice(11, 14), slice(14, 17))
    P_SW = (17, 18)

    def _split(self, x, u, p):
        cs = [p[self.P_C[0]], p[self.P_C[1]]]
        fs = [u[0:3], u[3:6]]
        return x[self.R_], x[self.O_], x[self.RD_], x[self.W_], cs, fs

    def f(self, x, u, p):
        r, o, rd, w, cs, fs = self._split(x, u, p)
        rddot, wdot = srbd_acc(self.cst, r, o, w, cs, fs)
        dt = self.cst.dt
        return np.concatenate([r + dt * rd, o + dt * quat_rate(o, w), rd + dt * rddot, w + dt * wdot])

    def f_jac(self, x, u, p):
        r, o, rd, w, cs, fs = self._split(x, u, p)
        dt = self.cst.dt
        J = srbd_acc_jac(self.cst, r, o, w, cs, fs)
        Jo, Jw = quat_rate_jac(o, w)
        fx = np.eye(13)
        fu = np.zeros((13, 6))
        fx[self.R_, self.RD_] += dt * np.eye(3)
        fx[self.O_, self.O_] += dt * Jo
        fx[self.O_, self.W_] += dt * Jw
        fx[self.W_, self.R_] += dt * J["wdot_r"]
        fx[self.W_, self.O_] += dt * J["wdot_o"]
        fx[self.W_, self.W_] += dt * J["wdot_w"]
        for i in range(2):
            fu[self.RD_, 3 * i:3 * i + 3] = dt * J["rddot_f"] * np.eye(3)
            fu[self.W_, 3 * i:3 * i + 3] = dt * J["wdot_f"][i]
        return fx, fu

    def residual_jac(self, x, u, p, k):
        c = self.cst
        rows = _Rows(13, 6)
        r, o, rd, w = x[self.R_], x[self.O_], x[self.RD_], x[self.W_]
        terminal = u is None
        if terminal or k >= 1:                                  # nodes 1..ns (prb.py:184-191)
            _srbd_state_rows(rows, c, r, o, rd, w, p[self.P_RDREF], p[self.P_WREF], p[self.P_OTG],
                             p[self.P_OREF], self.R_, self.O_, self.RD_, self.W_)
        if not terminal:                                        # nodes 0..ns-1 (prb.py:200-204)
            cs = [p[self.P_C[0]], p[self.P_C[1]]]
            fs = [u[0:3], u[3:6]]
            rddot, wdot = srbd_acc(c, r, o, w, cs, fs)
            J = srbd_acc_jac(c, r, o, w, cs, fs)
            g = np.sqrt(c.min_qddot_gain)
            Jx = np.zeros((6, 13))
            Ju = np.zeros((6, 6))
            Jx[3:6, self.R_] = J["wdot_r"]
            Jx[3:6, self.O_] = J["wdot_o"]
            Jx[3:6, self.W_] = J["wdot_w"]
            for i in range(2):
                Ju[0:3, 3 * i:3 * i + 3] = J["rddot_f"] * np.eye(3)
                Ju[3:6, 3 * i:3 * i + 3] = J["wdot_f"][i]
            rows.add(g * np.concatenate([rddot, wdot]), g * Jx, g * Ju)          # min_qddot  prb.py:200
            for i in range(2):
                _force_rows(rows, c, fs[i], p[self.P_SW[i]], 3 * i)
            _bound_rows(rows, c, x, u)
        return rows.stack()

    def second_order_ux(self, x, u, p, vp):
        M, _ = world_inertia(self.cst, x[self.O_])
        y = np.linalg.solve(M, self.cst.dt * vp[self.W_])
        S = np.zeros((6, 13))
        for i in range(2):
            S[3 * i:3 * i + 3, self.R_] = -self.cst.lever_sign * skew(y)      # d2 (y.((c-r) x f)) / df dr
        return S

    def second_order_full(self, x, u, p, k, vp):
        r, o, rd, w, cs, fs = self._split(x, u, p)
        S = _srbd_second_order_full(self.cst, 19, r, o, w, cs, fs, vp[self.O_], vp[self.W_], self.O_.start, self.W_.start,
                                    None, [13, 16])
        return S + np.diag(_bound_hess_extra(self.cst, x, u))

    def initial_state(self):
        return np.concatenate([self.cst.com, [0, 0, 0, 1.0], np.zeros(6)])        # prb.py:224-240 reduced

    def static_input(self):
        fz = self.cst.m * GRAVITY / self.cst.force_scaling / 2                    # prb.py:243 with 2 contacts
        return np.array([0, 0, fz, 0, 0, fz])

    def foot_centers(self):
        feet = np.asarray(self.cst.feet)
        return 0.5 * (feet[0] + feet[1]), 0.5 * (feet[2] + feet[3])

    def default_params(self, N):
        P = np.zeros((N + 1, 19))
        P[:, self.P_OTG] = 1e1                                                     # prb.py:144
        P[:, self.P_OREF] = [-0.0, -0.0, -0.0, 1.0]                                # prb.py:186
        cl, cr = self.foot_centers()
        P[:, self.P_C[0]] = cl
        P[:, self.P_C[1]] = cr
        P[:, self.P_SW[0]] = 1.0
        P[:, self.P_SW[1]] = 1.0                                                   # prb.py:163
        return P


def _srbd_second_order_full(cst, nz, r, o, w, cs, fs, vp_o, vp_w, o0, w0, c_idx, f_idx):
    """Shared by srbd13 / srbd37.  The only non-linear pieces are wdot (dynamics rows w, and the wdot rows of the min_qddot
    residual, prb.py:200) and the bilinear quaternion kinematics (prb.py:107-108):
        S = sum_m lam_m d2 wdot_m  +  dt * (v'_o . d2 odot),   lam = dt v'_w + 2 min_qddot_gain wdot.
    c_idx: state columns of the contact points (None: contacts are parameters), f_idx: z columns of the forces."""
    nc = len(cs)
    T, wdot = srbd_wdot_hess(cst, r, o, w, cs, fs)
    lam = cst.dt * vp_w + 2.0 * cst.min_qddot_gain * wdot
    Sl = np.tensordot(lam, T, axes=(0, 0))
    n = 10 + 6 * nc
    gl = np.full(n, -1)                                            # local -> global z index
    gl[0:3] = [0, 1, 2]
    gl[3:7] = o0 + np.arange(4)
    gl[7:10] = w0 + np.arange(3)
    for i in range(nc):
        if c_idx is not None:
            gl[10 + 3 * i:13 + 3 * i] = c_idx[i] + np.arange(3)
        gl[10 + 3 * nc + 3 * i:13 + 3 * nc + 3 * i] = f_idx[i] + np.arange(3)
    S = np.zeros((nz, nz))
    keep = gl >= 0
    S[np.ix_(gl[keep], gl[keep])] = Sl[np.ix_(keep, keep)]
    Q = cst.dt * quat_rate_hess_contract(vp_o)
    S[o0:o0 + 4, w0:w0 + 3] += Q
    S[w0:w0 + 3, o0:o0 + 4] += Q.T
    if cst.friction_barrier_weight > 0.0:
        # opt-in friction-cone barrier (_force_rows): cost w sum_j exp(s a_j.f) has the exact Hessian w s^2 e_j a_j a_j^T, its
        # residual form r_j = sqrt(w) exp(s a_j.f / 2) the Gauss-Newton Hessian (w s^2 / 2) e_j a_j a_j^T: the difference is the same again
        A = friction_cone_rows(cst.friction_cone_coefficient)
        for i in range(nc):
            e = cst.friction_barrier_weight * np.exp(cst.friction_barrier_sharpness * (A @ fs[i]))
            H = 0.5 * cst.friction_barrier_sharpness ** 2 * (A.T * e) @ A
            S[f_idx[i]:f_idx[i] + 3, f_idx[i]:f_idx[i] + 3] += H
    return S


def _srbd_state_rows(rows, c, r, o, rd, w, rdot_ref, w_ref, otg, oref, R_, O_, RD_, W_):
    nx = rows.nx
    J = np.zeros((1, nx)); J[0, R_.start + 2] = 1.0
    g = np.sqrt(c.r_tracking_gain)
    rows.add(g * (r[2] - c.com[2]), g * J)                                         # rz_tracking  prb.py:184
    e = quat_mul(o, oref)                                                          # prb.py:187
    Je = np.zeros((4, nx))
    Je[0:3, O_.start:O_.start + 3] = oref[3] * np.eye(3) - skew(oref[0:3])
    Je[0:3, O_.start + 3] = oref[0:3]
    Je[3, O_.start:O_.start + 3] = -oref[0:3]
    Je[3, O_.start + 3] = oref[3]
    rows.add(otg * e[0:3], otg * Je[0:3])                                          # o_tracking_xyz prb.py:188
    rows.add(otg * (e[3] - 1.0), otg * Je[3:4])                                    # o_tracking_w   prb.py:189
    J = np.zeros((3, nx)); J[:, RD_] = np.eye(3)
    g = np.sqrt(c.rdot_tracking_gain)
    rows.add(g * (rd - rdot_ref), g * J)                                           # rdot_tracking prb.py:190
    J = np.zeros((3, nx)); J[:, W_] = np.eye(3)
    g = np.sqrt(c.w_tracking_gain)
    rows.add(g * (w - w_ref), g * J)                                               # w_tracking    prb.py:191


def friction_cone_rows(mu):
    """A of the linearised friction cone A f <= 0 with the environment rotation = identity (prb.py:175-176; Horizon's
    kin_dyn.linearized_friction_cone is absent: inner pyramid mu/sqrt(2) + unilateral f_z >= 0, UPSTREAM-UNVERIFIED)."""
    ml = mu / np.sqrt(2.0)
    return np.array([[1.0, 0.0, -ml], [-1.0, 0.0, -ml], [0.0, 1.0, -ml], [0.0, -1.0, -ml], [0.0, 0.0, -1.0]])


def _force_rows(rows, c, f, sw, ucol):
    J = np.zeros((3, rows.nu)); J[:, ucol:ucol + 3] = np.eye(3)
    g = c.force_scaling * np.sqrt(c.min_f_gain)
    rows.add(g * f, None, g * J)                                                   # min_f_i     prb.py:202
    g = c.force_scaling * np.sqrt(c.force_switch_weight) * (1.0 - sw)
    rows.add(g * f, None, g * J)                                                   # f_i_active  prb.py:203-204
    if c.friction_barrier_weight > 0.0:            # exponential barrier as residuals r_j = sqrt(w) exp(s a_j.f / 2): r^2 = w exp(s g)
        A = friction_cone_rows(c.friction_cone_coefficient)
        r = np.sqrt(c.friction_barrier_weight) * np.exp(0.5 * c.friction_barrier_sharpness * (A @ f))
        Jf = (0.5 * c.friction_barrier_sharpness) * r[:, None] * A                 # d r_j / d f
        Ju = np.zeros((5, rows.nu)); Ju[:, ucol:ucol + 3] = Jf
        rows.add(r, None, Ju)


def _bound_rows(rows, c, x, u):
    """Opt-in exponential barrier on the bounds of the state and input variables (ddp.py:203-208, commented out upstream):
    cost w [exp(s (z_j - ub_j)) + exp(s (lb_j - z_j))] per bounded entry of z = [x u], written like the friction barrier as
    residuals r = sqrt(w) exp(s (z_j - ub_j) / 2) so that its Gauss-Newton Hessian (w s^2 / 2) e is of the common form.
    One row per finite bound, in the order j = 0..nz-1, upper before lower."""
    if not (c.bound_barrier_weight > 0.0):
        return
    nx, nu = rows.nx, rows.nu
    z = np.concatenate([x, u])
    lb = np.full(nx + nu, -np.inf) if c.lower is None else np.asarray(c.lower, dtype=float)[:nx + nu]
    ub = np.full(nx + nu, np.inf) if c.upper is None else np.asarray(c.upper, dtype=float)[:nx + nu]
    sw, hs = np.sqrt(c.bound_barrier_weight), 0.5 * c.bound_barrier_sharpness
    for j in range(nx + nu):
        for bound, sign in ((ub[j], 1.0), (lb[j], -1.0)):
            if not np.isfinite(bound):
                continue
            r = sw * np.exp(hs * sign * (z[j] - bound))
            Jx, Ju = np.zeros((1, nx)), np.zeros((1, nu))
            (Jx if j < nx else Ju)[0, j if j < nx else j - nx] = hs * sign * r
            rows.add(r, Jx, Ju)


def _bound_hess_extra(c, x, u):
    """exact minus Gauss-Newton Hessian of the bound barrier = its (diagonal) Gauss-Newton Hessian once more -> [nz]"""
    nz = x.shape[0] + u.shape[0]
    d = np.zeros(nz)
    if not (c.bound_barrier_weight > 0.0):
        return d
    z = np.concatenate([x, u])
    lb = np.full(nz, -np.inf) if c.lower is None else np.asarray(c.lower, dtype=float)[:nz]
    ub = np.full(nz, np.inf) if c.upper is None else np.asarray(c.upper, dtype=float)[:nz]
    with np.errstate(over="ignore"):
        e = np.where(np.isfinite(ub), np.exp(c.bound_barrier_sharpness * (z - ub)), 0.0) + \
            np.where(np.isfinite(lb), np.exp(c.bound_barrier_sharpness * (lb - z)), 0.0)
    return 0.5 * c.bound_barrier_weight * c.bound_barrier_sharpness ** 2 * e


def _contact_penalty_rows(rows, cs, cds, c_ref, sw, c_idx, cd_idx, contact_model):
    """Equality constraints as sqrt(1e6)-weighted residuals (ddp.py:195-196; prb.py:166-170, :179-181)."""
    nx = rows.nx
    g = np.sqrt(CONSTRAINT_WEIGHT)
    nc = len(cs)
    if contact_model > 1:
        for i in range(1, contact_model):                                           # relative_vel_left_i
            J = np.zeros((2, nx)); J[0, cd_idx[0]] = J[1, cd_idx[0] + 1] = 1; J[0, cd_idx[i]] = J[1, cd_idx[i] + 1] = -1
            rows.add(g * (cds[0][0:2] - cds[i][0:2]), g * J)
        for i in range(contact_model + 1, 2 * contact_model):                       # relative_vel_right_i
            b = contact_model
            J = np.zeros((2, nx)); J[0, cd_idx[b]] = J[1, cd_idx[b] + 1] = 1; J[0, cd_idx[i]] = J[1, cd_idx[i] + 1] = -1
            rows.add(g * (cds[b][0:2] - cds[i][0:2]), g * J)
    for i in range(nc):
        J = np.zeros((1, nx)); J[0, c_idx[i] + 2] = 1
        rows.add(g * (cs[i][2] - c_ref[i]), g * J)                                  # cz_tracking_i
        J = np.zeros((2, nx)); J[0, cd_idx[i]] = J[1, cd_idx[i] + 1] = sw[i]
        rows.add(g * sw[i] * cds[i][0:2], g * J)                                    # cdotxy_tracking_i


def _rel_pos_rows(rows, c, cs, c_idx, feet):
    """rel_pos_{y,x}_1_4 and _3_6 (prb.py:192-199) with d1 = p2-p0, d2 = p3-p1 (prb.py:153-154)."""
    nx = rows.nx
    g = np.sqrt(c.rel_pos_gain)
    d1 = -(feet[0][0:2] - feet[2][0:2])
    d2 = -(feet[1][0:2] - feet[3][0:2])
    for (a, b, d) in ((0, 2, d1), (1, 3, d2)):
        for comp in (1, 0):                                                         # y first, then x
            J = np.zeros((1, nx)); J[0, c_idx[a] + comp] = -1; J[0, c_idx[b] + comp] = 1
            rows.add(g * (-cs[a][comp] + cs[b][comp] - d[comp]), g * J)


# ----------------------------------------------------------------------------------------------------------
# srbd37 / srbd61 -- reference-faithful SRBD (prb.py:16-246), contacts are states; nc = number_of_legs * contact_model
# (prb.py:39-41): 4 (the launch file's contact_model = 2) and 8 (the default in the code, contact_model = 4)
# ----------------------------------------------------------------------------------------------------------
class SRBD37(Model):
    name = "srbd37"
    nc, contact_model = 4, 2                                                        # launch:16-17
    nx, nu, np_ = 37, 24, 19
    R_, O_, RD_, W_ = slice(0, 3), slice(3, 7), slice(19, 22), slice(22, 25)
    C_IDX = [7, 10, 13, 16]
    CD_IDX = [25, 28, 31, 34]
    # p = rdot_ref | w_ref | otg | (c_ref_i, sw_i) x nc | oref   (creation order)
    P_RDREF, P_WREF, P_OTG, P_OREF = slice(0, 3), slice(3, 6), 6, slice(15, 19)

    @staticmethod
    def p_cref(i):
        return 7 + 2 * i

    @staticmethod
    def p_sw(i):
        return 8 + 2 * i

    def feet(self):
        """contact points 0..nc-1 (prb.py:130-131)"""
        return np.asarray(self.cst.feet, dtype=float)

    def _split(self, x, u):
        nc = self.nc
        cs = [x[i:i + 3] for i in self.C_IDX]
        cds = [x[i:i + 3] for i in self.CD_IDX]
        cdd = [u[6 * i:6 * i + 3] for i in range(nc)]
        fs = [u[6 * i + 3:6 * i + 6] for i in range(nc)]                             # interleaved prb.py:66-68
        return x[self.R_], x[self.O_], x[self.RD_], x[self.W_], cs, cds, cdd, fs

    def f(self, x, u, p):
        r, o, rd, w, cs, cds, cdd, fs = self._split(x, u)
        rddot, wdot = srbd_acc(self.cst, r, o, w, cs, fs)
        xdot = np.concatenate([rd, quat_rate(o, w)] + cds + [rddot, wdot] + cdd)    # App. A.3
        return x + self.cst.dt * xdot

    def f_jac(self, x, u, p):
        r, o, rd, w, cs, cds, cdd, fs = self._split(x, u)
        dt = self.cst.dt
        J = srbd_acc_jac(self.cst, r, o, w, cs, fs)
        Jo, Jw = quat_rate_jac(o, w)
        A = np.zeros((self.nx, self.nx))
        B = np.zeros((self.nx, self.nu))
        A[self.R_, self.RD_] = np.eye(3)
        A[self.O_, self.O_] = Jo
        A[self.O_, self.W_] = Jw
        for i in range(self.nc):
            A[self.C_IDX[i]:self.C_IDX[i] + 3, self.CD_IDX[i]:self.CD_IDX[i] + 3] = np.eye(3)
            A[self.W_, self.C_IDX[i]:self.C_IDX[i] + 3] = J["wdot_c"][i]
            B[self.RD_, 6 * i + 3:6 * i + 6] = J["rddot_f"] * np.eye(3)
            B[self.W_, 6 * i + 3:6 * i + 6] = J["wdot_f"][i]
            B[self.CD_IDX[i]:self.CD_IDX[i] + 3, 6 * i:6 * i + 3] = np.eye(3)
        A[self.W_, self.R_] = J["wdot_r"]
        A[self.W_, self.O_] = J["wdot_o"]
        A[self.W_, self.W_] = J["wdot_w"]
        return np.eye(self.nx) + dt * A, dt * B

    def residual_jac(self, x, u, p, k):
        c = self.cst
        nc, nx, nu = self.nc, self.nx, self.nu
        rows = _Rows(nx, nu)
        terminal = u is None
        xs = x
        r, o, rd, w = xs[self.R_], xs[self.O_], xs[self.RD_], xs[self.W_]
        cs = [xs[i:i + 3] for i in self.C_IDX]
        cds = [xs[i:i + 3] for i in self.CD_IDX]
        if terminal or k >= 1:
            _srbd_state_rows(rows, c, r, o, rd, w, p[self.P_RDREF], p[self.P_WREF], p[self.P_OTG],
                             p[self.P_OREF], self.R_, self.O_, self.RD_, self.W_)
            _rel_pos_rows(rows, c, cs, self.C_IDX, self.feet())
        if not terminal:
            cdd = [u[6 * i:6 * i + 3] for i in range(nc)]
            fs = [u[6 * i + 3:6 * i + 6] for i in range(nc)]
            rddot, wdot = srbd_acc(c, r, o, w, cs, fs)
            J = srbd_acc_jac(c, r, o, w, cs, fs)
            g = np.sqrt(c.min_qddot_gain)
            Jx = np.zeros((6 + 3 * nc, nx))
            Ju = np.zeros((6 + 3 * nc, nu))
            Jx[3:6, self.R_] = J["wdot_r"]
            Jx[3:6, self.O_] = J["wdot_o"]
            Jx[3:6, self.W_] = J["wdot_w"]
            for i in range(nc):
                Jx[3:6, self.C_IDX[i]:self.C_IDX[i] + 3] = J["wdot_c"][i]
                Ju[0:3, 6 * i + 3:6 * i + 6] = J["rddot_f"] * np.eye(3)
                Ju[3:6, 6 * i + 3:6 * i + 6] = J["wdot_f"][i]
                Ju[6 + 3 * i:9 + 3 * i, 6 * i:6 * i + 3] = np.eye(3)
            rows.add(g * np.concatenate([rddot, wdot] + cdd), g * Jx, g * Ju)       # min_qddot prb.py:200
            for i in range(nc):
                _force_rows(rows, c, fs[i], p[self.p_sw(i)], 6 * i + 3)
            _contact_penalty_rows(rows, cs, cds, [p[self.p_cref(i)] for i in range(nc)],
                                  [p[self.p_sw(i)] for i in range(nc)], self.C_IDX, self.CD_IDX,
                                  self.contact_model if c.relative_velocity_constraints else 1)
            _bound_rows(rows, c, xs, u)
        return rows.stack()

    def second_order_ux(self, x, u, p, vp):
        M, _ = world_inertia(self.cst, x[self.O_])
        y = np.linalg.solve(M, self.cst.dt * vp[self.W_])
        S = np.zeros((self.nu, self.nx))
        for i in range(self.nc):
            S[6 * i + 3:6 * i + 6, self.R_] = -self.cst.lever_sign * skew(y)
            S[6 * i + 3:6 * i + 6, self.C_IDX[i]:self.C_IDX[i] + 3] = self.cst.lever_sign * skew(y)
        return S

    def second_order_full(self, x, u, p, k, vp):
        r, o, rd, w, cs, cds, cdd, fs = self._split(x, u)
        S = _srbd_second_order_full(self.cst, self.nx + self.nu, r, o, w, cs, fs, vp[self.O_], vp[self.W_], self.O_.start, self.W_.start,
                                    self.C_IDX, [self.nx + 6 * i + 3 for i in range(self.nc)])
        return S + np.diag(_bound_hess_extra(self.cst, x, u))

    def initial_state(self):
        # prb.py:224-240 (written out for 4 contact points there; the same pattern for nc: com, identity, feet, zero velocities)
        return np.concatenate([self.cst.com, [0, 0, 0, 1.0], self.feet().reshape(-1), np.zeros(6 + 3 * self.nc)])

    def static_input(self):
        # prb.py:242-246 writes m g / force_scaling / 4 for its 4 contact points: the weight shared by the contact points
        fz = self.cst.m * GRAVITY / self.cst.force_scaling / self.nc
        return np.tile([0, 0, 0, 0, 0, fz], self.nc)

    def default_params(self, N):
        P = np.zeros((N + 1, self.np_))
        P[:, self.P_OTG] = 1e1
        feet = self.feet()
        for i in range(self.nc):
            P[:, self.p_cref(i)] = feet[i][2]                                       # prb.py:161
            P[:, self.p_sw(i)] = 1.0                                                # prb.py:163
        P[:, self.P_OREF] = [-0.0, -0.0, -0.0, 1.0]
        return P


class SRBD61(SRBD37):
    """contact_model = 4, number_of_legs = 2 (the defaults of prb.py:39-40): nc = 8 -- contact points 0..3 on the left foot,
    4..7 on the right one.  Everything prb.py builds scales with nc except the rel_pos residuals, which name the contact points
    0, 2 and 1, 3 literally (prb.py:153-154, :192-199: with nc = 8 all four sit on the left foot; restated as written)."""
    name = "srbd61"
    nc, contact_model = 8, 4
    nx, nu, np_ = 61, 48, 27
    R_, O_, RD_, W_ = slice(0, 3), slice(3, 7), slice(31, 34), slice(34, 37)
    C_IDX = [7 + 3 * i for i in range(8)]
    CD_IDX = [37 + 3 * i for i in range(8)]
    P_RDREF, P_WREF, P_OTG, P_OREF = slice(0, 3), slice(3, 6), 6, slice(23, 27)

    def feet(self):
        return np.asarray(self.cst.feet8, dtype=float)


# ----------------------------------------------------------------------------------------------------------
# lip30 -- reference LIP problem (prb.py:248-441), linear dynamics + quadratic cost
# ----------------------------------------------------------------------------------------------------------
class LIP30(Model):
    name = "lip30"
    nx, nu, np_ = 30, 15, 11
    nc, contact_model = 4, 2
    R_, RD_ = slice(0, 3), slice(15, 18)
    C_IDX = [3, 6, 9, 12]
    CD_IDX = [18, 21, 24, 27]
    P_RDREF = slice(0, 3)

    @staticmethod
    def p_cref(i):
        return 3 + 2 * i

    @staticmethod
    def p_sw(i):
        return 4 + 2 * i

    def _AB(self):
        eta2 = GRAVITY / self.cst.lip_height                                        # prb.py:317
        A = np.zeros((30, 30)); B = np.zeros((30, 15))
        A[0:15, 15:30] = np.eye(15)                                                 # qdot
        A[self.RD_, self.R_] = eta2 * np.eye(3)                                     # rddot = eta2 (r - z) - g
        B[self.RD_, 0:3] = -eta2 * np.eye(3)
        for i in range(4):
            B[self.CD_IDX[i]:self.CD_IDX[i] + 3, 3 + 3 * i:6 + 3 * i] = np.eye(3)
        b = np.zeros(30); b[self.RD_.start + 2] = -GRAVITY                          # prb.py:318
        return A, B, b

    def f(self, x, u, p):
        A, B, b = self._AB()
        return x + self.cst.dt * (A @ x + B @ u + b)

    def f_jac(self, x, u, p):
        A, B, _ = self._AB()
        return np.eye(30) + self.cst.dt * A, self.cst.dt * B

    def residual_jac(self, x, u, p, k):
        c = self.cst
        rows = _Rows(30, 15)
        terminal = u is None
        r, rd = x[self.R_], x[self.RD_]
        cs = [x[i:i + 3] for i in self.C_IDX]
        cds = [x[i:i + 3] for i in self.CD_IDX]
        mean_c = 0.25 * (cs[0] + cs[1] + cs[2] + cs[3])
        Jmean = np.zeros((3, 30))
        for i in self.C_IDX:
            Jmean[:, i:i + 3] = 0.25 * np.eye(3)
        if terminal or k >= 1:
            g = np.sqrt(c.r_tracking_gain)
            J = np.zeros((1, 30)); J[0, 2] = 1
            rows.add(g * (r[2] - c.com[2]), g * J)                                  # rz_tracking prb.py:390
            J = np.zeros((2, 30)); J[:, 0:2] = np.eye(2); J -= Jmean[0:2]
            rows.add(g * (r[0:2] - mean_c[0:2]), g * J)                             # rxy_tracking prb.py:391
            g = np.sqrt(c.rdot_tracking_gain)
            J = np.zeros((3, 30)); J[:, self.RD_] = np.eye(3)
            rows.add(g * (rd - p[self.P_RDREF]), g * J)                             # rdot_tracking prb.py:392
        if not terminal:
            z = u[0:3]
            g = np.sqrt(c.zmp_tracking_gain)
            Ju = np.zeros((3, 15)); Ju[:, 0:3] = np.eye(3)
            rows.add(g * (z - mean_c), -g * Jmean, g * Ju)                          # zmp_tracking prb.py:393
        if terminal or k >= 1:
            _rel_pos_rows(rows, c, cs, self.C_IDX, np.asarray(c.feet))              # prb.py:394-401
        if not terminal:
            eta2 = GRAVITY / c.lip_height
            g = np.sqrt(c.min_qddot_gain)
            rddot = eta2 * (r - u[0:3]) - np.array([0, 0, GRAVITY])
            Jx = np.zeros((15, 30)); Ju = np.zeros((15, 15))
            Jx[0:3, self.R_] = eta2 * np.eye(3)
            Ju[0:3, 0:3] = -eta2 * np.eye(3)
            Ju[3:15, 3:15] = np.eye(12)
            rows.add(g * np.concatenate([rddot, u[3:15]]), g * Jx, g * Ju)          # min_qddot prb.py:402
            _contact_penalty_rows(rows, cs, cds, [p[self.p_cref(i)] for i in range(4)],
                                  [p[self.p_sw(i)] for i in range(4)], self.C_IDX, self.CD_IDX,
                                  self.contact_model if c.relative_velocity_constraints else 1)
        return rows.stack()

    def initial_state(self):
        feet = np.asarray(self.cst.feet)
        return np.concatenate([self.cst.com, feet.reshape(-1), np.zeros(15)])       # prb.py:420-433

    def static_input(self):
        return np.concatenate([[self.cst.com[0], self.cst.com[1], 0.0], np.zeros(12)])   # prb.py:435-441

    def default_params(self, N):
        P = np.zeros((N + 1, 11))
        feet = np.asarray(self.cst.feet)
        for i in range(4):
            P[:, self.p_cref(i)] = feet[i][2]
            P[:, self.p_sw(i)] = 1.0
        return P


MODELS = {"srbd13": SRBD13, "srbd37": SRBD37, "lip30": LIP30, "srbd61": SRBD61}


NXR = 8      # extra linear residual rows an "_x" build carries (= extra parameter columns)


class WithLinearRows(Model):
    """A model plus user-declared linear residual rows (RobotConsts.extra_rows): the parameter vector is NXR columns wider, the
    last NXR columns are the rows' per-knot references.  Dynamics untouched."""

    def __init__(self, base: Model):
        self.base, self.cst = base, base.cst
        self.name = base.name
        self.nx, self.nu, self.npb = base.nx, base.nu, base.np_
        self.np_ = base.np_ + NXR
        rows = list(self.cst.extra_rows or ())
        if len(rows) > NXR:
            raise ValueError(f"at most {NXR} extra rows")
        for r in rows:
            a = np.asarray(r["a"], dtype=float)
            if a.shape != (self.nx + self.nu,) or r["kind"] not in ("state", "stage") or not (r["w"] >= 0.0):
                raise ValueError("extra row: a [nx + nu], w >= 0, kind 'state' | 'stage'")
            if r["kind"] == "state" and np.any(a[self.nx:] != 0.0):
                raise ValueError("a 'state' row (nodes 1..N, terminal node included) cannot touch the inputs")
        self.rows = rows

    def f(self, x, u, p):
        return self.base.f(x, u, p[:self.npb])

    def f_jac(self, x, u, p):
        return self.base.f_jac(x, u, p[:self.npb])

    def residual_jac(self, x, u, p, k):
        r, Jx, Ju = self.base.residual_jac(x, u, p[:self.npb], k)
        rr, jx, ju = [r], [Jx], [Ju]
        for j, row in enumerate(self.rows):
            active = (k >= 1) if row["kind"] == "state" else (u is not None)
            if not active:
                continue
            a = np.asarray(row["a"], dtype=float)
            g = np.sqrt(row["w"])
            z = np.concatenate([x, u if u is not None else np.zeros(self.nu)])
            rr.append(np.array([g * (a @ z - p[self.npb + j] - row.get("const", 0.0))]))
            jx.append(g * a[None, :self.nx]); ju.append(g * a[None, self.nx:])
        return np.concatenate(rr), np.vstack(jx), np.vstack(ju)

    def second_order_ux(self, x, u, p, vp):
        return self.base.second_order_ux(x, u, p[:self.npb], vp)

    def second_order_full(self, x, u, p, k, vp):
        return self.base.second_order_full(x, u, p[:self.npb], k, vp)

    def initial_state(self):
        return self.base.initial_state()

    def static_input(self):
        return self.base.static_input()

    def default_params(self, N):
        P = self.base.default_params(N)
        return np.hstack([P, np.zeros((P.shape[0], NXR))])


def make_model(name: str, cst: RobotConsts | None = None) -> Model:
    m = MODELS[name](cst)
    return WithLinearRows(m) if (cst is not None and cst.extra_rows) else m
