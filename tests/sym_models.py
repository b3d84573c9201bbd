"""Independent sympy restatement of SURVEY.md App. A (dynamics + residuals), used ONLY to pin the oracle's
hand-written Jacobians by symbolic differentiation.  Written from the equations, not from oracle/models.py."""
import functools

import numpy as np
import sympy as sp

G = sp.Float(9.81)


def _toRot(q):
    x, y, z, w = q
    return sp.Matrix([[1 - 2 * (y**2 + z**2), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x**2 + z**2), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x**2 + y**2)]])


def _qmul(q, p):
    qv, qw = sp.Matrix(q[0:3]), q[3]
    pv, pw = sp.Matrix(p[0:3]), p[3]
    v = qw * pv + pw * qv + qv.cross(pv)
    return sp.Matrix([v[0], v[1], v[2], qw * pw - qv.dot(pv)])


def _srbd(cst, r, o, w, cs, fs):
    s = cst.force_scaling
    ms = cst.m / s
    Is = sp.Matrix(np.asarray(cst.I) / s)
    R = _toRot(o)
    if cst.inertia_mode == 0:
        M = sp.Matrix(3, 3, lambda i, j: R[i, j] * Is[i, j] * R[j, i])
    else:
        M = R * Is * R.T
    rddot = sp.Matrix([0, 0, -G])
    tau = sp.zeros(3, 1)
    for c, f in zip(cs, fs):
        rddot += f / ms
        tau += cst.lever_sign * (c - r).cross(f)
    tau -= w.cross(M * w)
    wdot = (M.adjugate() / M.det()) * tau
    odot = sp.Rational(1, 2) * _qmul(sp.Matrix([w[0], w[1], w[2], 0]), o)
    return rddot, wdot, odot


def _state_res(cst, r, o, rd, w, rdot_ref, w_ref, otg, oref):
    e = _qmul(o, oref)
    return [sp.sqrt(cst.r_tracking_gain) * (r[2] - cst.com[2]),
            otg * e[0], otg * e[1], otg * e[2], otg * (e[3] - 1),
            *(sp.sqrt(cst.rdot_tracking_gain) * (rd - rdot_ref)),
            *(sp.sqrt(cst.w_tracking_gain) * (w - w_ref))]


def _force_res(cst, f, sw):
    g1 = cst.force_scaling * sp.sqrt(cst.min_f_gain)
    g2 = cst.force_scaling * sp.sqrt(cst.force_switch_weight)
    return [*(g1 * f), *(g2 * (1 - sw) * f)]


def _penalties(cs, cds, cref, sw, rel_vel=True):
    """prb.py:166-170 (relative velocities inside a foot: contact_model = nc / 2 points per foot; none with contact_model = 1,
    i.e. number_of_legs = 4 point feet: rel_vel False) and :179-181"""
    g = sp.sqrt(1e6)
    nc = len(cs)
    cm = nc // 2 if rel_vel else 1
    out = []
    for i in range(1, cm):
        out += [*(g * (cds[0][0:2, 0] - cds[i][0:2, 0]))]
    for i in range(cm + 1, 2 * cm):
        out += [*(g * (cds[cm][0:2, 0] - cds[i][0:2, 0]))]
    for i in range(nc):
        out.append(g * (cs[i][2] - cref[i]))
        out += [g * sw[i] * cds[i][0], g * sw[i] * cds[i][1]]
    return out


def _relpos(cst, cs):
    feet = np.asarray(cst.feet if len(cs) == 4 else cst.feet8)
    d1 = feet[2][0:2] - feet[0][0:2]
    d2 = feet[3][0:2] - feet[1][0:2]
    g = sp.sqrt(cst.rel_pos_gain)
    return [g * (-cs[0][1] + cs[2][1] - d1[1]), g * (-cs[0][0] + cs[2][0] - d1[0]),
            g * (-cs[1][1] + cs[3][1] - d2[1]), g * (-cs[1][0] + cs[3][0] - d2[0])]


def _build(name, cst):
    if name == "srbd13":
        nx, nu, npar = 13, 6, 19
    elif name == "srbd37":
        nx, nu, npar = 37, 24, 19
    elif name == "srbd61":
        nx, nu, npar = 61, 48, 27
    else:
        nx, nu, npar = 30, 15, 11
    rows = list(getattr(cst, "extra_rows", None) or ())      # user-declared linear residual rows: 8 more parameter columns
    npb = npar
    if rows:
        npar += 8
    x = sp.Matrix(sp.symbols(f"x0:{nx}"))
    u = sp.Matrix(sp.symbols(f"u0:{nu}"))
    p = sp.Matrix(sp.symbols(f"p0:{npar}"))
    dt = cst.dt
    if name == "srbd13":
        r, o, rd, w = x[0:3, 0], x[3:7, 0], x[7:10, 0], x[10:13, 0]
        cs = [p[11:14, 0], p[14:17, 0]]
        fs = [u[0:3, 0], u[3:6, 0]]
        rddot, wdot, odot = _srbd(cst, r, o, w, cs, fs)
        xdot = sp.Matrix([*rd, *odot, *rddot, *wdot])
        sres = _state_res(cst, r, o, rd, w, p[0:3, 0], p[3:6, 0], p[6], p[7:11, 0])
        ires = [*(sp.sqrt(cst.min_qddot_gain) * sp.Matrix([*rddot, *wdot]))]
        for i in range(2):
            ires += _force_res(cst, fs[i], p[17 + i])
    elif name in ("srbd37", "srbd61"):
        # x = r | o | c_i | rdot | w | cdot_i ; u = (cddot_i, f_i) interleaved ; p = rdot_ref | w_ref | otg | (c_ref_i, sw_i) | oref
        nc = 4 if name == "srbd37" else 8
        rd0 = 7 + 3 * nc
        r, o, rd, w = x[0:3, 0], x[3:7, 0], x[rd0:rd0 + 3, 0], x[rd0 + 3:rd0 + 6, 0]
        cs = [x[7 + 3 * i:10 + 3 * i, 0] for i in range(nc)]
        cds = [x[rd0 + 6 + 3 * i:rd0 + 9 + 3 * i, 0] for i in range(nc)]
        cdd = [u[6 * i:6 * i + 3, 0] for i in range(nc)]
        fs = [u[6 * i + 3:6 * i + 6, 0] for i in range(nc)]
        rddot, wdot, odot = _srbd(cst, r, o, w, cs, fs)
        xdot = sp.Matrix([*rd, *odot, *[e for c in cds for e in c], *rddot, *wdot, *[e for c in cdd for e in c]])
        sres = _state_res(cst, r, o, rd, w, p[0:3, 0], p[3:6, 0], p[6], p[7 + 2 * nc:11 + 2 * nc, 0]) + _relpos(cst, cs)
        ires = [*(sp.sqrt(cst.min_qddot_gain) * sp.Matrix([*rddot, *wdot, *[e for c in cdd for e in c]]))]
        for i in range(nc):
            ires += _force_res(cst, fs[i], p[8 + 2 * i])
        ires += _penalties(cs, cds, [p[7 + 2 * i] for i in range(nc)], [p[8 + 2 * i] for i in range(nc)],
                           getattr(cst, "relative_velocity_constraints", True))
    else:
        r, rd = x[0:3, 0], x[15:18, 0]
        cs = [x[3 + 3 * i:6 + 3 * i, 0] for i in range(4)]
        cds = [x[18 + 3 * i:21 + 3 * i, 0] for i in range(4)]
        z = u[0:3, 0]
        cdd = [u[3 + 3 * i:6 + 3 * i, 0] for i in range(4)]
        eta2 = 9.81 / cst.lip_height
        rddot = eta2 * (r - z) - sp.Matrix([0, 0, G])
        xdot = sp.Matrix([*rd, *cds[0], *cds[1], *cds[2], *cds[3], *rddot, *cdd[0], *cdd[1], *cdd[2], *cdd[3]])
        mean_c = (cs[0] + cs[1] + cs[2] + cs[3]) / 4
        sres = [sp.sqrt(cst.r_tracking_gain) * (r[2] - cst.com[2]),
                *(sp.sqrt(cst.r_tracking_gain) * (r[0:2, 0] - mean_c[0:2, 0])),
                *(sp.sqrt(cst.rdot_tracking_gain) * (rd - p[0:3, 0]))] + _relpos(cst, cs)
        ires = [*(sp.sqrt(cst.zmp_tracking_gain) * (z - mean_c)),
                *(sp.sqrt(cst.min_qddot_gain) * sp.Matrix([*rddot, *cdd[0], *cdd[1], *cdd[2], *cdd[3]]))]
        ires += _penalties(cs, cds, [p[3 + 2 * i] for i in range(4)], [p[4 + 2 * i] for i in range(4)],
                           getattr(cst, "relative_velocity_constraints", True))
    z_all = sp.Matrix([*x, *u])
    for j, r in enumerate(rows):                             # sqrt(w) (a . z - (p[np + j] + const)): state rows (nodes 1..N) or stage rows
        a = np.asarray(r["a"], dtype=float)
        e = sp.sqrt(r["w"]) * (sum(float(a[i]) * z_all[i] for i in range(nx + nu) if a[i] != 0.0) - p[npb + j] - float(r.get("const", 0.0)))
        (sres if r["kind"] == "state" else ires).append(e)
    f = x + dt * xdot
    out = {}
    args = (list(x), list(u), list(p))
    lam = lambda e: sp.lambdify(args, e, "numpy", cse=True)
    out["f"] = lam(f)
    out["F"] = lam(f.jacobian(z_all))
    # residual vectors and their SYMBOLIC Jacobians; costs / gradients / GN Hessians are assembled numerically
    ires_m, sres_m = sp.Matrix(ires), sp.Matrix(sres)
    out["ires"], out["sres"] = lam(ires_m), lam(sres_m)
    out["Ji"], out["Js"] = lam(ires_m.jacobian(z_all)), lam(sres_m.jacobian(z_all))
    out["nx"] = nx
    out["_sym"] = (x, u, p, f, ires_m, sres_m, z_all)
    return out


@functools.lru_cache(maxsize=None)
def second_order_symbolic(name, inertia_mode=0, lever_sign=1.0, rel_vel=True, rows_key=None):
    """lambda (x, u, p, vp) -> Hessian_z[vp.f + L_k] - 2 J^T J  (stage node k >= 1: input and state residuals), symbolic.
    rows_key: key into EXTRA_ROWS (user-declared linear rows; hashable for the cache)."""
    from oracle.models import RobotConsts
    sym = _build(name, RobotConsts(inertia_mode=inertia_mode, lever_sign=lever_sign, relative_velocity_constraints=rel_vel,
                                   extra_rows=EXTRA_ROWS.get(rows_key)))
    x, u, p, f, ires_m, sres_m, z_all = sym["_sym"]
    vp = sp.Matrix(sp.symbols(f"v0:{len(x)}"))
    res = sp.Matrix([*ires_m, *sres_m])
    # only the non-linear rows contribute: sum_i vp_i Hess(f_i) + sum_j 2 res_j Hess(res_j)
    H = sp.zeros(len(z_all), len(z_all))
    zl, zset = list(z_all), set(z_all)

    def add_hessian(weight, e):
        """H += weight * Hessian_z(e): upper triangle mirrored, second derivatives only of gradient entries that still depend on z
        (most rows are linear: sp.hessian would differentiate all nz^2 entries of every one of them)"""
        for a, za in enumerate(zl):
            ga = sp.diff(e, za)
            if not (ga.free_symbols & zset):
                continue
            for b in range(a, len(zl)):
                h = sp.diff(ga, zl[b])
                if h != 0:
                    H[a, b] += weight * h
                    if b != a:
                        H[b, a] += weight * h

    for i in range(len(x)):
        add_hessian(vp[i], f[i])
    for j in range(len(res)):
        add_hessian(2 * res[j], res[j])
    return sp.lambdify((list(x), list(u), list(p), list(vp)), H, "numpy", cse=True)

EXTRA_ROWS = {}      # rows_key -> tuple of row dicts (registered by the caller: lru_cache needs hashable arguments)


@functools.lru_cache(maxsize=None)
def symbolic(name, inertia_mode=0, lever_sign=1.0, rel_vel=True, rows_key=None):   # the consts object is returned so that callers can key caches on it
    from oracle.models import RobotConsts
    cst = RobotConsts(inertia_mode=inertia_mode, lever_sign=lever_sign, relative_velocity_constraints=rel_vel,
                      extra_rows=EXTRA_ROWS.get(rows_key))
    return _build(name, cst), cst


def cost_terms(sym, key, x, u, p):
    """(L, gradient wrt [x;u], GN Hessian) of node class key in {'0','k','N'} from the symbolic residuals."""
    nx = sym["nx"]
    ri = np.asarray(sym["ires"](x, u, p), dtype=float).reshape(-1)
    rs = np.asarray(sym["sres"](x, u, p), dtype=float).reshape(-1)
    Ji = np.asarray(sym["Ji"](x, u, p), dtype=float)
    Js = np.asarray(sym["Js"](x, u, p), dtype=float)
    if key == "0":
        r, J = ri, Ji
    elif key == "k":
        r, J = np.concatenate([ri, rs]), np.vstack([Ji, Js])
    else:
        r, J = rs, Js[:, :nx]
    return float(r @ r), 2 * J.T @ r, 2 * J.T @ J
