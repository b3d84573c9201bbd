"""TEST INFRASTRUCTURE ONLY -- ctypes face of the plain-C oracle (oracle/c/sddp_oracle.c + ddp_engine.inc, built by oracle/Makefile)."""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = [os.path.join(_HERE, "c", f) for f in ("sddp_oracle.c", "ddp_engine.inc", "srbd_cs.inc")]
_libs = {}
_variant = "off"          # which build the module-level functions use (see use_variant)

MODEL_IDS = {"srbd13": 0, "srbd37": 1, "lip30": 2, "srbd61": 3}
DIMS = {"srbd13": (13, 6, 19), "srbd37": (37, 24, 19), "lip30": (30, 15, 11), "srbd61": (61, 48, 27)}


def _host_tag():
    """The library is compiled with -march=native: one build per CPU type (a build from another machine may not run here)."""
    try:
        flags = [l for l in open("/proc/cpuinfo") if l.startswith(("flags", "model name"))][:2]
    except OSError:
        flags = []
    return hashlib.sha1("".join(flags).encode()).hexdigest()[:10]


def lib_path(variant="off"):
    return os.path.join(_HERE, "_build", f"liboracle-{_host_tag()}{'' if variant == 'off' else '-' + variant}.so")


def load(variant=None):
    """variant "off" (default): -ffp-contract=off; "fast": the same sources with gcc free to fuse multiply-adds (oracle/Makefile)."""
    variant = variant or _variant
    if variant not in ("off", "fast"):
        raise ValueError(variant)
    if variant not in _libs:
        path = lib_path(variant)
        deps = _SRC + [os.path.join(_HERE, "Makefile")]
        if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in deps):
            subprocess.run(["make", "-B", "-C", _HERE, "-s", f"LIB={os.path.relpath(path, _HERE)}", f"CONTRACT={variant}"], check=True)
        _libs[variant] = C.CDLL(path)
    return _libs[variant]


def pack_consts(cst, model=None):
    """cst: oracle.models.RobotConsts.  The packed record holds contact points 0..3 -- all the problem graphs use (d_initial_1/2,
    prb.py:153-154); srbd61's eight points live in cst.feet8, whose first four take that place."""
    feet = np.asarray(cst.feet8 if model == "srbd61" else cst.feet, dtype=float).reshape(-1)[:12]
    return np.array([cst.m, *np.asarray(cst.I, dtype=float).reshape(-1), cst.com[2], cst.dt, cst.force_scaling,
                     cst.r_tracking_gain, cst.rdot_tracking_gain, cst.w_tracking_gain, cst.force_switch_weight,
                     cst.min_qddot_gain, cst.min_f_gain, float(cst.inertia_mode), cst.lever_sign,
                     cst.friction_cone_coefficient, cst.friction_barrier_weight, cst.friction_barrier_sharpness,
                     cst.rel_pos_gain, cst.zmp_tracking_gain, cst.lip_height, *feet,
                     cst.bound_barrier_weight, cst.bound_barrier_sharpness, *_bounds64(cst.lower, -np.inf), *_bounds64(cst.upper, np.inf),
                     float(bool(getattr(cst, "relative_velocity_constraints", True))), *_extra_rows(cst, model)],
                    dtype=np.float64)


NXR = 8


def _extra_rows(cst, model):
    """xr_on, xr_n, then 8 x (kind, w, const, a[128]) (oracle/c/sddp_oracle.c unpack_consts)"""
    rows = list(getattr(cst, "extra_rows", None) or ())
    out = [float(bool(rows)), float(len(rows))]
    nz = sum(DIMS[model or "srbd13"][:2])
    for j in range(NXR):
        a = np.zeros(128)
        if j < len(rows):
            r = rows[j]
            a[:nz] = np.asarray(r["a"], dtype=float)
            out += [0.0 if r["kind"] == "state" else 1.0, float(r["w"]), float(r.get("const", 0.0)), *a]
        else:
            out += [0.0, 0.0, 0.0, *a]
    return out


def n_params(cst, model):
    """parameter columns per node: the model's own, + NXR reference columns when user rows are declared"""
    return DIMS[model][2] + (NXR if getattr(cst, "extra_rows", None) else 0)


def _bounds64(b, fill):
    out = np.full(64, fill)
    if b is not None:
        b = np.asarray(b, dtype=float).reshape(-1)
        out[:b.size] = b
    return out


def pack_opts(o, resume=None):
    """o: oracle.ddp.DdpOptions.  resume: None, or dict(rho, theta, closed, mu) -- the state an iteration carries over, to continue
    from another solve's iterate (ddp_engine.inc)."""
    r = resume or {}
    return np.array([o.max_iters, o.alpha_0, o.alpha_converge_threshold, o.line_search_decrease_factor, o.beta,
                     o.cost_reduction_ths, o.mu0, float(o.initial_rollout), o.gap_tol, o.mu_min, o.mu_max, float(o.second_order),
                     float(r.get("rho", 0.0)), float(r.get("theta", 0.0)), float(bool(r.get("closed", False))), float(r.get("mu", -1.0))],
                    dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def solve_batch(cst, opts, x0, P, xs, us, threads=1, model="srbd13", variant=None):
    """-> xs [B,N+1,nx], us [B,N,nu], stats [B,8] = cost, iters, converged, alpha, gap, mu, status, rho"""
    lib = load(variant)
    B, N = us.shape[0], us.shape[1]
    nx, nu, npar = DIMS[model]
    xs = np.ascontiguousarray(xs, dtype=np.float64).copy()
    us = np.ascontiguousarray(us, dtype=np.float64).copy()
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    P = np.ascontiguousarray(P, dtype=np.float64)
    npar = n_params(cst, model)
    assert xs.shape == (B, N + 1, nx) and us.shape == (B, N, nu) and x0.shape == (B, nx) and P.shape == (B, N + 1, npar)
    stats = np.zeros((B, 8))
    cp, op = pack_consts(cst, model), pack_opts(opts)
    rc = lib.oracle_solve_batch(C.c_int(MODEL_IDS[model]), _p(cp), C.c_int(N), C.c_int(B), _p(x0), _p(P), _p(xs), _p(us), _p(op),
                                _p(stats), C.c_int(threads))
    assert rc == 0
    return xs, us, stats


def eval_knot(cst, x, u, p, k, terminal, model="srbd13"):
    lib = load()
    nx, nu, _ = DIMS[model]
    nz = nx + nu
    f = np.zeros(nx); F = np.zeros((nx, nz)); H = np.zeros((nz, nz)); g = np.zeros(nz); L = np.zeros(1)
    cp = pack_consts(cst, model)
    x = np.ascontiguousarray(x, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64); p = np.ascontiguousarray(p, dtype=np.float64)
    rc = lib.oracle_eval(C.c_int(MODEL_IDS[model]), _p(cp), _p(x), _p(u), _p(p), C.c_int(k), C.c_int(int(terminal)), _p(f), _p(F), _p(H),
                         _p(g), _p(L))
    assert rc == 0
    return f, F, H, g, float(L[0])


TRACE_FIELDS = ("J", "A1", "B2", "rho", "gap", "expected", "alpha", "J_new", "theta", "mu", "tried", "slack")


def solve_trace(cst, opts, x0, P, xs, us, model="srbd13", variant=None, cap=256, resume=None):
    """One instance with its line-search record -> xs, us, stats[8], list of dicts (one per line search: TRACE_FIELDS plus
    "margin" / "J_cand", the Armijo margin  dphi - (beta pred + slack)  and the cost of every candidate tried, largest step first;
    a candidate is accepted iff its margin <= 0)."""
    lib = load(variant)
    N = us.shape[0]
    nx, nu, npar = DIMS[model]
    xs = np.ascontiguousarray(xs, dtype=np.float64).copy(); us = np.ascontiguousarray(us, dtype=np.float64).copy()
    x0 = np.ascontiguousarray(x0, dtype=np.float64); P = np.ascontiguousarray(P, dtype=np.float64)
    npar = n_params(cst, model)
    assert xs.shape == (N + 1, nx) and us.shape == (N, nu) and x0.shape == (nx,) and P.shape == (N + 1, npar)
    W = lib.oracle_trace_width()
    tr = np.zeros((cap, W)); stats = np.zeros(8)
    cp, op = pack_consts(cst, model), pack_opts(opts, resume)
    n = lib.oracle_solve_trace(C.c_int(MODEL_IDS[model]), _p(cp), C.c_int(N), _p(x0), _p(P), _p(xs), _p(us), _p(op), _p(stats), _p(tr),
                               C.c_int(cap))
    assert n >= 0
    out = []
    for r in tr[:n]:
        rec = dict(zip(TRACE_FIELDS, r[:12].tolist()))
        t = min(int(rec["tried"]), (W - 12) // 2)
        rec["margin"] = r[12:12 + 2 * t:2].copy(); rec["J_cand"] = r[13:13 + 2 * t:2].copy()
        out.append(rec)
    return xs, us, stats, out
