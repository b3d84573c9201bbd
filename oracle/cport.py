"""TEST INFRASTRUCTURE ONLY -- ctypes face of the plain-C oracle (oracle/c/sddp_oracle.c, built by oracle/Makefile)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "c", "sddp_oracle.c")):
            subprocess.run(["make", "-C", _HERE, "-s"], check=True)
        _lib = C.CDLL(_LIB)
    return _lib


def pack_consts(cst):
    """cst: oracle.models.RobotConsts"""
    return np.array([cst.m, *np.asarray(cst.I, dtype=float).reshape(-1), cst.com[2], cst.dt, cst.force_scaling,
                     cst.r_tracking_gain, cst.rdot_tracking_gain, cst.w_tracking_gain, cst.force_switch_weight,
                     cst.min_qddot_gain, cst.min_f_gain, float(cst.inertia_mode), cst.lever_sign,
                     cst.friction_cone_coefficient, cst.friction_barrier_weight, cst.friction_barrier_sharpness], dtype=np.float64)


def pack_opts(o):
    """o: oracle.ddp.DdpOptions"""
    return np.array([o.max_iters, o.alpha_0, o.alpha_converge_threshold, o.line_search_decrease_factor, o.beta,
                     o.cost_reduction_ths, o.mu0, float(o.initial_rollout), o.gap_tol, o.mu_min, o.mu_max, float(o.second_order)],
                    dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def solve_batch(cst, opts, x0, P, xs, us, threads=1):
    """-> xs [B,N+1,13], us [B,N,6], stats [B,7] = cost, iters, converged, alpha, gap, mu, status"""
    lib = load()
    B, N = us.shape[0], us.shape[1]
    xs = np.ascontiguousarray(xs, dtype=np.float64).copy()
    us = np.ascontiguousarray(us, dtype=np.float64).copy()
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    P = np.ascontiguousarray(P, dtype=np.float64)
    stats = np.zeros((B, 7))
    cp, op = pack_consts(cst), pack_opts(opts)
    lib.oracle_srbd13_solve_batch(_p(cp), C.c_int(N), C.c_int(B), _p(x0), _p(P), _p(xs), _p(us), _p(op), _p(stats), C.c_int(threads))
    return xs, us, stats


def eval_knot(cst, x, u, p, k, terminal):
    lib = load()
    f = np.zeros(13); F = np.zeros((13, 19)); H = np.zeros((19, 19)); g = np.zeros(19); L = np.zeros(1)
    cp = pack_consts(cst)
    x = np.ascontiguousarray(x, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64); p = np.ascontiguousarray(p, dtype=np.float64)
    lib.oracle_srbd13_eval(_p(cp), _p(x), _p(u), _p(p), C.c_int(k), C.c_int(int(terminal)), _p(f), _p(F), _p(H), _p(g), _p(L))
    return f, F, H, g, float(L[0])
