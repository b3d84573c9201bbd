"""ctypes binding of the C ABI declared in ``include/sddp.h`` (``libsddp_hip.so``).

This is the thin Python<->HIP boundary that stands where the reference's ``import pyddp`` stands
(reference python/ddp.py:1).  There is NO CPU fallback: if the library is missing or no HIP device is visible,
the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("SDDP_LIB", os.path.join(_HERE, "libsddp_hip.so"))   # SDDP_LIB: diagnostic builds only
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")

MODEL_IDS = {"srbd13": 0, "srbd37": 1, "lip30": 2}


class SddpOptions(C.Structure):
    _fields_ = [("max_iters", C.c_int), ("alpha_0", C.c_double), ("alpha_converge_threshold", C.c_double),
                ("line_search_decrease_factor", C.c_double), ("beta", C.c_double), ("cost_reduction_ths", C.c_double),
                ("mu0", C.c_double), ("initial_rollout", C.c_int), ("gap_tol", C.c_double), ("mu_min", C.c_double),
                ("mu_max", C.c_double), ("second_order", C.c_int), ("waves_per_simd", C.c_int), ("queue_order", C.c_int),
                ("max_slots", C.c_int)]


class SddpModelConsts(C.Structure):
    _fields_ = [("m", C.c_double), ("I", C.c_double * 9), ("com", C.c_double * 3), ("feet", C.c_double * 12),
                ("dt", C.c_double), ("force_scaling", C.c_double),
                ("r_tracking_gain", C.c_double), ("rdot_tracking_gain", C.c_double), ("w_tracking_gain", C.c_double),
                ("rel_pos_gain", C.c_double), ("force_switch_weight", C.c_double), ("min_qddot_gain", C.c_double),
                ("min_f_gain", C.c_double), ("zmp_tracking_gain", C.c_double), ("lip_height", C.c_double),
                ("inertia_mode", C.c_int), ("lever_sign", C.c_double), ("friction_cone_coefficient", C.c_double),
                ("friction_barrier_weight", C.c_double), ("friction_barrier_sharpness", C.c_double),
                ("bound_barrier_weight", C.c_double), ("bound_barrier_sharpness", C.c_double),
                ("lower", C.c_double * 64), ("upper", C.c_double * 64)]


class SddpStats(C.Structure):
    _fields_ = [("cost", C.c_double), ("alpha", C.c_double), ("gap", C.c_double), ("mu", C.c_double),
                ("expected", C.c_double), ("iters", C.c_int), ("converged", C.c_int), ("status", C.c_int),
                ("rollouts", C.c_int)]


STATS_DTYPE = np.dtype([("cost", "f8"), ("alpha", "f8"), ("gap", "f8"), ("mu", "f8"), ("expected", "f8"),
                        ("iters", "i4"), ("converged", "i4"), ("status", "i4"), ("rollouts", "i4")])
assert STATS_DTYPE.itemsize == C.sizeof(SddpStats)

_P = C.POINTER
_dp = _P(C.c_double)
_vp = C.c_void_p

# every symbol include/sddp.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "sddp_abi_version": (C.c_int, []),
    "sddp_model_dims": (C.c_int, [C.c_int, _P(C.c_int), _P(C.c_int), _P(C.c_int)]),
    "sddp_default_options": (None, [_P(SddpOptions)]),
    "sddp_default_consts": (None, [_P(SddpModelConsts)]),
    "sddp_set_params": (C.c_int, [_vp, _vp]),
    "sddp_advance": (C.c_int, [_vp, _vp, _vp]),
    "sddp_solve_resident": (C.c_int, [_vp, _vp, _vp, _vp]),
    "sddp_solve_resident_first": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "sddp_model_step": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp]),
    "sddp_create": (C.c_int, [_P(_vp), C.c_int, C.c_int, C.c_int, _P(SddpOptions), _P(SddpModelConsts)]),
    "sddp_destroy": (None, [_vp]),
    "sddp_last_error": (C.c_char_p, [_vp]),
    "sddp_set_options": (C.c_int, [_vp, _P(SddpOptions)]),
    "sddp_set_initial_state": (C.c_int, [_vp, _vp]),
    "sddp_set_x_warmstart": (C.c_int, [_vp, _vp]),
    "sddp_set_u_warmstart": (C.c_int, [_vp, _vp]),
    "sddp_solve": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "sddp_is_converged": (C.c_int, [_vp, _vp]),
    "sddp_set_stream": (C.c_int, [_vp, _vp]),
    "sddp_set_initial_state_device": (C.c_int, [_vp, _vp]),
    "sddp_set_x_warmstart_device": (C.c_int, [_vp, _vp]),
    "sddp_set_u_warmstart_device": (C.c_int, [_vp, _vp]),
    "sddp_solve_device": (C.c_int, [_vp, _vp]),
    "sddp_load_range_device": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "sddp_solve_range_device": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "sddp_queue_info": (C.c_int, [_vp, _P(C.c_int), _P(C.c_int), _P(C.c_int)]),
    "sddp_fetch": (C.c_int, [_vp, _vp, _vp, _vp]),
    "sddp_synchronize": (C.c_int, [_vp]),
    "sddp_device_ptr": (C.c_int, [_vp, C.c_int, _P(_vp), _P(C.c_longlong)]),
    "sddp_last_kernel_ms": (C.c_int, [_vp, _P(C.c_double)]),
    "sddp_enable_timing": (C.c_int, [_vp, C.c_int]),
    "sddp_kernel_time_stats": (C.c_int, [_vp, _P(C.c_double), _P(C.c_longlong), C.c_int]),
    "sddp_eval_knots": (C.c_int, [C.c_int, _P(SddpModelConsts), C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sddp_backward": (C.c_int, [_vp, _vp, C.c_double, _vp, _vp]),
    "sddp_forward": (C.c_int, [_vp, _vp, C.c_double, _vp, _vp, _vp]),
}

_lib = None


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/sddp_api.hip for gfx950 into libsddp_hip.so (in-tree, so it travels to the GPU box)."""
    src = os.path.join(CSRC, "sddp_api.hip")
    sort_src = os.path.join(CSRC, "sddp_sort.hip")
    deps = [src, sort_src, os.path.join(CSRC, "sddp_kernels.hpp"), os.path.join(CSRC, "sddp_kernels_mw.hpp"), os.path.join(CSRC, "sddp_models.hpp"),
            os.path.join(CSRC, "sddp_sort.hpp"), os.path.join(INCLUDE, "sddp.h")]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("SDDP_CXXFLAGS", "").split()      # diagnostic builds only (-DSDDP_NO_MFMA, -DSDDP_STAMPS), with SDDP_LIB
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-I" + INCLUDE, "-I" + CSRC, *extra,
           src, sort_src, "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


def load():
    """dlopen the HIP library and type every symbol of include/sddp.h.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the DDP engine)")
    try:
        # PyTorch bundles its own HIP runtime with the same soname (libamdhip64.so.7).  Load it first so the process has
        # ONE HIP runtime shared by torch tensors / streams and this library (loading the system one first and torch's
        # second leaves this library without a visible device).
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.sddp_abi_version() != 7:
        raise RuntimeError("libsddp_hip.so ABI version mismatch")
    _lib = lib
    return lib


def default_options(**over) -> SddpOptions:
    o = SddpOptions()
    load().sddp_default_options(C.byref(o))
    for k, v in over.items():
        if not hasattr(o, k):
            raise KeyError(f"unknown option {k!r}")
        setattr(o, k, v)
    return o


def default_consts(**over) -> SddpModelConsts:
    c = SddpModelConsts()
    load().sddp_default_consts(C.byref(c))
    set_consts(c, **over)
    return c


def set_consts(c: SddpModelConsts, **over):
    for k, v in over.items():
        if k in ("lower", "upper"):                    # bounds of z = [x u]: the first nx + nu entries, the rest stays unbounded
            field = getattr(c, k)
            if v is not None:
                arr = np.asarray(v, dtype=float).reshape(-1)
                if arr.size > len(field):
                    raise ValueError(f"{k}: at most {len(field)} values")
                for i, a in enumerate(arr):
                    field[i] = float(a)
        elif k in ("I", "com", "feet"):
            arr = np.asarray(v, dtype=float).reshape(-1)
            field = getattr(c, k)
            if arr.size != len(field):
                raise ValueError(f"{k}: expected {len(field)} values")
            for i, a in enumerate(arr):
                field[i] = float(a)
        elif hasattr(c, k):
            setattr(c, k, v)
        else:
            raise KeyError(f"unknown model constant {k!r}")
    return c


def check(rc: int, handle=None):
    if rc != 0:
        msg = load().sddp_last_error(handle)
        raise RuntimeError(f"sddp error {rc}: {msg.decode() if msg else '?'}")


def ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def model_dims(model: str):
    nx, nu, npar = C.c_int(), C.c_int(), C.c_int()
    check(load().sddp_model_dims(MODEL_IDS[model], C.byref(nx), C.byref(nu), C.byref(npar)))
    return nx.value, nu.value, npar.value
