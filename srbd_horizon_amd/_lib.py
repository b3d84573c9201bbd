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

MODEL_IDS = {"srbd13": 0, "srbd37": 1, "lip30": 2, "srbd61": 3}


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
                ("lower", C.c_double * 64), ("upper", C.c_double * 64), ("relative_velocity_constraints", C.c_int),
                ("n_extra", C.c_int), ("extra_kind", C.c_int * 8), ("extra_weight", C.c_double * 8), ("extra_const", C.c_double * 8),
                ("extra_a", C.c_double * 1024)]


class SddpStats(C.Structure):
    _fields_ = [("cost", C.c_double), ("alpha", C.c_double), ("gap", C.c_double), ("mu", C.c_double),
                ("expected", C.c_double), ("rho", C.c_double), ("iters", C.c_int), ("converged", C.c_int), ("status", C.c_int),
                ("rollouts", C.c_int)]


STATS_DTYPE = np.dtype([("cost", "f8"), ("alpha", "f8"), ("gap", "f8"), ("mu", "f8"), ("expected", "f8"), ("rho", "f8"),
                        ("iters", "i4"), ("converged", "i4"), ("status", "i4"), ("rollouts", "i4")])
assert STATS_DTYPE.itemsize == C.sizeof(SddpStats) == 64
# one sddp_stats record seen as words (the zero-copy device views of engine.fetch_device_views)
STATS_F64_WORDS, STATS_I32_WORDS = 8, 16
STATS_F64_COST, STATS_I32_ITERS, STATS_I32_STATUS, STATS_I32_ROLLOUTS = 0, 12, 14, 15

_P = C.POINTER
_dp = _P(C.c_double)
_vp = C.c_void_p

# every symbol include/sddp.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "sddp_abi_version": (C.c_int, []),
    "sddp_model_dims": (C.c_int, [C.c_int, _P(C.c_int), _P(C.c_int), _P(C.c_int)]),
    "sddp_handle_dims": (C.c_int, [_vp, _P(C.c_int), _P(C.c_int), _P(C.c_int)]),
    "sddp_default_options": (None, [_P(SddpOptions)]),
    "sddp_default_consts": (None, [_P(SddpModelConsts)]),
    "sddp_default_consts_for": (C.c_int, [C.c_int, _P(SddpModelConsts)]),
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
    "sddp_record_words": (C.c_int, [_vp, C.c_int, _P(C.c_int)]),
    "sddp_pack_records_device": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp]),
    "sddp_set_instance_classes": (C.c_int, [_vp, _vp, C.c_int]),
    "sddp_set_instance_classes_range_device": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int]),
    "sddp_class_history": (C.c_int, [_vp, C.c_int, _P(C.c_double), _P(C.c_longlong)]),
    "sddp_queue_info": (C.c_int, [_vp, _P(C.c_int), _P(C.c_int), _P(C.c_int)]),
    "sddp_fetch": (C.c_int, [_vp, _vp, _vp, _vp]),
    "sddp_synchronize": (C.c_int, [_vp]),
    "sddp_kernel_info": (C.c_int, [_vp, _P(C.c_int), _P(C.c_int), _P(C.c_char_p)]),
    "sddp_kernel_resources": (C.c_int, [_vp, _P(C.c_int), _P(C.c_int), _P(C.c_int), _P(C.c_int)]),
    "sddp_debug_poison_lds": (C.c_int, [_vp]),
    "sddp_device_ptr": (C.c_int, [_vp, C.c_int, _P(_vp), _P(C.c_longlong)]),
    "sddp_last_kernel_ms": (C.c_int, [_vp, _P(C.c_double)]),
    "sddp_enable_timing": (C.c_int, [_vp, C.c_int]),
    "sddp_kernel_time_stats": (C.c_int, [_vp, _P(C.c_double), _P(C.c_longlong), C.c_int]),
    "sddp_eval_knots": (C.c_int, [C.c_int, _P(SddpModelConsts), C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sddp_backward": (C.c_int, [_vp, _vp, C.c_double, _vp, _vp]),
    "sddp_forward": (C.c_int, [_vp, _vp, C.c_double, _vp, _vp, _vp]),
}

_lib = None


# model builds of the library: (accessor suffix, device model type, model name).  One translation unit each (csrc/sddp_inst.hip).
INSTANCES = [
    ("srbd13", "Srbd13", "srbd13"), ("srbd13_b", "Srbd13B", "srbd13"), ("srbd13_s", "Srbd13S", "srbd13"), ("srbd13_bs", "Srbd13BS", "srbd13"),
    ("srbd37", "Srbd37", "srbd37"), ("srbd37_b", "Srbd37B", "srbd37"), ("srbd37_s", "Srbd37S", "srbd37"), ("srbd37_bs", "Srbd37BS", "srbd37"),
    ("lip30", "Lip30", "lip30"), ("srbd61", "Srbd61", "srbd61"),
    ("srbd13_x", "Srbd13X", "srbd13"), ("srbd37_x", "Srbd37X", "srbd37"), ("lip30_x", "Lip30X", "lip30"),    # user rows (n_extra > 0)
    ("srbd61_x", "Srbd61X", "srbd61"), ("srbd61_b", "Srbd61B", "srbd61"),    # srbd61: user rows; friction-cone barrier
]
# Per-build compiler options.  srbd61 (one workgroup per CU, 512 registers a lane, still 1.8 KB of scratch): LLVM's
# -sink-insts-to-avoid-spills moves hoisted loop-invariant address arithmetic back into the loops instead of spilling it:
# 25.8 -> 27.3 k solves/s when it went in, 31.8 -> 32.3 k on the round's final kernel (profiles/r04/experiments/README.md).  Measured and NOT applied elsewhere: srbd13 -1 % (scratch 524 -> 288 B
# but slower), srbd37 +2.3 % in the two-per-SIMD build and -3 % in the other, which share a translation unit.
INSTANCE_FLAGS = {k: ["-mllvm", "-sink-insts-to-avoid-spills"] for k in ("srbd61", "srbd61_x", "srbd61_b")}
HEADERS = ["sddp_kernels.hpp", "sddp_kernels_mw.hpp", "sddp_models.hpp", "sddp_sort.hpp", "sddp_handle.hpp", "sddp_launch.hpp", "sddp_kernels_host.hpp"]


def _newer(target: str, deps) -> bool:
    return os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(d) for d in deps)


def _cmd_stamp(obj: str) -> str:
    return obj + ".cmd"


def _fresh(obj: str, deps, cmd) -> bool:
    """An object is fresh when it is newer than its sources AND was compiled by exactly this command line (flags, defines)."""
    if not _newer(obj, deps):
        return False
    try:
        return open(_cmd_stamp(obj)).read() == " ".join(cmd)
    except OSError:
        return False


def build(force: bool = False, verbose: bool = False, only=None) -> str:
    """Compile the HIP library for gfx950 into libsddp_hip.so (in-tree, so it travels to the GPU box): the host API
    (csrc/sddp_api.hip), the queue sort (csrc/sddp_sort.hip) and one object per model build (csrc/sddp_inst.hip x INSTANCES),
    compiled in parallel and linked.  Objects live in build/obj (git-ignored), each beside a stamp of the command line that
    made it; stale ones (older than a source or header, or made by another command line) are recompiled.
    only: recompile just these model builds (development); refuses to link while another object is stale."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("SDDP_CXXFLAGS", "").split()      # diagnostic builds only (-DSDDP_STAMPS), with SDDP_LIB
    default = LIB_PATH.endswith("libsddp_hip.so") and not extra
    tag = "" if default else "_" + hashlib.sha1(repr((LIB_PATH, tuple(extra))).encode()).hexdigest()[:10]
    objdir = os.path.join(ROOT, "build", "obj" + tag)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(INCLUDE, "sddp.h")]
    srcs = [os.path.join(CSRC, n + ".hip") for n in ("sddp_api", "sddp_sort", "sddp_inst")]
    if not force and only is None and not os.path.isdir(objdir) and _newer(LIB_PATH, srcs + hdrs):
        return LIB_PATH                                      # a shipped library newer than every source: nothing to do
    os.makedirs(objdir, exist_ok=True)
    base = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC, *extra, "-c"]
    jobs = []
    for name in ("sddp_api", "sddp_sort"):
        src = os.path.join(CSRC, name + ".hip")
        obj = os.path.join(objdir, name + ".o")
        jobs.append((obj, src, base + [src, "-o", obj]))
    inst = os.path.join(CSRC, "sddp_inst.hip")
    for fn, model, mname in INSTANCES:
        obj = os.path.join(objdir, "inst_" + fn + ".o")
        defs = ["-DSDDP_INST_MODEL=" + model, "-DSDDP_INST_FN=ops_" + fn, '-DSDDP_INST_NAME="' + mname + '"', *INSTANCE_FLAGS.get(fn, [])]
        jobs.append((obj, inst, base + defs + [inst, "-o", obj]))
    todo, left_stale = [], []
    for obj, src, cmd in jobs:
        picked = only is not None and any(obj.endswith("inst_" + o + ".o") for o in only)
        if force or picked or not _fresh(obj, [src] + hdrs, cmd):
            if only is not None and not picked and not force:
                left_stale.append(os.path.basename(obj))
                continue
            todo.append((obj, src, cmd))
    if left_stale:
        raise RuntimeError(f"build(only={only}): {left_stale} are stale too; build without `only` first")
    if not todo and _newer(LIB_PATH, [j[0] for j in jobs]):
        return LIB_PATH

    def compile_one(job):
        obj, src, cmd = job
        if verbose:
            print(" ".join(cmd), flush=True)
        if os.path.exists(_cmd_stamp(obj)):
            os.remove(_cmd_stamp(obj))
        subprocess.run(cmd, check=True)
        with open(_cmd_stamp(obj), "w") as f:
            f.write(" ".join(cmd))

    workers = max(1, min(len(todo), int(os.environ.get("SDDP_BUILD_JOBS", str(os.cpu_count() or 4)))))
    if todo:
        with ThreadPoolExecutor(max_workers=workers) as ex:
            list(ex.map(compile_one, todo))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[j[0] for j in jobs], "-o", LIB_PATH]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.run(link, check=True)
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
    if lib.sddp_abi_version() != 9:
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


def default_consts(model: str | None = None, **over) -> SddpModelConsts:
    """The synthetic robot as `model` needs it (sddp_default_consts_for: srbd61's `feet` are the first four sole corners of the
    eight-point contact model); model None = the line-foot robot of sddp_default_consts."""
    c = SddpModelConsts()
    if model is None:
        load().sddp_default_consts(C.byref(c))
    else:
        check(load().sddp_default_consts_for(MODEL_IDS[model], C.byref(c)))
    set_consts(c, **over)
    return c


MAX_EXTRA = 8


def _set_extra_rows(c: SddpModelConsts, rows):
    """rows: sequence of dict(a [nx + nu], w, kind "state" | "stage", const) -- user-declared linear residual rows
    sqrt(w) (a . z - (p[np + j] + const)) (include/sddp.h extra_*); None / () clears them."""
    rows = list(rows or ())
    if len(rows) > MAX_EXTRA:
        raise ValueError(f"at most {MAX_EXTRA} extra rows")
    c.n_extra = len(rows)
    for j in range(MAX_EXTRA):
        c.extra_kind[j] = 0
        c.extra_weight[j] = c.extra_const[j] = 0.0
        for i in range(128):
            c.extra_a[128 * j + i] = 0.0
    for j, r in enumerate(rows):
        a = np.asarray(r["a"], dtype=float).reshape(-1)
        if a.size > 128:
            raise ValueError("extra row: at most 128 coefficients")
        if r["kind"] not in ("state", "stage"):
            raise ValueError("extra row kind must be 'state' or 'stage'")
        c.extra_kind[j] = 0 if r["kind"] == "state" else 1
        c.extra_weight[j] = float(r["w"])
        c.extra_const[j] = float(r.get("const", 0.0))
        for i, v in enumerate(a):
            c.extra_a[128 * j + i] = float(v)


def set_consts(c: SddpModelConsts, **over):
    for k, v in over.items():
        if k == "extra_rows":
            _set_extra_rows(c, v)
            continue
        if k in ("lower", "upper"):                    # bounds of z = [x u]: the first nx + nu entries, the rest stays unbounded
            field = getattr(c, k)
            if v is not None:
                arr = np.asarray(v, dtype=float).reshape(-1)
                if arr.size > len(field):
                    raise ValueError(f"{k}: at most {len(field)} values")
                for i in range(len(field)):              # the tail is reset: a reused struct keeps no stale bound
                    field[i] = float(arr[i]) if i < arr.size else (-np.inf if k == "lower" else np.inf)
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
