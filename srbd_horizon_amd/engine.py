"""Batched DDP engine object: the Python face of one ``sddp_handle`` (include/sddp.h).

Stands where ``pyddp.DdpSolver`` stands in the reference (python/ddp.py:93-94): constructed once, kept across MPC
ticks, ``solve(params) -> (x, u)``, ``is_converged()``, ``set_initial_state``, ``set_x_warmstart``,
``set_u_warmstart`` -- widened from one problem to a batch of B independent problems (knot-major C ABI layout).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class DdpEngine:
    def __init__(self, model: str, N: int, batch: int = 1, opts: dict | None = None, consts: dict | None = None):
        self.lib = _lib.load()
        self.model = model
        self.N, self.B = int(N), int(batch)
        self.nx, self.nu, self.np_ = _lib.model_dims(model)
        self.opts = _lib.default_options(**(opts or {}))
        self.consts = _lib.default_consts(model, **(consts or {}))
        h = C.c_void_p()
        _lib.check(self.lib.sddp_create(C.byref(h), _lib.MODEL_IDS[model], self.N, self.B,
                                        C.byref(self.opts), C.byref(self.consts)))
        self.h = h
        npar = C.c_int()                     # a handle with user rows (consts extra_rows) has MAX_EXTRA more parameter columns
        self._chk(self.lib.sddp_handle_dims(h, None, None, C.byref(npar)))
        self.np_ = npar.value
        self._keep = []

    # ---- lifetime ------------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.lib.sddp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        _lib.check(rc, self.h)

    @staticmethod
    def _c(a, shape):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.shape != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
        return a

    def set_options(self, **over):
        for k, v in over.items():
            if not hasattr(self.opts, k):
                raise KeyError(f"unknown option {k!r}")
            setattr(self.opts, k, v)
        self._chk(self.lib.sddp_set_options(self.h, C.byref(self.opts)))

    # ---- host (numpy) path: [B][N+1][nx] etc. -------------------------------------------------------------------------
    def set_initial_state(self, x0):
        a = self._c(x0, (self.B, self.nx))
        self._chk(self.lib.sddp_set_initial_state(self.h, _lib.ptr(a)))

    def set_x_warmstart(self, x):
        a = self._c(x, (self.B, self.N + 1, self.nx))
        self._chk(self.lib.sddp_set_x_warmstart(self.h, _lib.ptr(a)))

    def set_u_warmstart(self, u):
        a = self._c(u, (self.B, self.N, self.nu))
        self._chk(self.lib.sddp_set_u_warmstart(self.h, _lib.ptr(a)))

    def solve(self, params):
        p = self._c(params, (self.B, self.N + 1, self.np_))
        x = np.empty((self.B, self.N + 1, self.nx))
        u = np.empty((self.B, self.N, self.nu))
        st = np.zeros(self.B, dtype=_lib.STATS_DTYPE)
        self._chk(self.lib.sddp_solve(self.h, _lib.ptr(p), _lib.ptr(x), _lib.ptr(u), _lib.ptr(st)))
        self.stats = st
        return x, u

    # ---- receding horizon with device-resident parameters / warm start -----------------------------
    def set_params(self, params):
        p = self._c(params, (self.B, self.N + 1, self.np_))
        self._chk(self.lib.sddp_set_params(self.h, _lib.ptr(p)))

    def advance(self, p_last, x0):
        """One tick on the device: parameters and previous solution shifted back by one knot, ``p_last`` written at node N,
        ``x0`` set as the initial state."""
        pl = self._c(p_last, (self.B, self.np_))
        x = self._c(x0, (self.B, self.nx))
        self._chk(self.lib.sddp_advance(self.h, _lib.ptr(pl), _lib.ptr(x)))

    def model_step(self, x, u, p, k: int = 0):
        """x_next = f_k(x, u; p) per instance through the solver's device model code (simulator step, dsrbd_example.py:158-159)."""
        xv = self._c(x, (self.B, self.nx)); uv = self._c(u, (self.B, self.nu)); pv = self._c(p, (self.B, self.np_))
        out = np.empty((self.B, self.nx))
        self._chk(self.lib.sddp_model_step(self.h, _lib.ptr(xv), _lib.ptr(uv), _lib.ptr(pv), int(k), _lib.ptr(out)))
        return out

    def solve_resident(self):
        x = np.empty((self.B, self.N + 1, self.nx))
        u = np.empty((self.B, self.N, self.nu))
        st = np.zeros(self.B, dtype=_lib.STATS_DTYPE)
        self._chk(self.lib.sddp_solve_resident(self.h, _lib.ptr(x), _lib.ptr(u), _lib.ptr(st)))
        self.stats = st
        return x, u

    def solve_resident_first(self):
        """One tick of a fleet in closed loop: solve on the resident data, fetch only u_0 [B,nu], x_1 [B,nx], cost, iterations and
        status per robot (the trajectories stay on the device as the next warm start).  -> (u0, x1); self.first_stats"""
        u0 = np.empty((self.B, self.nu))
        x1 = np.empty((self.B, self.nx))
        cost = np.empty(self.B)
        iters = np.empty(self.B, dtype=np.int32)
        status = np.empty(self.B, dtype=np.int32)
        self._chk(self.lib.sddp_solve_resident_first(self.h, _lib.ptr(u0), _lib.ptr(x1), _lib.ptr(cost), _lib.ptr(iters), _lib.ptr(status)))
        self.first_stats = dict(cost=cost, iters=iters, status=status)
        return u0, x1

    def is_converged(self):
        f = np.zeros(self.B, dtype=np.int32)
        self._chk(self.lib.sddp_is_converged(self.h, _lib.ptr(f)))
        return f.astype(bool)

    # ---- phase-level entry points (parity tests) ---------------------------------------------------------------------------
    def backward(self, params, mu=0.0):
        p = self._c(params, (self.B, self.N + 1, self.np_))
        gains = np.empty((self.B, self.N, self.nu * (self.nx + 1)))
        scal = np.empty((self.B, 8))
        self._chk(self.lib.sddp_backward(self.h, _lib.ptr(p), float(mu), _lib.ptr(gains), _lib.ptr(scal)))
        kff = gains[:, :, :self.nu]
        K = gains[:, :, self.nu:].reshape(self.B, self.N, self.nu, self.nx)
        return kff, K, scal

    def forward(self, params, alpha):
        p = self._c(params, (self.B, self.N + 1, self.np_))
        x = np.empty((self.B, self.N + 1, self.nx))
        u = np.empty((self.B, self.N, self.nu))
        cost = np.empty(self.B)
        self._chk(self.lib.sddp_forward(self.h, _lib.ptr(p), float(alpha), _lib.ptr(x), _lib.ptr(u), _lib.ptr(cost)))
        return x, u, cost

    # ---- HBM-resident path (torch tensors on the GPU; PyTorch is plumbing for device memory and streams) ------------------
    def _dev(self, t, shape):
        import torch
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
            raise ValueError("expected a contiguous float64 CUDA tensor")
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return C.c_void_p(t.data_ptr())

    def use_torch_stream(self, stream=None):
        import torch
        s = stream if stream is not None else torch.cuda.current_stream()
        self._chk(self.lib.sddp_set_stream(self.h, C.c_void_p(s.cuda_stream)))

    def set_initial_state_device(self, x0):
        self._chk(self.lib.sddp_set_initial_state_device(self.h, self._dev(x0, (self.B, self.nx))))

    def set_x_warmstart_device(self, x):
        self._chk(self.lib.sddp_set_x_warmstart_device(self.h, self._dev(x, (self.B, self.N + 1, self.nx))))

    def set_u_warmstart_device(self, u):
        self._chk(self.lib.sddp_set_u_warmstart_device(self.h, self._dev(u, (self.B, self.N, self.nu))))

    def solve_device(self, params):
        """Asynchronous launch on the handle's stream; results stay in the handle's HBM buffers."""
        self._chk(self.lib.sddp_solve_device(self.h, self._dev(params, (self.B, self.N + 1, self.np_))))

    def synchronize(self):
        self._chk(self.lib.sddp_synchronize(self.h))

    def enable_timing(self, on=True):
        self._chk(self.lib.sddp_enable_timing(self.h, int(on)))

    def last_kernel_ms(self):
        ms = C.c_double()
        self._chk(self.lib.sddp_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def kernel_time_stats(self, reset=False):
        """(sum_ms, count) of the solve-kernel durations measured with HIP events on the handle's stream."""
        sm, n = C.c_double(), C.c_longlong()
        self._chk(self.lib.sddp_kernel_time_stats(self.h, C.byref(sm), C.byref(n), int(reset)))
        return sm.value, n.value

    def device_buffer(self, which: int):
        p, n = C.c_void_p(), C.c_longlong()
        self._chk(self.lib.sddp_device_ptr(self.h, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def fetch_device_views(self):
        """Zero-copy torch views of the handle's HBM buffers: x [B,N+1,nx], u [B,N,nu], stats as f64 [B,8] and i32 [B,16]
        (sddp_stats: cost, alpha, gap, mu, expected, rho | iters, converged, status, rollouts = i32 words 12..15;
        _lib.STATS_I32_ITERS etc.)."""
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())

        class _Dev:
            def __init__(self, ptr, shape, typestr):
                self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}

        px, _ = self.device_buffer(0)
        pu, _ = self.device_buffer(1)
        ps, _ = self.device_buffer(2)
        x = torch.as_tensor(_Dev(px, (self.B, self.N + 1, self.nx), "<f8"), device=dev)
        u = torch.as_tensor(_Dev(pu, (self.B, self.N, self.nu), "<f8"), device=dev)
        sf = torch.as_tensor(_Dev(ps, (self.B, _lib.STATS_F64_WORDS), "<f8"), device=dev)
        si = torch.as_tensor(_Dev(ps, (self.B, _lib.STATS_I32_WORDS), "<i4"), device=dev)
        return x, u, sf, si

    def fetch(self):
        """Copy the solution and stats of the last device solve to host numpy arrays (waits for the handle's stream)."""
        x = np.empty((self.B, self.N + 1, self.nx))
        u = np.empty((self.B, self.N, self.nu))
        st = np.zeros(self.B, dtype=_lib.STATS_DTYPE)
        self._chk(self.lib.sddp_fetch(self.h, _lib.ptr(x), _lib.ptr(u), _lib.ptr(st)))
        self.stats = st
        return x, u, st

    # ---- the handle as a queue of instances (more instances than resident workgroups: one launch, work queue) -----------------
    def _dev_or_null(self, t, shape):
        return C.c_void_p(None) if t is None else self._dev(t, shape)

    def load_range_device(self, first: int, count: int, x0=None, x=None, u=None):
        """Initial state / warm start of the instances [first, first + count) from device tensors of `count` instances."""
        self._chk(self.lib.sddp_load_range_device(self.h, int(first), int(count), self._dev_or_null(x0, (count, self.nx)),
                                                  self._dev_or_null(x, (count, self.N + 1, self.nx)),
                                                  self._dev_or_null(u, (count, self.N, self.nu))))

    def solve_range_device(self, params, first: int, count: int):
        """One asynchronous launch over the instances [first, first + count); `params` is the whole [B, N+1, np] tensor."""
        self._chk(self.lib.sddp_solve_range_device(self.h, self._dev(params, (self.B, self.N + 1, self.np_)), int(first), int(count)))

    # ---- class history (queue_order = 3) -------------------------------------------------------------------------------------
    def set_instance_classes(self, classes, n_classes: int):
        """classes [B] int (host): what kind of problem each instance is (-1: unlabelled); sddp.h queue_order 3"""
        c = np.ascontiguousarray(classes, dtype=np.int32)
        if c.shape != (self.B,):
            raise ValueError(f"expected {self.B} class labels")
        self._chk(self.lib.sddp_set_instance_classes(self.h, _lib.ptr(c), int(n_classes)))

    def set_instance_classes_range_device(self, first: int, count: int, classes, n_classes: int):
        """labels of the instances [first, first + count) from a device int32 tensor (asynchronous on the handle's stream)"""
        import torch
        if not (isinstance(classes, torch.Tensor) and classes.is_cuda and classes.dtype == torch.int32 and classes.is_contiguous()
                and tuple(classes.shape) == (count,)):
            raise ValueError("expected a contiguous int32 CUDA tensor of `count` labels")
        self._chk(self.lib.sddp_set_instance_classes_range_device(self.h, int(first), int(count), C.c_void_p(classes.data_ptr()), int(n_classes)))

    def class_history(self, cls: int):
        """-> (mean iterations, solves) of class `cls` on this handle so far"""
        m, n = C.c_double(), C.c_longlong()
        self._chk(self.lib.sddp_class_history(self.h, int(cls), C.byref(m), C.byref(n)))
        return m.value, n.value

    RECORD_MODES = {"full": 0, "first_knot": 1}

    def record_words(self, mode="full"):
        w = C.c_int()
        self._chk(self.lib.sddp_record_words(self.h, self.RECORD_MODES[mode], C.byref(w)))
        return w.value

    def pack_records_device(self, out, first: int, count: int, mode="full"):
        """The solution records of the instances [first, first + count) into the device tensor `out` [count, record_words(mode)]
        (one kernel on the handle's stream, behind the solve): what a sharded fleet's all-gather sends."""
        self._chk(self.lib.sddp_pack_records_device(self.h, int(first), int(count), self.RECORD_MODES[mode],
                                                    self._dev(out, (count, self.record_words(mode)))))
        return out

    def last_queue_order(self):
        """Instance indices in the order the last queued launch handed them out (queue_order 1 or 2); waits for the stream."""
        import torch
        self.synchronize()
        ptr, nbytes = self.device_buffer(6)

        class _Dev:
            __cuda_array_interface__ = {"shape": (nbytes // 4,), "typestr": "<i4", "data": (int(ptr), False), "version": 2}

        return torch.as_tensor(_Dev(), device=torch.device("cuda", torch.cuda.current_device())).cpu().numpy().copy()

    def slot_times(self):
        """[grid, 2] uint64: when each slot of the last solve launch started its first instance and when it found the queue empty
        (100 MHz constant-rate clock); waits for the stream."""
        import torch
        self.synchronize()
        ptr, nbytes = self.device_buffer(7)

        class _Dev:
            __cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<i8", "data": (int(ptr), False), "version": 2}

        t = torch.as_tensor(_Dev(), device=torch.device("cuda", torch.cuda.current_device())).cpu().numpy().copy()
        return t.view(np.uint64).reshape(-1, 2)

    def queue_info(self):
        """(slots the work buffers exist for, grid of the last launch, queue length of the last launch or 0)."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._chk(self.lib.sddp_queue_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def poison_lds(self):
        """Diagnostic (include/sddp.h): fill the LDS of every CU with NaNs, so the next launch cannot pass on what an earlier kernel
        happened to leave in a word it reads before writing."""
        self._chk(self.lib.sddp_debug_poison_lds(self.h))

    def kernel_info(self):
        """-> dict(kernel, wavefronts_per_instance, waves_per_simd): the solve kernel the LAST launch of this handle ran
        (``solve_kernel[_w2]<model>`` on one wavefront per instance or ``solve_kernel_mw[_w2]<model>`` on four; ``_w2`` = the
        half-register-file build, which a handle asked for two per SIMD falls back from where it gains nothing)."""
        w, b, nm = C.c_int(), C.c_int(), C.c_char_p()
        self._chk(self.lib.sddp_kernel_info(self.h, C.byref(w), C.byref(b), C.byref(nm)))
        base = "solve_kernel_mw" if w.value == 4 else "solve_kernel"
        out = dict(kernel=f"{base}{'_w2' if b.value == 2 else ''}<{nm.value.decode()}>", wavefronts_per_instance=w.value,
                   waves_per_simd=b.value)
        v, sc, lds, per = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        if self.lib.sddp_kernel_resources(self.h, C.byref(v), C.byref(sc), C.byref(lds), C.byref(per)) == 0:
            out["resources"] = dict(vgprs=v.value, scratch_bytes_per_lane=sc.value, lds_bytes_per_workgroup=lds.value,
                                    workgroups_per_cu=per.value)
        return out


def eval_knots(model: str, N: int, k, x, u, p, consts: dict | None = None):
    """Per-knot model evaluation on the GPU: f, [fx fu], GN Hessian, gradient, cost (parity tests)."""
    lib = _lib.load()
    nx, nu, npar = _lib.model_dims(model)
    nz = nx + nu
    k = np.ascontiguousarray(k, dtype=np.int32)
    nk = k.shape[0]
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(nk, nx)
    u = np.ascontiguousarray(u, dtype=np.float64).reshape(nk, nu)
    cst = _lib.default_consts(model, **(consts or {}))
    if cst.n_extra:
        npar += _lib.MAX_EXTRA               # the user rows' reference columns
    p = np.ascontiguousarray(p, dtype=np.float64).reshape(nk, npar)
    f = np.empty((nk, nx)); F = np.empty((nk, nx, nz)); H = np.empty((nk, nz, nz)); g = np.empty((nk, nz)); L = np.empty(nk)
    _lib.check(lib.sddp_eval_knots(_lib.MODEL_IDS[model], C.byref(cst), int(N), nk, _lib.ptr(k), _lib.ptr(x), _lib.ptr(u),
                                   _lib.ptr(p), _lib.ptr(f), _lib.ptr(F), _lib.ptr(H), _lib.ptr(g), _lib.ptr(L)))
    return f, F, H, g, L
