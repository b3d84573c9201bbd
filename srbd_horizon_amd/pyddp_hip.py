"""``pyddp``-shaped module over the HIP engine: what the reference adapter imports (reference python/ddp.py:1) and calls.

    import srbd_horizon_amd.pyddp_hip as pyddp            # ddp.py:1
    opts = pyddp.DdpSolverOptions()                        # ddp.py:14, fields set at :18-35
    solver = pyddp.DdpSolver(nx, nu, f_list, L_list, L_term, opts)     # ddp.py:93-94
    x, u = solver.solve(param_values_list)                 # ddp.py:101   list[N+1] of list[np] -> ndarray [nx, N+1], [nu, N]
    solver.is_converged()                                  # ddp.py:106
    solver.set_initial_state(x0); solver.set_x_warmstart(x); solver.set_u_warmstart(u)     # ddp.py:113-123

The one thing that cannot keep its shape: ``f_list`` / ``L_list`` / ``L_term`` are CasADi ``Function`` objects in the reference
(ddp.py:83-87, built at :179-230); here the dynamics and costs are hand-written analytic HIP models, so the lists hold
``ModelFunction`` stand-ins made by ``model_functions(model, N, consts)`` -- one per node like the originals, each naming the
registered model, its constants and its node.  Everything else (argument order, list-of-lists parameters, variable-major result
arrays, persistent solver object that keeps the previous solution as the next warm start) is the reference's.  No CPU fallback.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .engine import DdpEngine


class DdpSolverOptions:
    """Mutable option record (ddp.py:14-35).  Defaults are the engine's (``sddp_default_options``); the reference assigns
    ``alpha_converge_threshold`` always and the other fields only when its ``opts`` dict names them."""
    _FIELDS = ("max_iters", "alpha_0", "alpha_converge_threshold", "line_search_decrease_factor", "beta", "cost_reduction_ths", "mu0")

    def __init__(self):
        d = _lib.default_options()
        for k in self._FIELDS:
            setattr(self, k, getattr(d, k))

    def as_dict(self):
        return {k: getattr(self, k) for k in self._FIELDS}


class ModelFunction:
    """Stands where one CasADi Function of f_list / L_list / L_term stands: (kind, registered model, node)."""

    def __init__(self, kind: str, model: str, node: int, consts: dict):
        self.kind, self.model, self.node, self.consts = kind, model, int(node), consts

    def name(self):                       # CasADi Function.name(): "f3", "L3" (ddp.py:209, :225, :230)
        return ("f" if self.kind == "f" else "L") + str(self.node)

    def __repr__(self):
        return f"<{self.model}:{self.name()}>"


def model_functions(model: str, N: int, consts: dict | None = None):
    """-> (f_list[N], L_list[N], L_term): the stand-ins for ddp.py:83-87 (one f_k and L_k per stage node, L_N)."""
    if model not in _lib.MODEL_IDS:
        raise ValueError(f"unknown model {model!r}")
    c = dict(consts or {})
    return ([ModelFunction("f", model, k, c) for k in range(N)], [ModelFunction("L", model, k, c) for k in range(N)],
            ModelFunction("L", model, N, c))


class DdpSolver:
    def __init__(self, nx: int, nu: int, f_list, L_list, L_term, opts: DdpSolverOptions | None = None):
        if len(f_list) != len(L_list) or len(f_list) < 1:
            raise ValueError("f_list and L_list must have one entry per stage node")
        fns = list(f_list) + list(L_list) + [L_term]
        if not all(isinstance(f, ModelFunction) for f in fns):
            raise TypeError("f_list / L_list / L_term must come from pyddp_hip.model_functions(): the HIP engine runs registered "
                            "analytic models, not CasADi graphs")
        if len({(f.model, id(f.consts)) for f in fns}) != 1:
            raise ValueError("all node functions must belong to one model_functions() call")
        N = len(f_list)
        if [f.node for f in f_list] != list(range(N)) or [f.node for f in L_list] != list(range(N)) or L_term.node != N:
            raise ValueError("node functions out of order")
        self.model, self.N = L_term.model, N
        mnx, mnu, mnp = _lib.model_dims(self.model)
        if (int(nx), int(nu)) != (mnx, mnu):
            raise ValueError(f"nx, nu = {nx}, {nu} but model {self.model} has {mnx}, {mnu}")
        self.nx, self.nu, self.np_ = mnx, mnu, mnp
        o = (opts or DdpSolverOptions()).as_dict()
        self._eng = DdpEngine(self.model, N, 1, opts=o, consts=L_term.consts)
        self._have_x0 = self._have_x = self._have_u = False
        self._x0 = None

    # ---- ddp.py:113-123 ------------------------------------------------------------------------------------------------
    def set_initial_state(self, x0):
        self._x0 = np.asarray(x0, dtype=float).reshape(1, self.nx).copy()
        self._eng.set_initial_state(self._x0)
        self._have_x0 = True

    def set_x_warmstart(self, x):
        x = np.asarray(x, dtype=float).reshape(self.nx, self.N + 1)
        self._eng.set_x_warmstart(np.ascontiguousarray(x.T)[None])
        self._have_x = True

    def set_u_warmstart(self, u):
        u = np.asarray(u, dtype=float).reshape(self.nu, self.N)
        self._eng.set_u_warmstart(np.ascontiguousarray(u.T)[None])
        self._have_u = True

    # ---- ddp.py:101, :106 ----------------------------------------------------------------------------------------------
    def solve(self, param_values_list):
        if len(param_values_list) != self.N + 1:
            raise ValueError(f"param_values_list must have one list per node ({self.N + 1})")
        P = np.asarray(param_values_list, dtype=float)
        if P.shape != (self.N + 1, self.np_):
            raise ValueError(f"every node needs {self.np_} parameter values")
        if not self._have_x0:
            raise RuntimeError("set_initial_state() must be called before solve()")
        if not self._have_u:       # default warm start (the examples never pass one): x = x0 everywhere, u = 0
            self._eng.set_u_warmstart(np.zeros((1, self.N, self.nu)))
            self._have_u = True
        if not self._have_x:
            self._eng.set_x_warmstart(np.repeat(self._x0[:, None, :], self.N + 1, axis=1))
            self._have_x = True
        x, u = self._eng.solve(P[None])
        self.stats = self._eng.stats[0]
        return np.ascontiguousarray(x[0].T), np.ascontiguousarray(u[0].T)

    def is_converged(self) -> bool:
        return bool(self._eng.is_converged()[0])
