"""``DDPSolver`` -- the Horizon-``Solver``-shaped adapter of reference python/ddp.py:10-230 over the HIP engine.

Same constructor, methods, option keys and result layout as the reference class:

    solver = DDPSolver(prb, opts)                 # ddp.py:11
    solver.setInitialState(x0)                    # ddp.py:122
    solver.set_x_warmstart(x) / set_u_warmstart(u)  # ddp.py:113-117  ([nx, N+1] / [nu, N])
    ok = solver.solve()                           # ddp.py:96   -> is_converged()
    sol = solver.getSolutionDict()                # ddp.py:119  {var: [dim, nodes], 'x_opt': [nx, N+1], 'u_opt': [nu, N]}

What differs, by construction: the per-node Python parameter loop (ddp.py:98-99, :165-177) is one vectorised
``parameter_matrix()``; the CasADi Function lists (ddp.py:83-94) are a registered analytic model id; the arithmetic
runs in ``libsddp_hip.so`` on the GPU (no CPU fallback).  Inequality constraints are ignored exactly as the reference
ignores them (ddp.py:197-209 is commented out).
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from .engine import DdpEngine
from .problem import Problem

# option keys the reference forwards to pyddp.DdpSolverOptions (ddp.py:16-35) + engine extras
_REFERENCE_KEYS = ("max_iters", "alpha_0", "alpha_converge_threshold", "line_search_decrease_factor", "beta",
                   "cost_reduction_ths", "mu0")
_EXTRA_KEYS = ("initial_rollout", "gap_tol", "mu_min", "mu_max", "second_order", "waves_per_simd")


class DDPSolver:
    def __init__(self, prb: Problem, opts: Dict) -> None:
        if prb.model is None:
            raise ValueError("the problem has no registered analytic model (Problem.setModel)")
        self.prb = prb
        self.opts = dict(opts or {})
        unknown = [k for k in self.opts if k not in _REFERENCE_KEYS + _EXTRA_KEYS]
        if unknown:
            raise KeyError(f"unknown DDP option(s): {unknown}")
        self.state_var = prb.getState().getVars()
        self.state_size = sum(v.getDim() for v in self.state_var)
        self.input_var = prb.getInput().getVars()
        self.input_size = sum(v.getDim() for v in self.input_var)
        self.param_var = prb.getParameters()
        consts = dict(prb.model_consts)
        if prb.getDt() is not None:
            consts["dt"] = prb.getDt()
        self.ddp_solver = DdpEngine(prb.model, prb.nodes - 1, 1, opts=self.opts, consts=consts)
        if (self.ddp_solver.nx, self.ddp_solver.nu) != (self.state_size, self.input_size):
            raise ValueError("problem variables do not match the registered model's dimensions")
        if self.ddp_solver.np_ != sum(p.getDim() for p in self.param_var.values()):
            raise ValueError("problem parameters do not match the registered model's parameter vector")
        self.var_solution = None
        self._have_x0 = self._have_x = self._have_u = False

    # ---- reference surface -----------------------------------------------------------------------------------------
    def setInitialState(self, x0):
        x0 = np.asarray(x0, dtype=float).reshape(1, self.state_size)
        self._x0 = x0.copy()
        self.ddp_solver.set_initial_state(x0)
        self._have_x0 = True

    def set_x_warmstart(self, x):
        x = np.asarray(x, dtype=float).reshape(self.state_size, self.prb.nodes)
        self.ddp_solver.set_x_warmstart(np.ascontiguousarray(x.T)[None])
        self._have_x = True

    def set_u_warmstart(self, u):
        u = np.asarray(u, dtype=float).reshape(self.input_size, self.prb.nodes - 1)
        self.ddp_solver.set_u_warmstart(np.ascontiguousarray(u.T)[None])
        self._have_u = True

    def solve(self) -> bool:
        if not self._have_x0:
            raise RuntimeError("setInitialState() must be called before solve()")
        # Warm start is never passed by the reference examples and pyddp's default is unpinned:
        # the documented default here is x = x0 at every node, u = 0; afterwards the previous solution is kept.
        if not self._have_u:
            self.ddp_solver.set_u_warmstart(np.zeros((1, self.prb.nodes - 1, self.input_size)))
            self._have_u = True
        if not self._have_x:
            self.ddp_solver.set_x_warmstart(np.repeat(self._x0[:, None, :], self.prb.nodes, axis=1))
            self._have_x = True
        params = self.prb.parameter_matrix()[None]                          # ddp.py:98-99, vectorised
        x, u = self.ddp_solver.solve(params)                                # ddp.py:101
        x, u = np.ascontiguousarray(x[0].T), np.ascontiguousarray(u[0].T)   # reference layout [dim, nodes]
        self.var_solution = self._createVarSolDict(x, u)
        self.var_solution["x_opt"] = x                                      # ddp.py:103
        self.var_solution["u_opt"] = u                                      # ddp.py:104
        self.stats = self.ddp_solver.stats[0]
        return bool(self.stats["converged"])                                # ddp.py:106

    def solve_receding(self, x0) -> bool:
        """One tick of the receding-horizon loop with device-resident data.

        Contract: since the previous call the caller has shifted every parameter back by one node and assigned node N -- which
        is all dsrbd_example.py:102-131 / wpg.py:74-99 ever do.  Then only the last parameter column and ``x0`` cross PCIe; the
        engine shifts its resident parameter tensor and warm-starts from its previous solution advanced by one knot.  Same
        result as ``setInitialState(x0)`` + shifted ``set_*_warmstart`` + ``solve()`` (tests/test_gpu_api.py)."""
        x0 = np.asarray(x0, dtype=float).reshape(1, self.state_size)
        self._x0 = x0.copy()
        pm = self.prb.parameter_matrix()
        if not getattr(self, "_resident", False):
            self.setInitialState(x0)
            if not self._have_u:
                self.ddp_solver.set_u_warmstart(np.zeros((1, self.prb.nodes - 1, self.input_size)))
                self._have_u = True
            if not self._have_x:
                self.ddp_solver.set_x_warmstart(np.repeat(self._x0[:, None, :], self.prb.nodes, axis=1))
                self._have_x = True
            self.ddp_solver.set_params(pm[None])
            self._resident = True
        else:
            self.ddp_solver.advance(pm[-1][None], x0)
        x, u = self.ddp_solver.solve_resident()
        x, u = np.ascontiguousarray(x[0].T), np.ascontiguousarray(u[0].T)
        self.var_solution = self._createVarSolDict(x, u)
        self.var_solution["x_opt"] = x
        self.var_solution["u_opt"] = u
        self.stats = self.ddp_solver.stats[0]
        return bool(self.stats["converged"])

    def getSolutionDict(self):
        return self.var_solution

    def is_equality_constraint(self, constr):                               # ddp.py:108-111
        upper = np.array(constr.getUpperBounds())
        lower = np.array(constr.getLowerBounds())
        return np.linalg.norm(upper - lower) <= 1e-6

    def _createVarSolDict(self, x, u):
        """ddp.py:125-151: walk the variables in creation order; states first, then inputs."""
        sol = {}
        acc = pos_x = pos_u = 0
        for var in self.prb.var_container.getVarList(offset=False):
            acc += var.size()[0]
            if acc <= self.state_size:
                sol[var.getName()] = x[pos_x:acc, :]
                pos_x = acc
            else:
                sol[var.getName()] = u[pos_u:acc - pos_x, :]
                pos_u = acc - pos_x
        return sol
