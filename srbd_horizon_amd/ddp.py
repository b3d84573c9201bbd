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
from .problem import LinearTerm, Problem

# option keys the reference forwards to pyddp.DdpSolverOptions (ddp.py:16-35) + engine extras
_REFERENCE_KEYS = ("max_iters", "alpha_0", "alpha_converge_threshold", "line_search_decrease_factor", "beta",
                   "cost_reduction_ths", "mu0")
_EXTRA_KEYS = ("initial_rollout", "gap_tol", "mu_min", "mu_max", "second_order", "waves_per_simd", "queue_order", "max_slots")
# the reference's commented-out bound barrier (ddp.py:203-208) as an opt-in: weight 0 / absent = the reference's behaviour (variable
# bounds ignored); > 0: exponential barrier on the bounds set with Variable.setBounds, sharpness = exp_parameter (ddp.py:182)
_BARRIER_KEYS = ("bound_barrier_weight", "bound_barrier_sharpness")

# What each registered analytic model implements, by the reference's function names (prb.py:166-204, :379-402):
#   cost:  name -> (model constant that carries its gain or None, "state" = nodes 1..ns | "stage" = nodes 0..ns-1)
#   eq:    equality constraints hard-wired as sqrt(1e6)-weighted penalty rows on nodes 0..ns-1 (ddp.py:195-196, :216-226)
#   ineq:  inequality constraints available as an opt-in barrier (the reference ignores them, ddp.py:197-202)
def _srbd_terms(nc, contact_states):
    cost = {"rz_tracking": ("r_tracking_gain", "state"), "o_tracking_xyz": (None, "state"), "o_tracking_w": (None, "state"),
            "rdot_tracking": ("rdot_tracking_gain", "state"), "w_tracking": ("w_tracking_gain", "state"),
            "min_qddot": ("min_qddot_gain", "stage")}
    eq = []
    if contact_states:
        for nm in ("rel_pos_y_1_4", "rel_pos_x_1_4", "rel_pos_y_3_6", "rel_pos_x_3_6"):
            cost[nm] = ("rel_pos_gain", "state")
        cm = nc // 2                                                           # prb.py:166-170
        eq = [f"relative_vel_left_{i}" for i in range(1, cm)] + [f"relative_vel_right_{i}" for i in range(cm + 1, 2 * cm)] + \
             [f"{t}{i}" for i in range(nc) for t in ("cz_tracking", "cdotxy_tracking")]
    for i in range(nc):
        cost[f"min_f{i}"] = ("min_f_gain", "stage")
        cost[f"f{i}_active"] = ("force_switch_weight", "stage")
    return dict(cost=cost, eq=eq, ineq=[f"f{i}_friction_cone" for i in range(nc)])


MODEL_TERMS = {
    "srbd37": _srbd_terms(4, True),
    "srbd61": _srbd_terms(8, True),
    "srbd13": _srbd_terms(2, False),
    "lip30": dict(cost={"rz_tracking": ("r_tracking_gain", "state"), "rxy_tracking": ("r_tracking_gain", "state"),
                        "rdot_tracking": ("rdot_tracking_gain", "state"), "zmp_tracking": ("zmp_tracking_gain", "stage"),
                        "rel_pos_y_1_4": ("rel_pos_gain", "state"), "rel_pos_x_1_4": ("rel_pos_gain", "state"),
                        "rel_pos_y_3_6": ("rel_pos_gain", "state"), "rel_pos_x_3_6": ("rel_pos_gain", "state"),
                        "min_qddot": ("min_qddot_gain", "stage")},
                  eq=["relative_vel_left_1", "relative_vel_right_3"] + [f"{t}{i}" for i in range(4) for t in ("cz_tracking", "cdotxy_tracking")],
                  ineq=[]),
}


class DDPSolver:
    def __init__(self, prb: Problem, opts: Dict) -> None:
        if prb.model is None:
            raise ValueError("the problem has no registered analytic model (Problem.setModel)")
        self.prb = prb
        self.opts = dict(opts or {})
        unknown = [k for k in self.opts if k not in _REFERENCE_KEYS + _EXTRA_KEYS + _BARRIER_KEYS]
        if unknown:
            raise KeyError(f"unknown DDP option(s): {unknown}")
        barrier = {k: float(self.opts.pop(k)) for k in _BARRIER_KEYS if k in self.opts}
        self.state_var = prb.getState().getVars()
        self.state_size = sum(v.getDim() for v in self.state_var)
        self.input_var = prb.getInput().getVars()
        self.input_size = sum(v.getDim() for v in self.input_var)
        self.param_var = prb.getParameters()
        self._collect_constraints()
        consts = self._model_consts_from_functions()
        if prb.getDt() is not None:
            consts["dt"] = prb.getDt()
        if barrier.get("bound_barrier_weight", 0.0) > 0.0:
            # variable bounds -> entries of z = [x u] in creation order (ddp.py:203-208 walks var_container.getVarList)
            consts.update(barrier)
            consts["lower"] = np.concatenate([v.getLowerBounds() for v in list(self.state_var) + list(self.input_var)])
            consts["upper"] = np.concatenate([v.getUpperBounds() for v in list(self.state_var) + list(self.input_var)])
        self.ddp_solver = DdpEngine(prb.model, prb.nodes - 1, 1, opts=self.opts, consts=consts)
        if (self.ddp_solver.nx, self.ddp_solver.nu) != (self.state_size, self.input_size):
            raise ValueError("problem variables do not match the registered model's dimensions")
        from . import _lib
        self._np_model = _lib.model_dims(prb.model)[2]
        own, user = 0, []
        for p in self.param_var.values():          # the model's own parameters come first (creation order); the rest are the user's
            if own < self._np_model:
                own += p.getDim()
            else:
                user.append(p)
        refs = {id(r[0]) for r in self._extra_refs if r is not None}
        if own != self._np_model or any(id(p) not in refs for p in user):
            raise ValueError("problem parameters do not match the registered model's parameter vector (further parameters must be "
                             "references of declared LinearTerm residuals)")
        if self.ddp_solver.np_ != self._np_model + (8 if self._extra_refs else 0):
            raise ValueError("engine / problem parameter widths disagree")
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
        params = self._parameter_matrix()[None]                             # ddp.py:98-99, vectorised
        x, u = self.ddp_solver.solve(params)                                # ddp.py:101
        x, u = np.ascontiguousarray(x[0].T), np.ascontiguousarray(u[0].T)   # reference layout [dim, nodes]
        self.var_solution = self._createVarSolDict(x, u)
        self.var_solution["x_opt"] = x                                      # ddp.py:103
        self.var_solution["u_opt"] = u                                      # ddp.py:104
        self.stats = self.ddp_solver.stats[0]
        return bool(self.stats["converged"])                                # ddp.py:106

    def solve_receding(self, x0) -> bool:
        """One tick of the receding-horizon loop with device-resident data.

        Since the previous call the caller normally has shifted every parameter back by one node and assigned node N -- which is
        all dsrbd_example.py:102-131 / wpg.py:74-99 ever do.  Then only the last parameter column and ``x0`` cross PCIe; the
        engine shifts its resident parameter tensor and warm-starts from its previous solution advanced by one knot.  A host
        shadow of the resident tensor checks that: any other change to the parameters re-uploads the whole tensor (``resyncs``).  Same
        result as ``setInitialState(x0)`` + shifted ``set_*_warmstart`` + ``solve()`` (tests/test_gpu_api.py)."""
        x0 = np.asarray(x0, dtype=float).reshape(1, self.state_size)
        self._x0 = x0.copy()
        pm = self._parameter_matrix()
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
            self._shadow = pm.copy()
            self.resyncs = 0
        else:
            # host shadow of the resident tensor: the device shifts EVERY parameter column back by one node; if the caller has
            # assigned anything else below node N since the last tick (or did not shift a parameter the device shifts), the two
            # would drift apart silently -- re-upload the whole tensor instead (and count it: `resyncs`)
            self._shadow[:-1] = self._shadow[1:]
            self._shadow[-1] = pm[-1]
            if np.array_equal(self._shadow, pm):
                self.ddp_solver.advance(pm[-1][None], x0)
            else:
                self.resyncs += 1
                self._shadow = pm.copy()
                self.ddp_solver.advance(pm[-1][None], x0)             # warm start and state advance as usual ...
                self.ddp_solver.set_params(pm[None])                  # ... then the parameters as the host has them
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

    def _collect_constraints(self):
        """ddp.py:38-48: equality constraints (coincident bounds) apart from inequality constraints."""
        self.var_container = self.prb.var_container
        self.fun_container = self.prb.function_container
        self.equality_constraints = []
        self.inequality_constraints = []
        for constr in self.fun_container.getCnstr().values():
            if self.is_equality_constraint(constr):
                self.equality_constraints.append(constr)
            else:
                self.inequality_constraints.append(constr)

    def _model_consts_from_functions(self):
        """Constants of the registered model from the problem's cost terms and constraints -- the counterpart of get_L /
        get_L_term (ddp.py:179-226), which sum whatever the function container holds.  A declared residual sets the gain of
        its term (removing it switches the term off); equality constraints must be exactly the penalties the model hard-wires;
        inequality constraints are ignored like the reference ignores them (ddp.py:197-202 is commented out) unless the model's
        opt-in barrier is on.  Anything the analytic model cannot express raises instead of being silently dropped."""
        prb = self.prb
        consts = dict(prb.model_consts)
        table = MODEL_TERMS[prb.model]
        ns = prb.nodes - 1
        ranges = {"state": list(range(1, ns + 1)), "stage": list(range(0, ns))}
        cost = self.fun_container.getCost()
        if not cost and not self.fun_container.getCnstr():
            return consts                                       # a problem without declarations: the model's built-in defaults
        self._extra_refs = []
        linear = {n: fn for n, fn in cost.items() if isinstance(fn.term, LinearTerm)}
        if linear:
            consts["extra_rows"] = self._linear_rows(linear, ranges)
        unknown = [n for n in cost if n not in table["cost"] and n not in linear]
        if unknown:
            raise NotImplementedError(f"model {prb.model} has no analytic term for residual(s) {unknown}: declare a linear residual "
                                      "as problem.LinearTerm, anything else needs a model term")
        gains = {}
        for name, (ckey, kind) in table["cost"].items():
            fn = cost.get(name)
            if fn is None:
                if ckey is None:
                    raise NotImplementedError(f"residual {name!r} cannot be removed from model {prb.model} (its weight is the "
                                              "orientation_tracking_gain parameter: assign 0 to switch it off)")
                gains.setdefault(ckey, []).append(0.0)
                continue
            if fn.getNodes() != ranges[kind]:
                raise NotImplementedError(f"residual {name!r}: model {prb.model} implements it on nodes {ranges[kind][0]}..{ranges[kind][-1]} only")
            if ckey is not None:
                gains.setdefault(ckey, []).append(float(fn.term.gain))
        for ckey, vals in gains.items():
            if len(set(vals)) != 1:
                raise NotImplementedError(f"model {prb.model} has ONE gain {ckey} for {len(vals)} residuals; got {sorted(set(vals))}")
            consts[ckey] = vals[0]
        eq_names = [c.getName() for c in self.equality_constraints]
        # the relative-velocity constraints inside a foot are a runtime switch of the model (prb.py:166 declares them only
        # `if contact_model > 1`): all of them or none
        rel = [n for n in table["eq"] if n.startswith("relative_vel_")]
        have_rel = [n for n in eq_names if n.startswith("relative_vel_")]
        if rel and not have_rel:
            consts["relative_velocity_constraints"] = 0
            expected = [n for n in table["eq"] if n not in rel]
        else:
            if rel:
                consts["relative_velocity_constraints"] = 1
            expected = table["eq"]
        if sorted(eq_names) != sorted(expected):
            raise NotImplementedError(f"model {prb.model} implements the equality constraints {table['eq']} (the relative_vel_* ones "
                                      f"all or none); the problem holds {eq_names}")
        for c in self.equality_constraints:
            if c.getNodes() != list(range(0, ns + 1)):
                raise NotImplementedError(f"constraint {c.getName()!r}: only the default node range is implemented")
        bad = [c.getName() for c in self.inequality_constraints if c.getName() not in table["ineq"]]
        if bad:
            raise NotImplementedError(f"model {prb.model} has no barrier for inequality constraint(s) {bad}")
        return consts

    def _linear_rows(self, linear, ranges):
        """User-declared linear residuals -> the extra rows of the model's "_x" build (include/sddp.h extra_*): coefficient vector
        over z = [x u] in creation order, weight = the gain, kind from the node range; self._extra_refs[j] = (Parameter, row) whose
        per-node values fill reference column j of the widened parameter vector."""
        off, o = {}, 0
        for v in list(self.state_var) + list(self.input_var):
            off[v] = o
            o += v.getDim()
        nz = self.state_size + self.input_size
        rows = []
        for name, fn in linear.items():
            t = fn.term
            kind = "state" if fn.getNodes() == ranges["state"] else ("stage" if fn.getNodes() == ranges["stage"] else None)
            if kind is None:
                raise NotImplementedError(f"residual {name!r}: a linear residual lives on nodes 1..N (state term) or 0..N-1 (stage term)")
            for r in range(t.dim):
                a = np.zeros(nz)
                for v, A in t.coeffs.items():
                    if v not in off:
                        raise ValueError(f"residual {name!r}: {v} is not a state or input variable of this problem")
                    a[off[v]:off[v] + v.getDim()] += A[r]
                if kind == "state" and np.any(a[self.state_size:] != 0.0):
                    raise NotImplementedError(f"residual {name!r}: a term on nodes 1..N is active at the terminal node and cannot touch the inputs")
                rows.append(dict(a=a, w=t.gain, kind=kind, const=float(t.const[r])))
                self._extra_refs.append(None if t.ref is None else (t.ref, r))
        if len(rows) > 8:
            raise NotImplementedError(f"{len(rows)} linear residual rows declared; the analytic models carry at most 8")
        return rows

    def _parameter_matrix(self):
        """[N+1, np]: the model's own parameters (ddp.py:165-177, creation order); with user rows, 8 more columns: their references"""
        P = self.prb.parameter_matrix()
        if not self._extra_refs:
            return P
        out = np.zeros((P.shape[0], self._np_model + 8))
        out[:, :self._np_model] = P[:, :self._np_model]
        for j, ref in enumerate(self._extra_refs):
            if ref is not None:
                out[:, self._np_model + j] = ref[0].values[ref[1], :]
        return out

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
