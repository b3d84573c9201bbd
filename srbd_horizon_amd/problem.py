"""A small Horizon-shaped problem surface (the part of ``horizon.problem.Problem`` the reference's DDP path touches).

The reference builds symbolic CasADi graphs through Horizon (reference python/prb.py:21, :32-72, :110, :160-163) and
its adapter walks them (python/ddp.py:38-58, :125-151, :165-177).  Here the dynamics and costs are *registered
analytic HIP models* (``Problem.setModel``); the surface keeps the call shapes the example loops and the walking-pattern
scheduler use: ``createStateVariable / createInputVariable / createParameter``, ``Parameter.assign / getValues``,
``getState().getVars()``, ``getInput().getVars()``, ``getParameters()``, ``getDt()``, ``nodes``,
``var_container.getVarList(offset=False)``, and the function container the adapter reads its costs and constraints from:
``createResidual / createConstraint / createIntermediateConstraint``, ``function_container.getCost() / getCnstr()``,
``getNodes() / getLowerBounds() / getUpperBounds()`` (python/prb.py:166-204; python/ddp.py:38-48, :184-196, :218-224).
Where the reference passes a CasADi expression, the builders here pass a ``Term``: the name of an analytic term of the
registered model plus the gain it is scaled with.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np


class Variable:
    def __init__(self, name: str, dim: int, kind: str):
        self._name, self._dim, self.kind = name, int(dim), kind
        self._lb, self._ub = np.full(self._dim, -np.inf), np.full(self._dim, np.inf)     # Horizon's default: unbounded

    # Horizon's variable bounds (same for every node here).  The reference's DDP adapter would turn them into exponential barriers
    # (ddp.py:203-208) but that block is commented out, and prb.py sets none: they only act with the solver option
    # "bound_barrier_weight" > 0 (srbd_horizon_amd/ddp.py)
    def setBounds(self, lb, ub, nodes=None):
        if nodes is not None:
            raise NotImplementedError("node-dependent variable bounds are not implemented")
        self.setLowerBounds(lb)
        self.setUpperBounds(ub)

    def setLowerBounds(self, lb, nodes=None):
        if nodes is not None:
            raise NotImplementedError("node-dependent variable bounds are not implemented")
        self._lb = np.broadcast_to(np.asarray(lb, dtype=float).reshape(-1), (self._dim,)).copy()

    def setUpperBounds(self, ub, nodes=None):
        if nodes is not None:
            raise NotImplementedError("node-dependent variable bounds are not implemented")
        self._ub = np.broadcast_to(np.asarray(ub, dtype=float).reshape(-1), (self._dim,)).copy()

    def getLowerBounds(self):
        return self._lb.copy()

    def getUpperBounds(self):
        return self._ub.copy()

    def getBounds(self):
        return self._lb.copy(), self._ub.copy()

    def getName(self):
        return self._name

    def getDim(self):
        return self._dim

    def size(self):
        return (self._dim, 1)

    @property
    def shape(self):
        return (self._dim, 1)

    def __repr__(self):
        return f"{self.kind}:{self._name}[{self._dim}]"


class Parameter:
    """Per-node parameter; value matrix ``[dim, nodes]`` (Horizon ``Parameter.assign/getValues`` semantics as used by
    reference python/wpg.py:76-99 and python/dsrbd_example.py:103-122)."""

    def __init__(self, name: str, dim: int, nodes: int):
        self._name, self._dim = name, int(dim)
        self.values = np.zeros((self._dim, int(nodes)))

    def getName(self):
        return self._name

    def getDim(self):
        return self._dim

    def assign(self, val, nodes=None):
        v = np.asarray(val, dtype=float)
        if nodes is None:
            idx = np.arange(self.values.shape[1])
        else:
            idx = np.atleast_1d(np.asarray(list(nodes) if isinstance(nodes, range) else nodes, dtype=int))
        if v.ndim == 2 and v.shape == (self._dim, idx.size):
            self.values[:, idx] = v
        else:
            v = v.reshape(-1)
            if v.size == 1:
                self.values[:, idx] = v[0]
            elif v.size == self._dim:
                self.values[:, idx] = v[:, None]
            else:
                raise ValueError(f"{self._name}: cannot assign {v.size} values to dim {self._dim}")

    def getValues(self, nodes=None):
        if nodes is None:
            return self.values.copy()
        idx = np.atleast_1d(np.asarray(list(nodes) if isinstance(nodes, range) else nodes, dtype=int))
        return self.values[:, idx].copy()


class ParameterRow:
    """One row of a Parameter presented as a 1-dim Parameter (lets the 2-contact metric model hand the z row of its
    contact-position parameters to the walking-pattern scheduler as ``c_ref``)."""

    def __init__(self, par: Parameter, row: int):
        self.par, self.row = par, int(row)

    def getName(self):
        return f"{self.par.getName()}[{self.row}]"

    def getDim(self):
        return 1

    @property
    def values(self):
        return self.par.values[self.row:self.row + 1, :]

    def assign(self, val, nodes=None):
        v = np.asarray(val, dtype=float).reshape(-1)
        if nodes is None:
            self.par.values[self.row, :] = v if v.size > 1 else v[0]
        else:
            idx = np.atleast_1d(np.asarray(list(nodes) if isinstance(nodes, range) else nodes, dtype=int))
            self.par.values[self.row, idx] = v if v.size == idx.size else v[0]

    def getValues(self, nodes=None):
        if nodes is None:
            return self.par.values[self.row:self.row + 1, :].copy()
        idx = np.atleast_1d(np.asarray(list(nodes) if isinstance(nodes, range) else nodes, dtype=int))
        return self.par.values[self.row:self.row + 1, idx].copy()


class Aggregate:
    """horizon.variables.Aggregate look-alike (prb.py:34-36)."""

    def __init__(self):
        self._vars = []

    def addVariable(self, v):
        self._vars.append(v)

    def getVars(self):
        return list(self._vars)

    def size(self):
        return (sum(v.getDim() for v in self._vars), 1)


class Term:
    """Stands where the symbolic expression of ``createResidual`` / ``createConstraint`` stands (prb.py:166-204): ``key`` names
    an analytic term the registered HIP model implements, ``gain`` is the weight the reference puts under the square root in front
    of it (``consts_key`` is the model constant that carries it; None: the term has no tunable gain)."""

    def __init__(self, key: str, consts_key: str | None = None, gain: float | None = None, dim: int = 1):
        self.key, self.consts_key, self.gain, self.dim = key, consts_key, gain, int(dim)

    def __repr__(self):
        return f"Term({self.key!r}, {self.consts_key}={self.gain})"


class LinearTerm(Term):
    """A user-declared LINEAR residual -- what ``createResidual(name, sqrt(gain) * (A_1 @ v_1 + A_2 @ v_2 + ... - ref))`` is in the
    reference, where ``get_L`` / ``get_L_term`` sum whatever residual the container holds (ddp.py:183-196, :216-226).  The analytic
    models take up to 8 such rows on top of prb.py's own terms (include/sddp.h ``extra_*``).

    coeffs: {Variable: matrix [dim, variable dim]} over state and input variables; gain: the weight under the square root;
    ref: a Parameter of dimension dim created for this term (its per-node values are the reference), or None; const: constant part
    of the reference (scalar or [dim]).  Node range: 1..N (a state term like prb.py's tracking terms; states only) or 0..N-1 (a
    stage term like min_qddot)."""

    def __init__(self, coeffs: dict, gain: float, ref=None, const=0.0):
        mats = {v: np.atleast_2d(np.asarray(A, dtype=float)) for v, A in coeffs.items()}
        dims = {A.shape[0] for A in mats.values()}
        if len(dims) != 1:
            raise ValueError("LinearTerm: every coefficient matrix needs the same number of rows")
        dim = dims.pop()
        for v, A in mats.items():
            if not isinstance(v, Variable) or A.shape[1] != v.getDim():
                raise ValueError("LinearTerm: coeffs maps state / input Variables to [dim, variable dim] matrices")
        if ref is not None and ref.getDim() != dim:
            raise ValueError("LinearTerm: the reference parameter needs one entry per row")
        if not (float(gain) >= 0.0):
            raise ValueError("LinearTerm: gain must be >= 0")
        super().__init__("linear", None, float(gain), dim)
        self.coeffs, self.ref = mats, ref
        self.const = np.broadcast_to(np.asarray(const, dtype=float).reshape(-1), (dim,)).copy()


class Function:
    """A cost term or constraint of the problem (Horizon ``Function`` / ``Constraint`` as ddp.py:42-48, :184-196 use them)."""

    def __init__(self, name: str, term: Term, nodes, lb=None, ub=None):
        self._name, self.term = name, term
        self._nodes = [int(n) for n in nodes]
        self._lb = None if lb is None else np.broadcast_to(np.asarray(lb, dtype=float), (term.dim,)).copy()
        self._ub = None if ub is None else np.broadcast_to(np.asarray(ub, dtype=float), (term.dim,)).copy()

    def getName(self):
        return self._name

    def getNodes(self):
        return list(self._nodes)

    def getDim(self):
        return self.term.dim

    def getLowerBounds(self):
        return self._lb.copy()

    def getUpperBounds(self):
        return self._ub.copy()


class _FunContainer:
    def __init__(self):
        self._cost, self._cnstr = OrderedDict(), OrderedDict()

    def getCost(self):
        return self._cost

    def getCnstr(self):
        return self._cnstr


class _VarContainer:
    def __init__(self, prb):
        self._prb = prb

    def getVarList(self, offset=False):
        return list(self._prb._vars)


class Problem:
    def __init__(self, N: int):
        self.N = int(N)
        self.nodes = self.N + 1                    # ddp.py:83 iterates range(0, prb.nodes-1)
        self._vars = []
        self._state = Aggregate()
        self._input = Aggregate()
        self._params = OrderedDict()
        self._dt = None
        self.model = None
        self.model_consts = {}
        self.var_container = _VarContainer(self)
        self.function_container = _FunContainer()

    # ---- costs and constraints (prb.py:166-204).  Default nodes as Horizon's: every node ------------------------------------
    def createResidual(self, name, term: Term, nodes=None):
        if not isinstance(term, Term):
            raise TypeError("createResidual needs a Term naming an analytic term of the registered model (no CasADi here)")
        f = Function(name, term, range(self.nodes) if nodes is None else nodes)
        self.function_container._cost[name] = f
        return f

    def createConstraint(self, name, term: Term, nodes=None, bounds=None):
        if not isinstance(term, Term):
            raise TypeError("createConstraint needs a Term naming an analytic term of the registered model (no CasADi here)")
        b = bounds or {}
        f = Function(name, term, range(self.nodes) if nodes is None else nodes, b.get("lb", 0.0), b.get("ub", 0.0))   # default: g = 0
        self.function_container._cnstr[name] = f
        return f

    def createIntermediateConstraint(self, name, term: Term, nodes=None, bounds=None):
        return self.createConstraint(name, term, range(self.nodes - 1) if nodes is None else nodes, bounds)

    def removeCostFunction(self, name):
        return self.function_container._cost.pop(name, None) is not None

    def removeConstraint(self, name):
        return self.function_container._cnstr.pop(name, None) is not None

    def createStateVariable(self, name, dim):
        if any(v.kind == "input" for v in self._vars):
            raise RuntimeError("states must be created before inputs (ddp.py:125-151 relies on it)")
        v = Variable(name, dim, "state")
        self._vars.append(v)
        self._state.addVariable(v)
        return v

    def createInputVariable(self, name, dim):
        v = Variable(name, dim, "input")
        self._vars.append(v)
        self._input.addVariable(v)
        return v

    def createParameter(self, name, dim):
        p = Parameter(name, dim, self.nodes)
        self._params[name] = p
        return p

    def getState(self):
        return self._state

    def getInput(self):
        return self._input

    def getParameters(self):
        return self._params

    def getNNodes(self):
        return self.nodes

    def setDt(self, dt):
        self._dt = float(dt)

    def getDt(self):
        return self._dt

    def setModel(self, name: str, consts: dict):
        """Register the analytic HIP model that implements this problem's dynamics and costs."""
        self.model = name
        self.model_consts = dict(consts)

    def parameter_matrix(self):
        """[N+1, np]: every Parameter flattened row by row in creation order -- the vectorised form of the reference's
        per-node Python loop ``get_params_value`` (ddp.py:165-177)."""
        return np.ascontiguousarray(np.vstack([p.values for p in self._params.values()]).T)
