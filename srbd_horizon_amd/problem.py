"""A small Horizon-shaped problem surface (the part of ``horizon.problem.Problem`` the reference's DDP path touches).

The reference builds symbolic CasADi graphs through Horizon (reference python/prb.py:21, :32-72, :110, :160-163) and
its adapter walks them (python/ddp.py:38-58, :125-151, :165-177).  Here the dynamics and costs are *registered
analytic HIP models* (``Problem.setModel``); the surface keeps the call shapes the example loops and the walking-pattern
scheduler use: ``createStateVariable / createInputVariable / createParameter``, ``Parameter.assign / getValues``,
``getState().getVars()``, ``getInput().getVars()``, ``getParameters()``, ``getDt()``, ``nodes``,
``var_container.getVarList(offset=False)``.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np


class Variable:
    def __init__(self, name: str, dim: int, kind: str):
        self._name, self._dim, self.kind = name, int(dim), kind

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
