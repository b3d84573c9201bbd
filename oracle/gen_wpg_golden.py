"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/wpg_*.npz by RUNNING the reference's own walking-pattern
scheduler (``/root/reference/python/wpg.py``, the only reference module importable offline)
with a duck-typed Horizon ``Parameter``.  Only numeric outputs are stored; the reference source never enters
this repository.  Run in the build container only (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_wpg_golden.py
"""
import importlib.util
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference/python/wpg.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class FakeParameter:
    """Minimal stand-in for horizon.variables.Parameter: value matrix [dim, nodes]."""

    def __init__(self, dim, nodes, init=0.0):
        self.v = np.full((dim, nodes), float(init))

    def assign(self, val, nodes=None):
        val = np.asarray(val, dtype=float).reshape(-1)
        if nodes is None:
            self.v[:, :] = val[:, None]
        else:
            for n in np.atleast_1d(nodes):
                self.v[:, int(n)] = val

    def getValues(self, nodes=None):
        if nodes is None:
            return self.v.copy()
        return self.v[:, np.atleast_1d(nodes)].copy()


def run(ns, actions, c_init_z, nc=4, contact_model=2, number_of_legs=2):
    spec = importlib.util.spec_from_file_location("ref_wpg", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    c_ref = {i: FakeParameter(1, ns + 1, c_init_z) for i in range(nc)}
    sw = {i: FakeParameter(1, ns + 1, 1.0) for i in range(nc)}
    w_ref = FakeParameter(3, ns + 1, 0.0)
    otg = FakeParameter(1, ns + 1, 1e1)
    dummy = {i: None for i in range(nc)}
    gen = ref.steps_phase(dummy, dummy, dummy, c_init_z, c_ref, w_ref, otg, sw, ns,
                          number_of_legs=number_of_legs, contact_model=contact_model)
    tables = dict(l_cycle=np.array(gen.l_cycle), r_cycle=np.array(gen.r_cycle),
                  l_cdot_switch=np.array(gen.l_cdot_switch), r_cdot_switch=np.array(gen.r_cdot_switch),
                  step_nodes=np.array(gen.step_nodes))
    hist = dict(c_ref=[], cdot_switch=[], w_ref=[], otg=[])
    for a in actions:
        gen.set(a)
        hist["c_ref"].append(np.vstack([c_ref[i].v for i in range(nc)]))
        hist["cdot_switch"].append(np.vstack([sw[i].v for i in range(nc)]))
        hist["w_ref"].append(w_ref.v.copy())
        hist["otg"].append(otg.v.copy())
    out = {k: np.array(v) for k, v in hist.items()}
    out.update(tables)
    out["actions"] = np.array(actions)
    out["c_init_z"] = np.array(c_init_z)
    out["ns"] = np.array(ns)
    out["number_of_legs"], out["contact_model"] = np.array(number_of_legs), np.array(contact_model)
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    mixed = ["standing"] * 3 + ["step"] * 27 + ["jump"] * 4 + ["standing"] * 3 + ["step"] * 8
    for ns in (20, 30, 60):
        np.savez_compressed(os.path.join(OUT, f"wpg_step_ns{ns}.npz"), **run(ns, ["step"] * 45, 0.0))
        np.savez_compressed(os.path.join(OUT, f"wpg_mixed_ns{ns}.npz"), **run(ns, mixed, 0.02))
    # the other contact configurations the problem builder accepts (prb.py:39-41): eight points (contact_model = 4, the default in
    # the code) and four point feet (number_of_legs = 4 x contact_model = 1)
    np.savez_compressed(os.path.join(OUT, "wpg_mixed_ns20_l2c4.npz"), **run(20, mixed, 0.02, nc=8, contact_model=4, number_of_legs=2))
    np.savez_compressed(os.path.join(OUT, "wpg_mixed_ns20_l4c1.npz"), **run(20, mixed, 0.02, nc=4, contact_model=1, number_of_legs=4))
    print("wrote", sorted(f for f in os.listdir(OUT) if f.startswith("wpg_")))


if __name__ == "__main__":
    main()
