"""tests/golden/nlp_*.npz: optima of the discrete SRBD / LIP problems found by direct transcription + trust-constr + Newton-KKT on the
sympy restatement (oracle/gen_nlp_golden.py) -- no DDP code involved.  The stand-in for north_star's "CasADi solve"."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OPTS = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)      # dsrbd_example.py:55-58
LINF_TOL = 1e-4          # BASELINE.json north_star: <= 1e-4 trajectory l-inf error
COST_RTOL = 1e-6


def load_all():
    out = []
    for f in sorted(glob.glob(os.path.join(GOLDEN, "nlp_*.npz"))):
        z = np.load(f)
        d = {k: z[k] for k in z.files}
        d["model"], d["N"], d["seed"] = str(z["model"]), int(z["N"]), int(z["seed"])
        c = json.loads(str(z["consts_json"]))
        d["consts"] = {k: (v[0] if len(v) == 1 else np.asarray(v)) for k, v in c.items()}
        d["consts"]["inertia_mode"] = int(d["consts"]["inertia_mode"])
        if "relative_velocity_constraints" in d["consts"]:
            d["consts"]["relative_velocity_constraints"] = int(d["consts"]["relative_velocity_constraints"])
        if "extra_rows_json" in z.files:                     # user-declared linear rows (the params carry 8 more columns)
            d["consts"]["extra_rows"] = tuple(dict(r, a=np.asarray(r["a"])) for r in json.loads(str(z["extra_rows_json"])))
        d["consts"]["I"] = d["consts"]["I"].reshape(3, 3)
        d["consts"]["feet"] = d["consts"]["feet"].reshape(-1, 3)
        d["name"] = os.path.basename(f)[:-4]
        out.append(d)
    return out


def check(fx, x, u, cost):
    ex, eu = float(np.max(np.abs(x - fx["x"]))), float(np.max(np.abs(u - fx["u"])))
    rc = abs(cost - float(fx["cost"])) / abs(float(fx["cost"]))
    assert float(fx["kkt_stationarity_rel"]) <= 1e-10 and float(fx["kkt_feasibility"]) <= 1e-10     # the fixture IS a KKT point
    assert ex <= LINF_TOL and eu <= LINF_TOL, (fx["name"], ex, eu)
    assert rc <= COST_RTOL, (fx["name"], rc)
    return ex, eu, rc
