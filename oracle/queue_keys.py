#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (it runs the C oracle).  What is knowable about an instance's iteration count BEFORE its first iteration, and what each queue key buys (DESIGN.md
section 5, "work queue").  CPU only: iteration counts from the plain-C oracle (test infrastructure; identical to
the GPU's on > 99.8 % of the instances), makespans from the list-scheduling model of tools/queue_sim.py (2048 slots).

Keys are FITTED on seed blocks 0..19 and EVALUATED on the held-out blocks 60..79 (bench.py times 0..59 at --steps 20).

    python oracle/queue_keys.py            # table in profiles/r05/queue_keys.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from queue_sim import R, S, simulate  # noqa: E402

N, B = 30, 1024
OPTS = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)


def collect(blocks):
    from oracle import cport, ddp as oddp, models as omodels
    from srbd_horizon_amd import workload
    seeds = np.concatenate([b * B + np.arange(B) for b in blocks])
    batch = workload.make_srbd13_batch(N, seeds)
    cst = omodels.RobotConsts(**batch["consts"])
    a = (batch["x0"], batch["params"], batch["xs"], batch["us"])
    thr = min(16, os.cpu_count() or 1)
    out = {}
    _, _, s = cport.solve_batch(cst, oddp.DdpOptions(**OPTS), *a, threads=thr)
    out["iters"] = s[:, 1].astype(int)
    for k in (0, 1, 3):                                   # what a probe of k iterations would know
        _, _, s = cport.solve_batch(cst, oddp.DdpOptions(**dict(OPTS, max_iters=k)), *a, threads=thr)
        out[f"J{k}"], out[f"a{k}"], out[f"gap{k}"] = s[:, 0], s[:, 3], s[:, 4]
    out["labels"], out["n_classes"] = workload.srbd13_schedule_classes(batch["params"])
    out["labels_unsigned"], _ = workload.srbd13_schedule_classes(batch["params"], signed=False)
    out["rdref"] = np.linalg.norm(batch["params"][:, N, 0:3], axis=1)
    return out


def ratio(d, key):
    it = d["iters"]
    ideal = float(np.sum(it * (S + R) + S)) / 2048
    return simulate(it, np.argsort(-np.asarray(key, dtype=float), kind="stable")) / ideal


def main():
    tr, te = collect(range(0, 20)), collect(range(60, 80))
    rows = []
    rows.append(("index order", ratio(tr, -np.arange(len(tr["iters"]))), ratio(te, -np.arange(len(te["iters"])))))
    rows.append(("exact foreknowledge (replay)", ratio(tr, tr["iters"]), ratio(te, te["iters"])))
    rows.append(("initial cost J0 (queue_order 2)", ratio(tr, tr["J0"]), ratio(te, te["J0"])))
    rows.append(("initial defect 1-norm", ratio(tr, tr["gap0"]), ratio(te, te["gap0"])))
    rows.append(("|rdot_ref(N)|", ratio(tr, tr["rdref"]), ratio(te, te["rdref"])))
    for lab, lname in (("labels_unsigned", "classes without the sign of the command: "), ("labels", "")):
        for stat, fn in (("mean", np.mean), ("90 % quantile", lambda v: np.quantile(v, 0.9)), ("maximum", np.max)):
            tab = {c: fn(tr["iters"][tr[lab] == c]) for c in np.unique(tr[lab])}
            for tie in (False, True):
                def key(d):
                    k = np.array([tab.get(c, 1e6) for c in d[lab]], dtype=float)
                    return k + (1e-3 * d["J0"] / (np.abs(d["J0"]) + 1e9) if tie else 0.0)
                rows.append((f"{lname}class {stat} of the training blocks{' + J0 tie-break (queue_order 3)' if tie and stat == 'mean' and not lname else (' + J0 tie-break' if tie else '')}",
                             ratio(tr, key(tr)), ratio(te, key(te))))
    # the same with what the bench's warm-up gives: class means over 5 blocks only
    w = collect(range(20, 25))
    for lab, lname in (("labels_unsigned", "classes without the sign of the command: "), ("labels", "")):
        tab = {c: np.mean(w["iters"][w[lab] == c]) for c in np.unique(w[lab])}
        glob = float(np.mean(w["iters"]))
        rows.append((f"{lname}class mean over 5 blocks (the bench's warm-up) + J0 tie-break", float("nan"),
                     ratio(te, np.array([tab.get(c, glob) for c in te[lab]]) + 1e-3 * te["J0"] / (np.abs(te["J0"]) + 1e9))))
    rows.append(("cost after ONE probe iteration of every instance", ratio(tr, tr["J1"]), ratio(te, te["J1"])))
    rows.append(("cost after THREE probe iterations", ratio(tr, tr["J3"]), ratio(te, te["J3"])))
    try:
        from sklearn.ensemble import HistGradientBoostingRegressor

        def X(d):
            return np.c_[np.log(d["J0"]), d["gap0"], d["rdref"], d["labels"]]
        m = HistGradientBoostingRegressor(max_iter=300, learning_rate=0.05).fit(X(tr), np.log(tr["iters"] + 1.0))
        rows.append(("boosted trees on (J0, defect norm, |rdot_ref|, class)", ratio(tr, m.predict(X(tr))), ratio(te, m.predict(X(te)))))
    except ImportError:
        pass
    print(f"makespan / ideal on 20 480 instances, 2048 slots (iterations: mean {te['iters'].mean():.2f}, max {te['iters'].max()}; "
          f"{len(np.unique(te['labels']))} classes occur)")
    print(f"{'queue key':100s} {'blocks 0-19 (fit)':>18s} {'blocks 60-79 (held out)':>24s}")
    for name, a, b in rows:
        print(f"{name:100s} {a:18.3f} {b:24.3f}")


if __name__ == "__main__":
    main()
