"""Soak run of the whole-batch parity statement of tests/test_gpu_divergence.py on instances NO test and no bench region ever
touches (seed blocks 60 .. 60 + n): not collected by pytest (minutes of GPU time), run by hand on the GPU box:

    python tests/soak_parity.py [first_block] [n_chunks] [blocks_per_chunk] [model] [N] [option=int ...]  ->  one JSON line per chunk + a total

Asserts exactly what the test asserts (end-to-end parity on the same path; one-step shadowing of every accepted GPU step on the
others) and reports the counts beside the number of instances on which the two CPU builds of the oracle split."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cport, ddp as oddp, models as omodels  # noqa: E402
from srbd_horizon_amd import workload  # noqa: E402
from tests import shadow  # noqa: E402
from tests.test_gpu_divergence import OPTS, THREADS, assert_batch  # noqa: E402


def main(first, chunks, per, model="srbd13", N=30, over=None):
    B = 1024
    OPTS.update(over or {})                                  # e.g. second_order=2: solver options of both sides
    tot = dict(instances=0, other_path=0, cpu_pair=0, both=0, unconverged_oracle=0, unconverged_gpu=0, other_optimum=0, failed_chunks=0, worst_end_linf=0.0, worst_same_linf=0.0)
    for c in range(chunks):
        blocks = range(first + c * per, first + (c + 1) * per)
        seeds = np.concatenate([b * B + np.arange(B) for b in blocks])
        t0 = time.time()
        batch = workload.make_batch(model, N, seeds)
        res = shadow.check_batch(model, N, batch, OPTS, dict(waves_per_simd=2, queue_order=2), omodels.RobotConsts(**batch["consts"]),
                                 threads=THREADS)
        failed = None
        try:
            assert_batch(res, f"soak_{blocks[0]}_{blocks[-1]}", lambda *a, **k: None, same_optimum=False)
        except AssertionError as e:                              # keep going: the report names the chunk and the record
            failed = str(e)[:4000]
        # instances where both converge, but not to the same point: what do the two CPU builds of the oracle do on them?
        other = [r for r in res["explained"] if r["gpu_status"] == 0 and r["oracle_status"] == 0 and r["end_linf"] > 1e-4]
        other_rec = []
        for r in other:
            i = r["instance"]
            a = [batch[k][i:i + 1] for k in ("x0", "params", "xs", "us")]
            cst = omodels.RobotConsts(**batch["consts"])
            xf, uf, sf = cport.solve_batch(cst, oddp.DdpOptions(**OPTS), *a, threads=1, variant="fast", model=model)
            sp = r["split_gpu"] or {}
            other_rec.append(dict(seed=int(seeds[i]), it=(r["gpu_iters"], r["oracle_iters"], r["oracle_fast_iters"]), end_linf=r["end_linf"],
                                  end_rel_cost=r["end_rel_cost"], split_step=sp.get("step"), drift_before=sp.get("drift_before"),
                                  shadow_max_rel_cost=float(r["shadow"]["max_rel_cost"]), shadow_violations=len(r["shadow"]["violations"]),
                                  cpu_builds_end_linf=float(np.max(np.abs(xf[0] - res["xo"][i]))), cpu_fast_status=int(sf[0, 6])))
            # the split must come after a visible drift, with every GPU step shadowed (assert_batch has checked the latter)
            if not (sp.get("drift_before") is not None and sp["drift_before"] >= 1e-9):
                failed = (failed or "") + f" | other optimum without a visible drift before the split: {r}"[:2000]
        same, so, st = res["same"], res["so"], res["st"]
        conv = same & (so[:, 2] == 1)
        rec = dict(blocks=[blocks[0], blocks[-1]], instances=len(seeds), other_path=len(res["explained"]), cpu_pair=int(res["n_cpu_pair"]),
                   both=len(set(r["instance"] for r in res["explained"]) & set(res["cpu_pair_idx"].tolist())), unconverged_oracle=int((so[:, 2] == 0).sum()), unconverged_gpu=int((st["converged"] == 0).sum()),
                   worst_same_linf=float(max(np.max(np.abs(res["x"][conv] - res["xo"][conv])), np.max(np.abs(res["u"][conv] - res["uo"][conv])))),
                   worst_end_linf=float(max([r["end_linf"] for r in res["explained"] if r["gpu_status"] == 0 and r["oracle_status"] == 0
                                             and r["end_linf"] <= 1e-4] or [0.0])),
                   other_optimum=other_rec, failed=failed,
                   seconds=round(time.time() - t0, 1))
        print(json.dumps(rec), flush=True)
        for k in ("instances", "other_path", "cpu_pair", "both", "unconverged_oracle", "unconverged_gpu"):
            tot[k] += rec[k]
        tot["other_optimum"] += len(other_rec)
        tot["failed_chunks"] += failed is not None
        tot["worst_end_linf"] = max(tot["worst_end_linf"], rec["worst_end_linf"])
        tot["worst_same_linf"] = max(tot["worst_same_linf"], rec["worst_same_linf"])
    print(json.dumps(dict(total=tot)), flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 60, int(a[1]) if len(a) > 1 else 10, int(a[2]) if len(a) > 2 else 20, a[3] if len(a) > 3 else "srbd13",
         int(a[4]) if len(a) > 4 else 30, {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a[5:]})
