"""Both CPU oracles against the oracle-independent NLP fixtures (tests/nlp_fixtures.py): from the same start the DDP iteration must
land on the KKT point that trust-constr + Newton found for the sympy-transcribed problem -- trajectories to 1e-4 l-inf
(north_star's tolerance), cost to 1e-6 relative."""
import numpy as np
import pytest

from oracle import cport, ddp as oddp, models as omodels
from tests import nlp_fixtures as nf

FIXTURES = nf.load_all()


def test_fixture_set_is_what_the_design_lists():
    names = sorted(f["name"] for f in FIXTURES)
    assert sum(n.startswith("nlp_srbd13_n30") for n in names) >= 4 and sum(n.startswith("nlp_srbd37_n20") for n in names) >= 2
    assert any(n.startswith("nlp_lip30") for n in names) and any(n.startswith("nlp_srbd61") for n in names)
    # at least one commanded-velocity instance per SRBD model (rdot_ref at the last node != 0)
    for m in ("srbd13", "srbd37"):
        assert any(f["model"] == m and np.any(f["params"][-1, 0:3] != 0.0) for f in FIXTURES)


@pytest.mark.parametrize("fx", FIXTURES, ids=[f["name"] for f in FIXTURES])
def test_c_oracle_lands_on_the_nlp_optimum(fx):
    cst = omodels.RobotConsts(**fx["consts"])
    x, u, st = cport.solve_batch(cst, oddp.DdpOptions(**nf.OPTS), fx["x0"][None], fx["params"][None], fx["xs0"][None], fx["us0"][None],
                                 model=fx["model"])
    assert int(st[0, 2]) == 1
    ex, eu, rc = nf.check(fx, x[0], u[0], st[0, 0])
    print(f"{fx['name']}: C oracle {int(st[0, 1])} iterations, linf x {ex:.1e} u {eu:.1e}, cost rel {rc:.1e}")


@pytest.mark.parametrize("fx", [f for f in FIXTURES if f["model"] != "srbd37" or f["seed"] == 0], ids=lambda f: f["name"])
def test_numpy_oracle_lands_on_the_nlp_optimum(fx):
    cst = omodels.RobotConsts(**fx["consts"])
    m = omodels.make_model(fx["model"], cst)
    r = oddp.solve(m, fx["x0"], fx["params"], fx["xs0"], fx["us0"], oddp.DdpOptions(**nf.OPTS))
    assert r.converged
    nf.check(fx, r.xs, r.us, r.cost)
