"""Whole bench batches of the 4-wavefront kernel against the plain-C oracle: the `srbd37_n20_batch` (2048 cold instances of the
reference's own problem, two workgroups per CU, through the work queue), `srbd37_n60_batch` (BASELINE configs[4]'s shape, 1024
instances), `lip30_n20_batch` (4096) and `srbd61_n20_batch` (1024 instances of the code-default contact configuration) extras of
bench.py -- every instance: same iteration count, status and convergence flag, the same trajectory
at the north_star tolerance.  The counts go to the warnings summary (PARITY-COUNT) like those of the srbd13 batches."""
import numpy as np
import pytest

from oracle import cport
from oracle import ddp as oddp
from oracle import models as omodels
from srbd_horizon_amd import workload
from srbd_horizon_amd.engine import DdpEngine
from tests.conftest import report_parity

pytestmark = pytest.mark.gpu
OPTS = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)      # dsrbd_example.py:55-58


@pytest.mark.parametrize("model,N,B,wps", [("srbd37", 20, 2048, 2), ("srbd37", 60, 1024, 2), ("lip30", 20, 4096, 2), ("srbd61", 20, 1024, 1)])
def test_whole_batch_matches_the_c_oracle(model, N, B, wps, record_property):
    from tests import shadow
    from tests.test_gpu_divergence import assert_batch
    batch = workload.make_batch(model, N, np.arange(B))
    cst = omodels.make_model(model).cst if model == "srbd61" else omodels.RobotConsts(**batch["consts"])
    res = shadow.check_batch(model, N, batch, OPTS, dict(waves_per_simd=wps, queue_order=2), cst, threads=8)
    slots, grid, queued = res["queue_info"]
    assert queued == B and grid == slots < B                                # the launch was a queue on the resident workgroups
    assert res["st"]["converged"].all()
    assert_batch(res, f"{model}_batch_{B}", record_property)
