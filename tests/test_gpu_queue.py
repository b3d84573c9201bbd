"""The solve launch as a work queue: more instances than resident workgroups, one launch, every
workgroup pulls instances until the queue is empty.  An instance's result must not depend on the slot that solved it, on what
that slot solved before, on the queue order or on the range of the batch a launch covers: everything here is BIT-exact
against one-workgroup-per-instance launches.  sddp_options.max_slots shrinks the slot count so that a few hundred
instances already queue; the last test runs BASELINE configs[3] (8 x 1024 instances, the 8 rank shards) through the real
queue (2048 slots) against the C oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import cport, ddp as oddp, models as omodels  # noqa: F401
from srbd_horizon_amd import workload
from srbd_horizon_amd.engine import DdpEngine
from srbd_horizon_amd.fleet import FleetQueue

pytestmark = pytest.mark.gpu

OPTS = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)      # dsrbd_example.py:55-58


def _engine(model, N, B, max_slots=None, **over):
    if max_slots is not None:
        over = dict(over, max_slots=max_slots)
    return DdpEngine(model, N, B, opts=dict(OPTS, **over))


def _solve(eng, batch):
    eng.set_initial_state(batch["x0"]); eng.set_x_warmstart(batch["xs"]); eng.set_u_warmstart(batch["us"])
    x, u = eng.solve(batch["params"])
    return x.copy(), u.copy(), eng.stats.copy()


@pytest.mark.parametrize("model,N,B,slots,wps", [("srbd13", 30, 300, 48, 1), ("srbd13", 30, 300, 37, 2), ("srbd37", 20, 7, 2, 1),
                                                 ("lip30", 20, 9, 4, 1)])
def test_queue_is_bit_identical_to_one_workgroup_per_instance(model, N, B, slots, wps):
    batch = workload.make_batch(model, N, np.arange(B) + 11)
    ref = _engine(model, N, B, waves_per_simd=wps)
    assert ref.queue_info()[0] == B                                   # every instance has its own slot: no queue
    x0, u0, s0 = _solve(ref, batch)
    assert ref.queue_info()[1:] == (B, 0)
    for order in (0, 1, 2):                                           # index, longest previous solve first, largest initial cost first
        q = _engine(model, N, B, max_slots=slots, waves_per_simd=wps, queue_order=order)
        assert q.queue_info()[0] == slots
        for rep in range(2):                                          # second solve: the order comes from the first one's history
            x, u, s = _solve(q, batch)
            assert q.queue_info()[1:] == (slots, B)
            np.testing.assert_array_equal(x, x0)
            np.testing.assert_array_equal(u, u0)
            for f in s0.dtype.names:
                np.testing.assert_array_equal(s[f], s0[f], err_msg=f)
    if model != "lip30":                                              # (the LQ problem takes one iteration everywhere)
        assert s0["iters"].max() > s0["iters"].min()                  # the instances do differ in length


def test_ranges_of_a_handle_solve_like_separate_batches():
    """The fleet queue of bench.py: blocks of a large handle loaded and solved range by range."""
    N, B, D = 30, 64, 3
    dev = torch.device("cuda", 0)
    batch = workload.make_batch("srbd13", N, np.arange(B) + 500)
    ref = _engine("srbd13", N, B)
    x0, u0, s0 = _solve(ref, batch)
    eng = _engine("srbd13", N, D * B, max_slots=40, waves_per_simd=2)
    eng.use_torch_stream(torch.cuda.current_stream())
    t = {k: torch.from_numpy(batch[k]).to(dev) for k in ("x0", "xs", "us", "params")}
    P_all = t["params"].repeat(D, 1, 1).contiguous()
    fleet = FleetQueue(eng, P_all, B, D)
    for steps in (1, 3, 5):                                           # partial handle, full handle, wrap-around (two launches)
        before = fleet.launches
        for _ in range(steps):
            if fleet.full:
                with pytest.raises(RuntimeError, match="full"):       # no silent flush: the caller launches
                    fleet.submit(t["x0"], t["xs"], t["us"])
                fleet.flush()
            fleet.submit(t["x0"], t["xs"], t["us"])
        fleet.flush()
        assert fleet.launches - before == -(-steps // D)
        x, u, s = eng.fetch()
        last = steps - D * ((steps - 1) // D)                         # blocks solved by the last launch
        for blk in range(last):
            sl = slice(blk * B, (blk + 1) * B)
            np.testing.assert_array_equal(x[sl], x0)
            np.testing.assert_array_equal(u[sl], u0)
            np.testing.assert_array_equal(s["iters"][sl], s0["iters"])
            np.testing.assert_array_equal(s["cost"][sl], s0["cost"])
    with pytest.raises(RuntimeError):
        eng.solve_range_device(P_all, D * B - 1, 2)                   # range past the batch
    with pytest.raises(RuntimeError):
        eng.backward(np.zeros((D * B, N + 1, 19)))                    # phase-level entry points need one slot per instance


def test_non_finite_options_are_rejected():
    for k, v in (("alpha_0", float("inf")), ("mu_max", float("nan")), ("beta", float("nan")), ("gap_tol", float("inf")),
                 ("mu_max", 1e-7), ("queue_order", 4), ("max_slots", -1), ("line_search_decrease_factor", 0.9999999), ("second_order", 3)):
        with pytest.raises(RuntimeError):
            DdpEngine("srbd13", 30, 1, opts=dict(OPTS, **{k: v}))


def test_configs3_all_eight_rank_shards_match_the_c_oracle(record_property):
    """BASELINE configs[3]: 8192 instances = the shards rank r = 0..7 of bench.py solve (seeds r * 1024 + arange(1024)), here
    through ONE handle on one GPU: a queue of 8192 instances on the device's resident slots, against the plain-C oracle.  The
    instances that take another iteration count are explained step by step (tests/shadow.py, tests/test_gpu_divergence.py)."""
    from tests import shadow
    from tests.test_gpu_divergence import assert_batch
    N, B, R = 30, 1024, 8
    batch = workload.make_batch("srbd13", N, np.arange(R * B))
    res = shadow.check_batch("srbd13", N, batch, OPTS, dict(waves_per_simd=2), omodels.RobotConsts(**batch["consts"]),
                             threads=min(16, os.cpu_count() or 1))
    slots, grid, queued = res["queue_info"]
    assert queued == R * B and grid == slots < R * B
    st, so = res["st"], res["so"]
    per_shard = [int((~res["same"][r * B:(r + 1) * B]).sum()) for r in range(R)]
    print(f"configs[3]: {len(res['explained'])} of {R * B} instances on another path, per rank shard {per_shard}; the two CPU builds "
          f"of the oracle split on {res['n_cpu_pair']}; slots {slots}; iterations mean {st['iters'].mean():.2f} max {st['iters'].max()}")
    assert_batch(res, "configs3_8192", record_property)
    assert int((so[:, 2] == 0).sum()) <= R * B // 500            # a handful of instances (of 8192) crawl past 100 iterations


def test_cold_queue_order_starts_the_costliest_warm_starts_first():
    """queue_order = 2: the pre-pass key is the initial total cost of each warm start, i.e. the cost the oracle reports for a
    solve cut off before its first iteration; the launch must hand out the instances in descending order of it.  Observed
    through a queue on ONE slot: the slot solves the instances one after another, so the order is the order in which the
    per-slot gains buffer is overwritten -- here simply checked through results (bit-exact, above) plus the key itself."""
    N, B = 30, 96
    batch = workload.make_batch("srbd13", N, np.arange(B) + 4000)
    cst, o0 = omodels.RobotConsts(**batch["consts"]), oddp.DdpOptions(**dict(OPTS, max_iters=0))
    _, _, so = cport.solve_batch(cst, o0, batch["x0"], batch["params"], batch["xs"], batch["us"])
    J0 = so[:, 0]
    eng = _engine("srbd13", N, B, max_slots=8, queue_order=2, max_iters=0)       # max_iters = 0: stats.cost = the initial cost
    x, u, st = _solve(eng, batch)
    np.testing.assert_allclose(st["cost"], J0, rtol=1e-12)
    assert eng.queue_info()[1:] == (8, B)
    # the engine's own view of the order it used
    order = eng.last_queue_order()
    assert sorted(order.tolist()) == list(range(B))
    assert np.all(np.diff(J0[order]) <= 1e-9 * np.abs(J0).max()), "queue not in descending initial-cost order"


def test_gains_pointer_is_refused_after_a_queued_launch():
    N, B = 30, 40
    batch = workload.make_batch("srbd13", N, np.arange(B))
    eng = _engine("srbd13", N, B)
    _solve(eng, batch)
    ptr, nbytes = eng.device_buffer(3)                                 # one slot per instance: row b is instance b's
    assert ptr and nbytes == B * N * 6 * 14 * 8
    q = _engine("srbd13", N, B, max_slots=8)
    _solve(q, batch)
    with pytest.raises(RuntimeError, match="per queue slot"):
        q.device_buffer(3)


@pytest.mark.parametrize("mode", ["full", "first_knot"])
def test_pack_records_kernel_writes_the_gather_record(mode):
    """sddp_pack_records_device (the one kernel that fills the all-gather's send buffer) against the host-side packing of
    srbd_horizon_amd.dist on the fetched results: bit-exact, any range of the batch."""
    from srbd_horizon_amd import dist as sdist
    N, B = 30, 24
    batch = workload.make_batch("srbd13", N, np.arange(B) + 77)
    eng = _engine("srbd13", N, B)
    x, u, st = _solve(eng, batch)
    W = eng.record_words(mode)
    assert W == sdist.record_words(N, 13, 6, mode)
    dev = torch.device("cuda", 0)
    for first, count in ((0, B), (5, 7), (B - 1, 1)):
        out = torch.full((count, W), float("nan"), dtype=torch.float64, device=dev)
        eng.pack_records_device(out, first, count, mode)
        eng.synchronize()
        ref = torch.empty((count, W), dtype=torch.float64)
        sl = slice(first, first + count)
        sdist.pack_records_into(ref, torch.from_numpy(x[sl]), torch.from_numpy(u[sl]), torch.from_numpy(st["cost"][sl].copy()),
                                torch.from_numpy(st["iters"][sl].copy()), mode)
        np.testing.assert_array_equal(out.cpu().numpy(), ref.numpy())
    with pytest.raises(RuntimeError):
        eng.pack_records_device(torch.empty((2, W), dtype=torch.float64, device=dev), B - 1, 2, mode)      # range past the batch


def test_class_history_orders_the_queue_and_changes_no_result():
    """queue_order = 3: instances labelled with a class; the handle learns the mean iteration count of each class from what it
    solves and starts the longest classes first.  Results stay bit-identical to any other order; the order follows the class means."""
    N, B = 30, 192
    batch = workload.make_batch("srbd13", N, np.arange(B) + 100)
    labels, ncls = workload.srbd13_schedule_classes(batch["params"])
    assert labels.min() >= 0 and labels.max() < ncls and len(np.unique(labels)) >= 6
    ref = _engine("srbd13", N, B)
    x0, u0, s0 = _solve(ref, batch)
    q = _engine("srbd13", N, B, max_slots=16, waves_per_simd=2, queue_order=3)
    q.set_instance_classes(labels, ncls)
    x, u, s = _solve(q, batch)                                        # no history yet: every class "unknown", ties by initial cost
    np.testing.assert_array_equal(x, x0); np.testing.assert_array_equal(u, u0)
    np.testing.assert_array_equal(s["iters"], s0["iters"])
    for c in np.unique(labels)[:4]:                                   # the statistics are what was solved
        m, n = q.class_history(int(c))
        assert n == int((labels == c).sum()) and abs(m - s0["iters"][labels == c].mean()) <= 1e-12
    x, u, s = _solve(q, batch)                                        # with history: by class mean
    np.testing.assert_array_equal(x, x0); np.testing.assert_array_equal(u, u0)
    order = q.last_queue_order()
    assert sorted(order.tolist()) == list(range(B))
    means = {int(c): s0["iters"][labels == c].mean() for c in np.unique(labels)}
    km = np.array([means[int(labels[i])] for i in order])
    assert np.all(np.diff(km) <= 1e-9)                                # descending class means along the queue
    with pytest.raises(RuntimeError):
        q.set_instance_classes(labels, ncls + 1)                      # n_classes is fixed by the first call
