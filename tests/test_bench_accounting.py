"""The byte accounting bench.py uses for the roofline: no GPU needed."""
import importlib.util
import os

import numpy as np

_spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
bench = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(bench)


def test_algorithmic_bytes_match_the_survey_figures():
    N, nx, nu, npar = 30, 13, 6, 19
    one = bench.algorithmic_bytes(N, nx, nu, npar, np.array([1]), np.array([1]), 1)
    assert one == 59024 + 14144                       # one iteration with one rollout + the I/O of one solve
    it_only = bench.algorithmic_bytes(N, nx, nu, npar, np.array([2]), np.array([0]), 1) - bench.algorithmic_bytes(
        N, nx, nu, npar, np.array([1]), np.array([0]), 1)
    ro_only = bench.algorithmic_bytes(N, nx, nu, npar, np.array([0]), np.array([1]), 1) - bench.algorithmic_bytes(
        N, nx, nu, npar, np.array([0]), np.array([0]), 1)
    assert it_only == 8 * (1172 + 2520 + 583) and ro_only == 8 * 3103      # read knots, write gains, write accepted; rollout reads
    # a batch is the sum of its instances
    iters, ro = np.array([3, 5, 7]), np.array([4, 5, 9])
    tot = bench.algorithmic_bytes(N, nx, nu, npar, iters, ro, 3)
    assert tot == sum(bench.algorithmic_bytes(N, nx, nu, npar, np.array([i]), np.array([r]), 1) for i, r in zip(iters, ro))


def test_bench_defaults_and_contract_fields():
    src = open(bench.__file__).read()
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"', '"higher_is_better"', '"scaling"',
                '"vs_baseline"', '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"', '"bound"', '"achieved"', '"peak"',
                '"frac"', '"traffic"', '"cores"', '"kind"', '"sample"'):
        assert key in src, key
    assert bench.HBM_PEAK_GBS == 8000.0
    assert "GPU_MAX_HW_QUEUES" not in src                            # one stream, one launch at a time: no hardware-queue tuning
    assert "traffic_source" in src                                   # the PMC figure is labelled as read from profiles/, not measured live


def test_no_timed_instance_is_solved_before_the_timed_region():
    """every step solves instances of its own; the warm-up blocks are disjoint from every rank's timed blocks,
    ranks do not share blocks, and the shipped default order uses no history of the timed instances themselves."""
    for world, steps, warmup in ((1, 20, 5), (8, 20, 5), (2, 96, 16), (1, 1, 0), (4, 3, 7)):
        timed = [bench.seed_block(r, world, steps, warmup, "timed", i, run=k) for k in range(bench.RUNS) for r in range(world) for i in range(steps)]
        warm = [bench.seed_block(r, world, steps, warmup, "warmup", i) for r in range(world) for i in range(warmup)]
        assert len(set(timed)) == bench.RUNS * world * steps and len(set(warm)) == world * warmup     # the timed regions share nothing either
        assert not set(timed) & set(warm)
        assert sorted(timed) == list(range(bench.RUNS * world * steps))   # region k, rank r owns the contiguous blocks (k*world + r)*steps ..
        assert sorted(timed[:world * steps]) == list(range(world * steps))
    src = open(bench.__file__).read()
    # `value` = longest class history first: the classes are schedule features, the history is OTHER instances (the warm-up steps
    # and earlier regions); nothing of a timed instance's solution is known when it starts
    assert '"--queue-order", type=int, default=3' in src
    assert "replay_history_order_solves_per_s" in src                 # the foreknowledge figure is an extra, named as a replay


def test_drain_profile_from_slot_clocks():
    """drain_frac = share of a launch during which fewer than half of its slots still hold an instance (100 MHz slot clocks)"""
    t = np.zeros((4, 2), dtype=np.uint64)
    t[:, 0] = 1000
    t[:, 1] = [1000 + 100, 1000 + 200, 1000 + 300, 1000 + 1000]       # one straggler: the median slot is dry at 200 of 1000
    d = bench.drain_profile(t)
    assert abs(d["drain_frac"] - 0.8) < 1e-12 and abs(d["idle_slot_time_frac"] - (1 - 400 / 1000)) < 1e-12 and d["slots"] == 4
    t[:, 1] = 1000 + 500                                               # perfectly balanced launch
    d = bench.drain_profile(t)
    assert d["drain_frac"] == 0.0 and d["idle_slot_time_frac"] == 0.0


def test_queue_model_reproduces_the_design_table():
    """tools/queue_sim.py is the list-scheduling model DESIGN.md section 5 quotes (queue orders, instances per wavefront): its
    data file and its conclusions are pinned here so that the table stays reproducible."""
    spec = importlib.util.spec_from_file_location("queue_sim", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "queue_sim.py"))
    qs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(qs)
    d = np.load(os.path.join(os.path.dirname(qs.__file__), "data", "bench_queue_iters.npz"))
    it, rank = d["iters"].astype(int), d["j0_rank"]
    assert len(it) == 20480 and it.max() == 100 and abs(it.mean() - 16.30) < 0.01
    index, cost, exact = np.arange(len(it)), np.argsort(rank), np.argsort(-it, kind="stable")
    m = {(name, G): qs.simulate(it, order, G) for name, order in (("index", index), ("cost", cost), ("exact", exact)) for G in (1, 2, 4)}
    ideal = float(np.sum(it * (qs.S + qs.R) + qs.S)) / 2048
    assert 1.45 < m["index", 1] / ideal < 1.55           # index order: ~1.5 x the balanced makespan
    assert 1.33 < m["cost", 1] / ideal < 1.42            # largest initial cost first: ~1.38 x
    assert m["exact", 1] / ideal < 1.04                  # exact foreknowledge: ~1.02 x
    assert m["cost", 1] < m["index", 1]
    # packing G instances per wavefront: a few per cent at G = 2 under the history-free order, a loss at G = 4, and no gain at
    # all once the order is right -- the stragglers' latency, not lane utilisation, bounds this queue
    assert 0.90 < m["cost", 2] / m["cost", 1] < 1.0 and m["cost", 4] >= m["cost", 1] * 0.99
    assert m["index", 4] > m["index", 1] and m["exact", 4] > 1.5 * m["exact", 1]


def test_steps_per_launch():
    """one launch per region by default at any world size (two drains cost more than the gather they would hide: measured);
    --launches-per-region splits it (bench.steps_per_launch)."""
    assert bench.steps_per_launch(20, 1) == 20 and bench.steps_per_launch(20, 8) == 20
    assert bench.steps_per_launch(20, 8, launches_per_region=2) == 10 and bench.steps_per_launch(5, 2, launches_per_region=2) == 3
    assert bench.steps_per_launch(20, 8, launches_per_region=4) == 5 and bench.steps_per_launch(96, 1, queue_depth=64) == 64
    assert bench.steps_per_launch(1, 8, launches_per_region=2) == 1


def test_class_labels_use_nothing_but_the_schedule():
    """queue_order 3's labels (workload.srbd13_schedule_classes): functions of the parameter tensor alone -- the contact schedule
    and the command -- not of the initial state, the warm start or anything a solve produces."""
    from srbd_horizon_amd import workload
    a = workload.make_srbd13_batch(30, np.arange(64))
    b = workload.make_srbd13_batch(30, np.arange(64), x0_draw=1)              # same robots and schedules, another initial state
    la, n = workload.srbd13_schedule_classes(a["params"])
    lb, _ = workload.srbd13_schedule_classes(b["params"])
    np.testing.assert_array_equal(la, lb)
    assert not np.array_equal(a["x0"], b["x0"])
    assert la.dtype == np.int32 and la.min() >= 0 and la.max() < n and len(np.unique(la)) > 4
    # the default labels carry the SIGN of the commanded velocity; they refine the zero / non-zero labels of the first version
    lu, nu = workload.srbd13_schedule_classes(a["params"], signed=False)
    assert n == 4 * 32 * 9 and nu == 4 * 32 * 4 and len(np.unique(la)) > len(np.unique(lu))
    assert all(len(np.unique(lu[la == c])) == 1 for c in np.unique(la))
