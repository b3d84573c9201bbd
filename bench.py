#!/usr/bin/env python3
"""bench.py -- SRBD-DDP solves/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: B = 1024 independent SRBD MPC instances per
GPU (BASELINE configs[2]; 8 GPUs x 1024 = configs[3]), N = 30 knots, nx = 13, nu = 6, each solved from a cold
warm start (x = x0 at every node, u = static input) to convergence with the reference example's solver options
(dsrbd_example.py:55-58).  EVERY step solves instances of its own (seed block rank * steps + step; the warm-up steps use
further blocks), so nothing solved in the timed region has been seen before.  Inputs are resident in HBM before the timed
region; a step = that batch's initial state, warm start and parameters (D2D) entering the engine's work queue
(srbd_horizon_amd/fleet.py).  The queue is solved by ONE launch per `--queue-depth` steps (and at the end of the timed region):
the resident wavefronts of the device (2 per SIMD = 2048) pull instances until the queue is empty (+ the asynchronous RCCL
all-gather of the solution records when N > 1), on ONE stream.

Why a queue: a batch ends with its slowest instance (up to 100 DDP iterations; the mean is 16) and 1024 instances do not fill
2048 wavefront slots, so one launch per batch leaves most SIMD time idle.  A launch of a queue still ends with its slowest
instance, so the order matters.  `value` is measured with `queue_order = 3`: every instance carries a CLASS label computed from
its schedule before it is solved (which feet stand at node 0, nodes to the first contact switch, the commanded forward / lateral
velocity as none / + / -: srbd_horizon_amd.workload.srbd13_schedule_classes), the handle keeps the mean iteration count of every class
over the instances it has solved so far -- at the start of the first timed region: the warm-up steps, other instances -- and the
queue starts the classes with the longest history first, the initial cost (a pre-pass of the launch) breaking ties.  No instance
of a timed region has been solved before, nothing of its solution is known.  Reported beside it: `initial_cost_order_solves_per_s`
(queue_order 2: the pre-pass alone, no history at all: rounds 3-4's headline), plain index order, and
`replay_history_order_solves_per_s` = longest-previous-solve-first on a handle that HAS solved the same instances once before
(exact foreknowledge: the upper bound a fleet of recurring robots approaches; round 2's headline), and the strictly sequential
one-batch-per-launch figure as `one_batch_in_flight_*`.  Weak scaling.

Rank 0 prints ONE JSON line; `roofline` and `cpu_baseline` are defined in DESIGN.md ("Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
W = 8                     # bytes per fp64 word


def algorithmic_bytes(N, nx, nu, npar, iters, rollouts, B):
    """per DDP iteration [read knots + write gains + accepted trajectory write] plus, per rollout,
    [gains + trajectory reads]; per solve the I/O of params, x0, warm start in and solution out."""
    it = W * (N * (nx + nu + npar) + nx + npar + N * (nu * nx + nu) + N * (nx + nu) + nx)
    ro = W * (N * (nu * nx + nu) + N * (nx + nu) + nx)
    io = W * ((N + 1) * npar + nx + 2 * ((N + 1) * nx + N * nu))
    return float(np.sum(iters) * it + np.sum(rollouts) * ro + B * io)


def pmc_traffic(steps, depth):
    """(HBM bytes per launch of the dominant kernel, source) from the committed rocprofv3 PMC passes of the latest round
    (profiles/<round>/pmc_summary.json: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command, gfx950
    corrections applied there).  NOT a measurement of this run.  The passes are taken at a given launch size; the figure is
    scaled to this run's average launch (instances per launch) since traffic is per instance."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        per_batch = float(d["traffic_bytes_per_launch"]) / float(d.get("batches_per_launch", 1))
        launches = -(-steps // depth)
        return per_batch * steps / launches, os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def pmc_issue():
    """What binds the dominant kernel, from the committed SQ pass of the latest round (NOT a measurement of this run): share of a
    wave's cycles in which it issues / waits, share of a SIMD's cycles in which it issues (waves per SIMD x the wave's share), and
    algorithmic FMAs per VALU wave-instruction (0.85 Mflop per iteration = 6.6 k wave-FMAs)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        sq, per = d["SQ_mean_per_dispatch"], d["per_instance_iteration"]
        wave_issue = sq["SQ_ACTIVE_INST_ANY"] / sq["SQ_WAVE_CYCLES"]
        return {"wave_issue_frac": wave_issue, "wave_wait_frac": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
                "simd_issue_frac": min(1.0, 2 * wave_issue), "valu_per_instance_iteration": per["SQ_INSTS_VALU"],
                "lds_per_instance_iteration": per["SQ_INSTS_LDS"], "salu_per_instance_iteration": per["SQ_INSTS_SALU"],
                "useful_valu_frac": 0.85e6 / 2 / 64 / per["SQ_INSTS_VALU"],
                "bound": "instruction issue (fp64 VALU), not HBM", "source": os.path.relpath(files[-1], ROOT)}
    except Exception:
        return None


def cpu_quota():
    """CPU time this process's cgroup may use, in cores (None: unlimited or unknown): a box that shares its host gets fewer cores'
    worth of time than the cores it may be scheduled on."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def cpu_baseline(N, B, blocks, budget_s=10.0):
    """The oracle's plain-C restatement (oracle/c/sddp_oracle.c, kind "port") timed on the host cores of this box on a bounded
    sample of the same workload: the first seed blocks the GPU solved in its timed region, OpenMP over instances.  `value` is the
    figure on ALL the cores this process may run on (north_star: "host cores, core count stated"); `share_of_one_gpu` the same
    sample on 16 threads (what a box with one of the node's eight GPUs gets), `one_thread` on one."""
    from oracle import cport, ddp as oddp, models as omodels
    from srbd_horizon_amd import workload
    cst = omodels.RobotConsts()
    opts = oddp.DdpOptions(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    quota = cpu_quota()
    seeds = np.concatenate([b * B + np.arange(B) for b in blocks])
    batch = workload.make_srbd13_batch(N, seeds)
    n_inst = len(seeds)

    def run(threads, budget):
        t0 = time.perf_counter()
        n = iters = reps = 0
        while True:
            _, _, st = cport.solve_batch(cst, opts, batch["x0"], batch["params"], batch["xs"], batch["us"], threads=threads)
            n += n_inst
            iters += int(st[:, 1].sum())
            reps += 1
            if time.perf_counter() - t0 > budget:
                break
        dt = time.perf_counter() - t0
        return n / dt, reps, iters / n, dt

    n1 = 64
    t0 = time.perf_counter()
    cport.solve_batch(cst, opts, batch["x0"][:n1], batch["params"][:n1], batch["xs"][:n1], batch["us"][:n1], threads=1)
    r1 = n1 / (time.perf_counter() - t0)
    v_all, reps_all, it_all, dt_all = run(cores, budget_s)
    t16 = max(1, min(16, cores))
    v16, reps16, _, dt16 = run(t16, budget_s * 0.6)
    what = f"seed blocks {list(blocks)} of the GPU's timed region ({n_inst} instances, {it_all:.2f} DDP iterations per solve)"
    qnote = "no cgroup CPU quota visible" if quota is None else f"cgroup CPU quota {quota:.1f} cores"
    all_cores = {"value": v_all, "unit": "solves/s", "cores": cores, "kind": "port",
                 "sample": f"{what} solved {reps_all}x in {dt_all:.1f} s with {cores} OpenMP threads = every core this process may be "
                           f"scheduled on (os.sched_getaffinity; the host reports {os.cpu_count()}; {qnote}); gcc -O3 -march=native"}
    share = {"value": v16, "unit": "solves/s", "cores": t16, "kind": "port",
             "sample": f"the same instances solved {reps16}x in {dt16:.1f} s with {t16} OpenMP threads (a box with one of the node's GPUs "
                       "gets a share of the host: more threads than that share only add contention)"}
    best = all_cores if v_all >= v16 else share
    # `value` / `cores`: the better of the two thread counts -- the best CPU figure this box can produce; both are reported
    return dict(best, all_affinity_cores=all_cores, share_of_one_gpu=share,
                one_thread={"value": r1, "unit": "solves/s", "cores": 1, "kind": "port", "sample": f"the first {n1} of these instances"})


def cpu_tick_baseline(model, trace, gpu_iters):
    """The recorded solver inputs of a receding-horizon run (mpc.MpcLoop.trace: x0, params, warm start of every tick) solved one
    by one by the plain-C oracle on ONE host thread: the CPU figure beside ms/MPC-tick (the reference's metric, the tic/toc
    around solver.solve() at dsrbd_example.py:134-136)."""
    from oracle import cport, ddp as oddp, models as omodels
    cst = omodels.RobotConsts()
    opts = oddp.DdpOptions(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)
    ms, its = [], []
    for t in trace:
        t1 = time.perf_counter()
        _, _, st = cport.solve_batch(cst, opts, t["x0"][None], t["params"][None], t["xs"][None], t["us"][None], threads=1, model=model)
        ms.append(1e3 * (time.perf_counter() - t1))
        its.append(int(st[0, 1]))
    return {"solve_median": float(np.median(ms)), "solve_p99": float(np.percentile(ms, 99)), "mean_iters": float(np.mean(its)),
            "same_iters_as_gpu_frac": float(np.mean(np.asarray(its) == np.asarray(gpu_iters))), "cores": 1, "kind": "port",
            "sample": f"{len(trace)} recorded ticks, each solved from the same x0 / parameters / warm start as the GPU tick"}


RUNS = 3     # timed regions per bench run, each over seed blocks of its own: `value` is their median


def seed_block(rank, world, steps, warmup, phase, i, run=0):
    """Block of instance seeds (block b = seeds b*B .. b*B+B-1) a step solves.  Timed step i of rank r in timed region `run`:
    block (run*world + r)*steps + i (the ranks shard the timed instances contiguously, every region has blocks of its own);
    warm-up step i of rank r: block RUNS*world*steps + r*warmup + i.  No instance of a timed region is ever solved before that
    region starts (tests/test_bench_accounting.py)."""
    if phase == "timed":
        assert 0 <= i < steps and 0 <= run < RUNS
        return (run * world + rank) * steps + i
    assert phase == "warmup" and 0 <= i < warmup
    return RUNS * world * steps + rank * warmup + i


def steps_per_launch(steps, world, launches_per_region=0, queue_depth=64):
    """Batches one launch covers.  Default: the whole region in ONE launch, for any number of ranks.  Splitting a region so that
    the all-gather of launch k hides behind launch k + 1 (--launches-per-region 2) was measured on one GPU with a one-rank RCCL
    communicator: 310 k solves/s against 420 k -- every launch drains on its own, and two half-length queues idle far more slot
    time (26 %) than the gather they hide could cost (47.9 MB per rank and launch: 1-2 ms of a 48 ms launch).  The option
    stays for nodes where the gather turns out slower than that."""
    lpr = launches_per_region if launches_per_region > 0 else 1
    return max(1, min(queue_depth, -(-steps // lpr)))


def drain_profile(slot_t):
    """From the per-slot clocks of ONE launch ([grid, 2]: first start, queue found empty): the share of the launch during which
    fewer than half of its slots still held an instance, and the share of slot-time spent idle behind the slowest slot."""
    t0 = float(slot_t[:, 0].min())
    ends = np.sort(slot_t[:, 1].astype(np.float64)) - t0
    total = ends[-1]
    if total <= 0:
        return None
    half = ends[(len(ends) - 1) // 2]                      # the moment the median slot ran dry: from here on < 50 % are busy
    return {"drain_frac": float(1.0 - half / total), "idle_slot_time_frac": float(1.0 - ends.mean() / total),
            "launch_ms_by_slot_clock": float(total / 1e5), "slots": int(len(ends))}


ORDER_NAMES = {0: "index", 1: "longest previous solve first (history of this handle)", 2: "largest initial cost first (pre-pass of the launch, no history)",
               3: "longest class history first: mean iterations of the instance's schedule class (stance at node 0, nodes to the first contact "
                  "switch, commanded forward / lateral velocity: none, + or -) over the OTHER instances this handle has solved -- here the warm-up steps; initial cost breaks ties"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--batch", type=int, default=1024, help="MPC instances per GPU and step")
    ap.add_argument("--horizon", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline and the single-instance extras")
    ap.add_argument("--queue-depth", type=int, default=64, help="steps (batches) one engine handle holds = most steps per launch")
    ap.add_argument("--class-signs", type=int, default=1, help="queue order 3: class labels with the SIGN of the commanded velocity (0: zero / non-zero only)")
    ap.add_argument("--waves-per-simd", type=int, default=2, help="kernel build: 1 = one wavefront per SIMD, 2 = two")
    ap.add_argument("--queue-order", type=int, default=3, help="sddp_options.queue_order: 3 longest class history first (classes = schedule "
                                                              "features known before the solve; history = the warm-up steps), 2 largest initial "
                                                              "cost first (no history), 0 index order, 1 longest previous solve first (needs history)")
    ap.add_argument("--launches-per-region", type=int, default=0,
                    help="split a timed region's steps over this many launches (0 = default = 1; 2: the all-gather of launch k is on "
                         "the wire while launch k + 1 solves, at the price of two drains)")
    ap.add_argument("--gather", default="full", choices=("full", "first_knot"),
                    help="what the N > 1 all-gather carries per instance: the whole plan (SURVEY 8(e), 4 680 B) or u_0 | x_1 | cost | iterations (168 B)")
    ap.add_argument("--no-extras", action="store_true", help="skip the index-order, replay and one-batch-in-flight measurements")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from srbd_horizon_amd import _lib, workload
    from srbd_horizon_amd.engine import DdpEngine
    from srbd_horizon_amd.fleet import FleetQueue

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the DDP engine has no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("SDDP_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of the N > 1 path on a 1-GPU box
    if world > 1 and backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    # SDDP_BENCH_FORCE_COLLECTIVE=1: rehearsal of the N > 1 step (record packing + all-gather inside the flush) with a one-rank
    # RCCL communicator on a 1-GPU box; the printed line then carries "collective_rehearsal": true
    collective = world > 1 or os.environ.get("SDDP_BENCH_FORCE_COLLECTIVE") == "1"
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    N, B = args.horizon, args.batch
    steps, warmup = max(args.steps, 1), max(args.warmup, 0)
    Q = steps_per_launch(steps, world, args.launches_per_region, args.queue_depth)
    nx, nu, npar = 13, 6, 19
    opts = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)      # dsrbd_example.py:55-58
    wps = args.waves_per_simd

    # every step solves its OWN instances: timed step i of this rank = seed block rank*steps + i, warm-up steps = other blocks
    def load_blocks(blocks, x0_draw=0):
        seeds = np.concatenate([b * B + np.arange(B) for b in blocks])
        h = workload.make_srbd13_batch(N, seeds, x0_draw=x0_draw)
        d = {k: torch.from_numpy(h[k]).to(dev).reshape((len(blocks), B) + h[k].shape[1:]) for k in ("x0", "xs", "us", "params")}
        lab, ncls = workload.srbd13_schedule_classes(h["params"], signed=bool(args.class_signs))   # what kind of problem each instance is (queue order 3)
        d["classes"], d["n_classes"] = torch.from_numpy(lab).to(dev).reshape(len(blocks), B), ncls
        return d

    run_blocks = [[seed_block(rank, world, steps, warmup, "timed", i, run=r) for i in range(steps)] for r in range(RUNS)]
    timed_blocks = run_blocks[0]
    warm_blocks = [seed_block(rank, world, steps, warmup, "warmup", i) for i in range(warmup)]
    d_runs = [load_blocks(bl) for bl in run_blocks]
    d_t = d_runs[0]
    d_w = load_blocks(warm_blocks) if warmup else None

    def make_queue(order, depth=None, **over):
        e = DdpEngine("srbd13", N, (depth or Q) * B, opts=dict(opts, waves_per_simd=wps, queue_order=order, **over))
        e.use_torch_stream(torch.cuda.current_stream())
        e.enable_timing(True)
        P_all = torch.zeros(((depth or Q) * B, N + 1, npar), dtype=torch.float64, device=dev)      # the handle's parameter tensor, resident
        return e, FleetQueue(e, P_all, B, depth or Q, collective=collective, gather=args.gather)

    eng, fleet = make_queue(args.queue_order)
    acc = torch.zeros(2, dtype=torch.int64, device=dev)        # DDP iterations, rollouts over all launches (device-side sums)

    def barrier():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    def launch(fl, count):
        n = fl.flush()                                         # ONE launch over the pending batches (+ async all-gather)
        if count and n:
            acc[0] += fl.si[:n, _lib.STATS_I32_ITERS].sum()                      # sddp_stats.iters / .rollouts of the launch, summed on the stream
            acc[1] += fl.si[:n, _lib.STATS_I32_ROLLOUTS].sum()

    def run_steps(fl, d, n_steps, count=False):
        for i in range(n_steps):
            if fl.full:
                launch(fl, count)
            fl.submit(d["x0"][i], d["xs"][i], d["us"][i], d["params"][i], d["classes"][i], d["n_classes"])    # one step: one batch of new instances enters the queue
        launch(fl, count)
        fl.wait()                                              # every collective of the region has completed

    def timed(fl, d, n_steps, count=False):
        barrier()
        t0 = time.perf_counter()
        run_steps(fl, d, n_steps, count)
        barrier()
        return time.perf_counter() - t0

    # warm-up: W steps on instances the timed region never sees (code paths incl. the device-side stats sums, caches, clocks)
    if warmup:
        run_steps(fleet, d_w, warmup, count=True)
    barrier()
    acc.zero_()
    eng.synchronize()
    eng.kernel_time_stats(reset=True)
    acc += fleet.si[:1, _lib.STATS_I32_ITERS].sum()                              # (first use of these device ops is never inside the timed region)
    acc.zero_()
    barrier()
    # RUNS timed regions, each EXACTLY `steps` steps over seed blocks of its own, each bracketed by barrier + synchronize and
    # reduced with MAX over the ranks; `value` is the median region (a launch ends with its slowest instance: one region is one
    # draw of the stragglers), all of them are reported
    regions = []
    for r in range(RUNS):
        acc.zero_()
        eng.synchronize()
        eng.kernel_time_stats(reset=True)
        barrier()
        l0, g0 = fleet.launches, fleet.gather_bytes
        el = timed(fleet, d_runs[r], steps, count=True)
        if collective:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        eng.synchronize()
        ks, kc = eng.kernel_time_stats(reset=True)
        it_r, ro_r = (int(v) for v in acc.tolist())
        x, u, st_r = eng.fetch()
        regions.append(dict(elapsed=el, ksum=ks, kcnt=kc, launches=fleet.launches - l0, gbytes=fleet.gather_bytes - g0, iters=it_r,
                            rollouts=ro_r, stats=st_r, drain=drain_profile(eng.slot_times()), blocks=[run_blocks[r][0], run_blocks[r][-1]]))
    order_el = sorted(range(RUNS), key=lambda r: regions[r]["elapsed"])
    med = regions[order_el[RUNS // 2]]
    elapsed, ksum, kcnt, launches = med["elapsed"], med["ksum"], med["kcnt"], med["launches"]
    tot_iters, tot_roll = med["iters"], med["rollouts"]
    timed_blocks = run_blocks[order_el[RUNS // 2]]
    d_t = d_runs[order_el[RUNS // 2]]
    slots, last_grid, last_queued = eng.queue_info()

    n_last = (steps - Q * ((steps - 1) // Q)) * B              # instances of the last launch: what the handle still holds
    st = med["stats"][:n_last]
    iters = st["iters"].astype(np.int64)
    kms = ksum / max(kcnt, 1)
    n_solves = steps * B
    abytes_total = algorithmic_bytes(N, nx, nu, npar, np.array([tot_iters]), np.array([tot_roll]), n_solves)
    abytes_launch = abytes_total / max(launches, 1)            # average launch of the timed region
    achieved = abytes_launch / (kms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(steps, Q) if (B == 1024 and N == 30 and world == 1) else (None, None)
    out = {
        "metric": "SRBD-DDP solves/sec (N=30, nx=13, nu=6)", "value": world * n_solves / elapsed, "unit": "solves/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
        "value_runs": [world * n_solves / r["elapsed"] for r in regions],
        "value_note": f"median of {RUNS} timed regions of exactly {steps} steps each, every region on seed blocks of its own "
                      "(value_runs: all of them, in execution order; ms_per_step, mean_iters, roofline and last_launch are the median region's)",
        "drain": med["drain"], "drain_runs": [r["drain"] for r in regions],
        "mean_iters_runs": [r["iters"] / n_solves for r in regions],
        "kernel_ms_runs": [r["ksum"] / max(r["kcnt"], 1) for r in regions],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"SRBD N={N} nx=13 nu=6, batch={B} independent MPC instances per GPU and step "
                               "(BASELINE configs[2]; x8 GPUs = configs[3]), cold start, whole line-search ladder "
                               "(alpha=1..1e-12, 40 candidates) rolled out per iteration; every step solves instances of its own "
                               "(distinct seeds), none of them seen before the timed region",
                   "batch_per_gpu": B, "horizon_N": N, "solver_opts": opts, "algorithm": "MS-DDP, Gauss-Newton Hessians + exact torque term",
                   "queue_depth_steps": Q, "launches_timed": launches, "resident_slots": slots, "grid_last_launch": last_grid,
                   "waves_per_simd": wps, "queue_order": ORDER_NAMES[args.queue_order], "streams": 1,
                   "seed_blocks": {"timed": [timed_blocks[0], timed_blocks[-1]], "warmup": ([warm_blocks[0], warm_blocks[-1]] if warmup else None),
                                   "block": f"seeds b*{B} .. b*{B}+{B - 1}"},
                   "launches_per_region": -(-steps // Q),
                   "collective": (f"asynchronous double-buffered all_gather of the '{args.gather}' record per launch (packed by one HIP kernel behind "
                                  "the solve); the gather of launch k is on the wire while launch k + 1 solves; all waited for inside the timed region")
                                 if collective else "none",
                   "gather_bytes_per_launch_per_rank": (med["gbytes"] // max(launches, 1)) if collective else 0,
                   "seed_blocks_of_the_runs": [r["blocks"] for r in regions]},
        "mean_iters": tot_iters / n_solves, "mean_rollouts": tot_roll / n_solves,
        "last_launch": {"instances": int(n_last), "mean_iters": float(np.mean(iters)), "max_iters": int(np.max(iters)),
                        "max_iters_hit_frac": float(np.mean(st["status"] == 1)), "converged_frac": float(np.mean(st["converged"] == 1)),
                        "line_search_stalled_frac": float(np.mean(st["status"] == 4))},
        "iterations_per_s": world * tot_iters / elapsed,
        # secondary: ~0.85 Mflop of fp64 per DDP iteration at (N, nx, nu) = (30, 13, 6) (dense backward
        # sweep 25.2 kflop/knot + model evaluation + one rollout), against the MI355X fp64 vector peak of 78.6 TFLOP/s
        # (DENSE-EQUIVALENT flops: the sweep that runs skips the zeros of [fx fu] and does fewer)
        "fp64_dense_equivalent_tflops": world * tot_iters / elapsed * 0.85e6 * (N / 30.0) / 1e12,
        "fp64_dense_equivalent_vector_peak_frac": tot_iters / elapsed * 0.85e6 * (N / 30.0) / 78.6e12,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src, "issue": pmc_issue(),
                     "kernel": eng.kernel_info()["kernel"], "resources": eng.kernel_info().get("resources"), "kernel_ms": kms,
                     "launches": int(kcnt), "algorithmic_bytes_per_launch": abytes_launch,
                     "note": "achieved = algorithmic bytes of the average timed launch (SURVEY 8(d) bytes per solve, from the iteration "
                             "and rollout counts of every instance of the timed region) / its HIP-event duration on the launch stream "
                             "(the events bracket the launch's queue-ordering pre-pass -- cost-key kernel + device sort -- and the solve kernel); "
                             "one launch at a time on one stream; traffic is not measured in this run: it is read from the committed "
                             "rocprofv3 PMC passes named in traffic_source"},
    }
    if rank == 0 and world == 1 and not args.no_extras:
        n_x = min(steps, Q)
        # the same instances in plain index order and (as a REPLAY: the handle has solved these very instances once before, so
        # the history is exact foreknowledge -- what a fleet of recurring robots approaches, not a cold-start figure) longest
        # previous solve first
        # history order as a recurring fleet would see it: the handle has solved the same ROBOTS before (same schedule, command and
        # footsteps) from another draw of the initial-state perturbation -- a similar problem, not the same one
        e_o, f_o = make_queue(1)
        d_sim = load_blocks(timed_blocks[:n_x], x0_draw=1)
        run_steps(f_o, d_sim, n_x)
        out["history_order_similar_problems_solves_per_s"] = B * n_x / timed(f_o, d_t, n_x)
        del f_o, e_o, d_sim
        for order, key in ((0, "index_order_solves_per_s"), (1, "replay_history_order_solves_per_s"), (2, "initial_cost_order_solves_per_s"),
                           (3, "class_history_order_solves_per_s")):
            if order == args.queue_order:
                continue
            e_o, f_o = make_queue(order)
            run_steps(f_o, d_t if order == 1 else (d_w if warmup else d_t), n_x if order == 1 else min(max(warmup, 1), n_x))
            out[key] = B * n_x / timed(f_o, d_t, n_x)
            del f_o, e_o
        # full second-order DDP (sddp_options.second_order = 2: its own kernel build), same instances, same queue order as `value`
        e_o, f_o = make_queue(args.queue_order, second_order=2)
        run_steps(f_o, d_w if warmup else d_t, min(max(warmup, 1), n_x))
        acc.zero_()
        el = timed(f_o, d_t, n_x, count=True)
        out["second_order2_solves_per_s"] = B * n_x / el
        out["second_order2_mean_iters"] = int(acc[0].item()) / (B * n_x)
        del f_o, e_o
        # strictly one batch per launch (the next step starts after the previous one's slowest instance has finished), the kernel
        # build with the full register file per instance: the latency of one batch
        e_lat = DdpEngine("srbd13", N, B, opts=dict(opts, waves_per_simd=1))
        f_lat = FleetQueue(e_lat, torch.zeros((B, N + 1, npar), dtype=torch.float64, device=dev), B, 1)
        n1 = min(steps, 6)
        el1 = 0.0
        for i in range(n1):
            barrier()
            t1 = time.perf_counter()
            f_lat.submit(d_t["x0"][i], d_t["xs"][i], d_t["us"][i], d_t["params"][i])
            f_lat.flush()
            barrier()
            el1 += time.perf_counter() - t1
        out["one_batch_in_flight_solves_per_s"] = B * n1 / el1
        out["one_batch_in_flight_ms_per_step"] = 1e3 * el1 / n1
        out["roofline"]["one_batch_in_flight_solves_per_s"] = out["one_batch_in_flight_solves_per_s"]
        del f_lat, e_lat
        # what the N > 1 default costs on ONE GPU: the same region as two launches of steps / 2 batches (the shape in which the
        # all-gather of launch k hides behind launch k + 1): each launch drains on its own
        if Q >= steps and steps >= 2:
            e_o, f_o = make_queue(args.queue_order, depth=-(-steps // 2))
            run_steps(f_o, d_w if warmup else d_t, min(max(warmup, 1), -(-steps // 2)))
            out["two_launches_per_region_solves_per_s"] = B * steps / timed(f_o, d_t, steps)
            del f_o, e_o
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # The tick loops below report a p99 / max per tick.  CPython's generation-2 garbage collection walks every tracked object
        # of the process (31 ms with torch imported) once per ~70 k container allocations: the "one-time 20-45 ms stall" of rounds
        # 2-3 (profiles/r04/experiments/host_stall_hiptrace.txt: no HIP call is in flight during it).  A real-time loop freezes
        # the set-up objects out of the collector's reach; so does this one.
        import gc
        gc.collect()
        gc.freeze()
        batch = workload.make_batch("srbd13", N, np.arange(B))
        out.update(single_instance_extras(N, opts, workload, DdpEngine))
        out["ms_per_fleet_tick"] = fleet_tick(N, B, opts, workload, DdpEngine)
        out["tick_ms_vs_batch"] = tick_curve(N, opts, workload, DdpEngine)
        out["ms_per_fleet_tick_srbd37"] = fleet_tick_reference_model(opts, workload, DdpEngine)
        # PCIe-inclusive batch rate (host-pointer C-ABI call: params in, x/u/stats out) -- reported, never `value`
        e_h = DdpEngine("srbd13", N, B, opts=dict(opts, waves_per_simd=1))
        e_h.set_initial_state(batch["x0"])
        t_host = []
        for _ in range(3):
            e_h.set_x_warmstart(batch["xs"]); e_h.set_u_warmstart(batch["us"])
            t1 = time.perf_counter()
            e_h.solve(batch["params"])
            t_host.append(time.perf_counter() - t1)
        out["pcie_inclusive_solves_per_s"] = B / min(t_host)
        out["roofline"]["pcie_inclusive_solves_per_s"] = out["pcie_inclusive_solves_per_s"]
        out["cpu_baseline"] = cpu_baseline(N, B, timed_blocks[:2])
    if collective and world == 1:
        out["collective_rehearsal"] = True
    if rank == 0:
        print(json.dumps(out))
    if collective:
        dist.destroy_process_group()


def sweep_flops_per_knot(nx, nu):
    """SURVEY.md section 8(d), dense count of one backward-sweep knot incl. the DDP tensor terms"""
    return (4 * nx ** 3 + 6 * nx ** 2 * nu + 2 * nu ** 2 * nx + (2 * nx ** 3 + 2 * nx ** 2 * nu + 2 * nx * nu ** 2) + nu ** 3 / 3
            + 2 * nu ** 2 * (nx + 1) + 4 * nx ** 2 + 4 * nx * nu)


def mw_batch(model, N, B, opts, workload, DdpEngine, reps=3):
    """One cold-started batch of a reference-size model through the 4-wavefront kernel (solve_kernel_mw), device-resident
    inputs, kernel time by HIP events on the launch stream: solves/s and the same roofline bookkeeping as the headline."""
    from srbd_horizon_amd import _lib
    nx, nu, npar = _lib.model_dims(model)
    b = workload.make_batch(model, N, np.arange(B))
    e = DdpEngine(model, N, B, opts=dict(opts, queue_order=2), consts=b["consts"])
    e.enable_timing(True)
    wall = []
    for _ in range(reps):
        e.set_initial_state(b["x0"]); e.set_x_warmstart(b["xs"]); e.set_u_warmstart(b["us"])
        e.set_params(b["params"]); e.synchronize()
        e.kernel_time_stats(reset=True)
        t1 = time.perf_counter()
        e.solve_resident()
        wall.append(time.perf_counter() - t1)
    ksum, kcnt = e.kernel_time_stats(reset=True)
    kms = ksum / max(kcnt, 1)
    st = e.stats
    iters, roll = st["iters"].astype(np.int64), st["rollouts"].astype(np.int64)
    ab = algorithmic_bytes(N, nx, nu, npar, iters, roll, B)
    flop_it = N * (sweep_flops_per_knot(nx, nu) + 2.0 * (nx * nu + nx * nx) + 6.0 * (nx + nu) ** 2)   # sweep + one rollout + model eval (approx.)
    slots, grid, queued = e.queue_info()
    return {"waves_per_simd": int(opts.get("waves_per_simd", 1)), "solves_per_s": B / min(wall), "kernel_solves_per_s": B / (kms * 1e-3), "batch": B, "horizon_N": N, "mean_iters": float(np.mean(iters)),
            "max_iters": int(iters.max()), "mean_rollouts": float(np.mean(roll)), "converged_frac": float(np.mean(st["converged"] == 1)),
            "slots": slots, "grid": grid, "kernel": e.kernel_info()["kernel"], "resources": e.kernel_info().get("resources"), "kernel_ms": kms,
            "algorithmic_bytes": ab, "achieved_gbs": ab / (kms * 1e-3) / 1e9, "hbm_frac": ab / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "fp64_dense_equivalent_tflops": float(iters.sum()) * flop_it / (kms * 1e-3) / 1e12,
            "fp64_dense_equivalent_vector_peak_frac": float(iters.sum()) * flop_it / (kms * 1e-3) / 78.6e12,
            "note": f"{model} N={N} cold start, one launch of {B} instances (queue on {grid} slots of 4 wavefronts, largest initial cost "
                    "first); solves_per_s includes the host-pointer result fetch, kernel_* is the HIP-event kernel time"}


def fleet_tick(N, B, opts, workload, DdpEngine, ticks=100, budget=6, cpu_ticks=6):
    """ms / MPC tick of a FLEET: B robots, each warm-started from its previous solution advanced by one knot (sddp_advance), one
    sddp_solve_resident per tick (results fetched to the host every tick), against the C port on the host threads for the same
    sequence of problems.  The per-robot figure (`ms_per_mpc_tick`, B = 1) is one wavefront of the chip; this is the chip.
    Run twice: to convergence (a tick lasts as long as its slowest robot: the tail), and with the real-time remedy of MPC -- at
    most `budget` DDP iterations per tick, an unfinished iterate (status 1) simply carried on as the next tick's warm start."""
    from oracle import cport, ddp as oddp, models as omodels
    b = workload.make_batch("srbd13", N, np.arange(B))
    cst, o = omodels.RobotConsts(), oddp.DdpOptions(**opts)
    threads = max(1, min(16, os.cpu_count() or 1))

    def run(max_iters, with_cpu):
        e = DdpEngine("srbd13", N, B, opts=dict(opts, waves_per_simd=1))
        e.set_initial_state(b["x0"]); e.set_x_warmstart(b["xs"]); e.set_u_warmstart(b["us"])
        e.set_params(b["params"])
        x, u = e.solve_resident()                              # cold solve: every robot's first tick (to convergence in both runs)
        if max_iters is not None:
            e.set_options(max_iters=max_iters)
        P = b["params"].copy()
        gms, cms, same, imean, imax, i99, unfinished, cost = [], [], [], [], [], [], [], []
        for t in range(ticks):
            p_last, x0 = P[:, -1].copy(), x[:, 1].copy()      # the plan's last column repeats; the robot is where the plan said
            xs_ws = np.concatenate([x[:, 1:], x[:, -1:]], axis=1); xs_ws[:, 0] = x0
            us_ws = np.concatenate([u[:, 1:], u[:, -1:]], axis=1)
            P = np.concatenate([P[:, 1:], p_last[:, None]], axis=1)
            t1 = time.perf_counter()
            e.advance(p_last, x0)
            x, u = e.solve_resident()
            gms.append(1e3 * (time.perf_counter() - t1))
            it = e.stats["iters"]
            imean.append(float(it.mean())); imax.append(int(it.max())); i99.append(float(np.percentile(it, 99)))
            unfinished.append(float(np.mean(e.stats["status"] == 1)))
            cost.append(float(e.stats["cost"].mean()))
            if with_cpu and t >= ticks - cpu_ticks:            # CPU: the last ticks only (bounded sample)
                t1 = time.perf_counter()
                _, _, st = cport.solve_batch(cst, o, x0, P, xs_ws, us_ws, threads=threads)
                cms.append(1e3 * (time.perf_counter() - t1))
                same.append(float(np.mean(st[:, 1].astype(int) == it)))
        return dict(gms=np.array(gms[4:]), cms=cms, same=same, imean=imean[4:], imax=np.array(imax[4:]), i99=i99[4:],
                    unfinished=unfinished[4:], cost=np.array(cost[4:]))

    def run_first(max_iters):
        """the same closed loop fetching only what a tick applies (u_0, x_1, cost, iterations): sddp_solve_resident_first"""
        e = DdpEngine("srbd13", N, B, opts=dict(opts, waves_per_simd=1))
        e.set_initial_state(b["x0"]); e.set_x_warmstart(b["xs"]); e.set_u_warmstart(b["us"])
        e.set_params(b["params"])
        u0, x1 = e.solve_resident_first()
        if max_iters is not None:
            e.set_options(max_iters=max_iters)
        p_last = b["params"][:, -1].copy()                     # the plan's last column repeats
        gms = []
        for t in range(ticks):
            t1 = time.perf_counter()
            e.advance(p_last, x1)
            u0, x1 = e.solve_resident_first()
            gms.append(1e3 * (time.perf_counter() - t1))
        return np.array(gms[4:])

    full, bud = run(None, True), run(budget, False)
    first_full, first_bud = run_first(None), run_first(budget)
    worst = np.argsort(-full["gms"])[:5]
    return {"batch": B, "ticks": ticks - 4,
            "gpu_ms_per_tick_median": float(np.median(full["gms"])), "gpu_ms_per_tick_p99": float(np.percentile(full["gms"], 99)),
            "gpu_ms_per_tick_max": float(full["gms"].max()),
            "mean_iters": float(np.mean(full["imean"])), "iters_p99": float(np.mean(full["i99"])), "iters_max_per_tick_median": float(np.median(full["imax"])),
            "iters_max": int(full["imax"].max()),
            "slowest_ticks": [{"tick": int(i) + 4, "ms": float(full["gms"][i]), "max_iters_of_a_robot": int(full["imax"][i])} for i in worst],
            "deadline_10ms_miss_frac": float(np.mean(full["gms"] > 10.0)),
            "budgeted": {"max_iters_per_tick": budget, "ms_per_tick_median": float(np.median(bud["gms"])),
                         "budgeted_p99_ms": float(np.percentile(bud["gms"], 99)), "ms_per_tick_max": float(bud["gms"].max()),
                         "deadline_10ms_miss_frac": float(np.mean(bud["gms"] > 10.0)),
                         "unfinished_robots_per_tick_mean": float(np.mean(bud["unfinished"])),
                         "mean_cost_ratio_to_converged": float(np.mean(bud["cost"] / full["cost"]))},
            "first_knot_fetch": {"ms_per_tick_median": float(np.median(first_full)), "ms_per_tick_p99": float(np.percentile(first_full, 99)),
                                 "budgeted_ms_per_tick_median": float(np.median(first_bud)), "budgeted_p99_ms": float(np.percentile(first_bud, 99)),
                                 "note": "sddp_solve_resident_first: only u_0, x_1, cost, iterations and status of every robot cross PCIe "
                                         "(156 KB instead of 4.8 MB per tick); the trajectories stay in HBM as the next warm start"},
            "cpu_ms_per_tick_median": float(np.median(full["cms"])), "cpu_threads": threads,
            "same_iters_as_gpu_frac": float(np.mean(full["same"])),
            "note": "srbd13 N=30, every robot warm-started from its previous solution advanced by one knot; GPU tick = sddp_advance + "
                    "sddp_solve_resident incl. the PCIe copies of p_last / x0 in and x / u / stats out; CPU = the C port, OpenMP over robots; "
                    "a tick to convergence ends with its slowest robot (slowest_ticks: its iteration count), the budgeted run caps "
                    "every robot at max_iters_per_tick and carries unfinished iterates over"}


def fleet_tick_reference_model(opts, workload, DdpEngine, model="srbd37", N=20, B=512, ticks=28, cpu_ticks=4):
    """ms / MPC tick of a fleet of the REFERENCE's own robots (srbd37, ns = 20: the problem dsrbd_example.py runs): B robots, one
    four-wavefront workgroup each, all resident at two workgroups per CU; warm-started ticks as in fleet_tick, only what a tick
    applies is fetched (sddp_solve_resident_first).  CPU: the C port on the host threads for the last ticks, same problems."""
    from oracle import cport, ddp as oddp, models as omodels
    b = workload.make_batch(model, N, np.arange(B) + 11000)
    cst, o = omodels.RobotConsts(), oddp.DdpOptions(**opts)
    threads = max(1, min(16, os.cpu_count() or 1))
    e = DdpEngine(model, N, B, opts=dict(opts, waves_per_simd=2))
    e.enable_timing(True)
    e.set_initial_state(b["x0"]); e.set_x_warmstart(b["xs"]); e.set_u_warmstart(b["us"])
    e.set_params(b["params"])
    x, u = e.solve_resident()
    P = b["params"].copy()
    gms, kms, cms, same, imean, imax = [], [], [], [], [], []
    for t in range(ticks):
        p_last, x0 = P[:, -1].copy(), x[:, 1].copy()
        xs_ws = np.concatenate([x[:, 1:], x[:, -1:]], axis=1); xs_ws[:, 0] = x0
        us_ws = np.concatenate([u[:, 1:], u[:, -1:]], axis=1)
        P = np.concatenate([P[:, 1:], p_last[:, None]], axis=1)
        t1 = time.perf_counter()
        e.advance(p_last, x0)
        e.solve_resident_first()
        gms.append(1e3 * (time.perf_counter() - t1))
        kms.append(e.last_kernel_ms())
        x, u, st = e.fetch()                                   # outside the timed tick: the CPU replay needs the whole warm start
        it = st["iters"]
        imean.append(float(it.mean())); imax.append(int(it.max()))
        if t >= ticks - cpu_ticks:
            t1 = time.perf_counter()
            _, _, cs = cport.solve_batch(cst, o, x0, P, xs_ws, us_ws, threads=threads, model=model)
            cms.append(1e3 * (time.perf_counter() - t1))
            same.append(float(np.mean(cs[:, 1].astype(int) == it)))
    g = np.array(gms[4:ticks - cpu_ticks])                     # the ticks before the host threads were handed to the CPU replay
    slots, grid, queued = e.queue_info()
    return {"model": model, "horizon_N": N, "batch": B, "ticks": len(g), "slots": slots,
            "gpu_ms_per_tick_median": float(np.median(g)), "gpu_ms_per_tick_p99": float(np.percentile(g, 99)), "gpu_ms_per_tick_max": float(g.max()),
            "mean_iters": float(np.mean(imean[4:])), "iters_max_per_tick_median": float(np.median(imax[4:])), "iters_max": int(np.max(imax[4:])),
            "gpu_kernel_ms_per_tick_median": float(np.median(kms[4:ticks - cpu_ticks])), "gpu_kernel_ms_per_tick_max": float(np.max(kms[4:ticks - cpu_ticks])),
            "cpu_ms_per_tick_median": float(np.median(cms[1:])), "cpu_ms_per_tick_all": [float(v) for v in cms], "cpu_threads": threads,
            "same_iters_as_gpu_frac": float(np.mean(same)),
            "note": f"{model} N={N}: {B} robots of the reference's own problem, warm-started ticks, sddp_advance + sddp_solve_resident_first "
                    "(first input, next state, cost, iterations, status of every robot over PCIe), two workgroups per CU; CPU = the C port, "
                    "OpenMP over robots, the same ticks (the first of its ticks warms the threads' work arrays up and is left out of the median).  "
                    "gpu_kernel_ms_*: the solve kernel alone by HIP events.  (Rounds 2-3 saw one tick per process with a host-side stall of "
                    "20-45 ms: CPython's generation-2 garbage collection, not the HIP runtime -- bench.py now freezes the set-up objects, "
                    "profiles/r04/experiments/host_stall_hiptrace.txt)"}


def tick_curve(N, opts, workload, DdpEngine, batches=(1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024), ticks=24, cpu_ticks=3):
    """ms per MPC tick against the number of robots per tick: GPU (sddp_advance + sddp_solve_resident, PCIe included) beside the C
    port on 1 and on all host threads, same problems (the GPU's tick sequence replayed).  One small robot is a single wavefront of
    a 256-CU chip and loses to a CPU core; the curve states from which fleet size the GPU path pays (crossover B*)."""
    from oracle import cport, ddp as oddp, models as omodels
    cst, o = omodels.RobotConsts(), oddp.DdpOptions(**opts)
    threads = max(1, min(16, os.cpu_count() or 1))
    rows = []
    for B in batches:
        b = workload.make_batch("srbd13", N, np.arange(B) + 7000)
        e = DdpEngine("srbd13", N, B, opts=dict(opts, waves_per_simd=1))
        e.set_initial_state(b["x0"]); e.set_x_warmstart(b["xs"]); e.set_u_warmstart(b["us"])
        e.set_params(b["params"])
        x, u = e.solve_resident()
        P = b["params"].copy()
        g, c1, cn, its = [], [], [], []
        for t in range(ticks):
            p_last, x0 = P[:, -1].copy(), x[:, 1].copy()
            xs_ws = np.concatenate([x[:, 1:], x[:, -1:]], axis=1); xs_ws[:, 0] = x0
            us_ws = np.concatenate([u[:, 1:], u[:, -1:]], axis=1)
            P = np.concatenate([P[:, 1:], p_last[:, None]], axis=1)
            t1 = time.perf_counter()
            e.advance(p_last, x0)
            x, u = e.solve_resident()
            g.append(1e3 * (time.perf_counter() - t1))
            its.append(float(e.stats["iters"].mean()))
            if t >= ticks - cpu_ticks:
                for th, acc_ in ((1, c1), (threads, cn)):
                    t1 = time.perf_counter()
                    cport.solve_batch(cst, o, x0, P, xs_ws, us_ws, threads=th)
                    acc_.append(1e3 * (time.perf_counter() - t1))
        rows.append({"batch": B, "gpu_ms": float(np.median(g[4:])), "cpu1_ms": float(np.median(c1)), f"cpu{threads}_ms": float(np.median(cn)),
                     "mean_iters": float(np.mean(its[4:]))})
    cross1 = next((r["batch"] for r in rows if r["gpu_ms"] < r["cpu1_ms"]), None)
    crossn = next((r["batch"] for r in rows if r["gpu_ms"] < r[f"cpu{threads}_ms"]), None)
    return {"rows": rows, "crossover_batch_vs_1_thread": cross1, f"crossover_batch_vs_{threads}_threads": crossn, "cpu_threads": threads,
            "note": "srbd13 N=30 warm-started ticks (previous solution advanced by one knot); GPU = sddp_advance + sddp_solve_resident with the "
                    "PCIe copies; CPU = the C port of the oracle (kind: port) on the same tick sequence; crossover = smallest measured batch "
                    "at which the GPU tick is shorter"}


def single_instance_extras(N, opts, workload, DdpEngine):
    """ms / MPC tick figures (BASELINE metric, second half): B = 1."""
    out = {}
    # configs[1]: one instance, host-pointer call (PCIe included), cold start
    e1 = DdpEngine("srbd13", N, 1, opts=opts)
    b1 = workload.make_batch("srbd13", N, [0])
    ticks = []
    for _ in range(30):
        e1.set_initial_state(b1["x0"]); e1.set_x_warmstart(b1["xs"]); e1.set_u_warmstart(b1["us"])
        t1 = time.perf_counter()
        e1.solve(b1["params"])
        ticks.append(1e3 * (time.perf_counter() - t1))
    out["ms_per_mpc_tick_b1"] = {"median": float(np.median(ticks[5:])), "p99": float(np.percentile(ticks[5:], 99)),
                                 "iters": int(e1.stats["iters"][0]), "note": "B=1, seed 0, cold start, host-pointer sddp_solve (PCIe-inclusive)"}
    # ms / MPC tick as SURVEY 8(d) defines it: receding-horizon loop (param shift + pack + solve + unpack + simulate),
    # B = 1, warm-started from the previous tick, walking with a forward command; 20 warm-up + 200 timed ticks
    from srbd_horizon_amd.mpc import MpcLoop
    def traced(model, ns, ticks):
        """the same (deterministic) loop once more, untimed, recording every tick's solver inputs for the CPU replay"""
        lp = MpcLoop(model, ns, warm_start="device")
        lp.trace = []
        for _ in range(ticks):
            lp.tick("walking", (1.0, 0.0))
        return lp.trace

    loop = MpcLoop("srbd13", N, warm_start="device")
    tick_ms, its = [], []
    for i in range(220):
        t1 = time.perf_counter()
        loop.tick("walking", (1.0, 0.0))
        tick_ms.append(1e3 * (time.perf_counter() - t1))
        its.append(int(loop.solver.stats["iters"]))
    out["ms_per_mpc_tick"] = {"median": float(np.median(tick_ms[20:])), "p99": float(np.percentile(tick_ms[20:], 99)),
                              "solve_median": float(np.median(loop.solve_ms[20:])), "mean_iters": float(np.mean(its[20:])),
                              "cpu": cpu_tick_baseline("srbd13", traced("srbd13", N, 220)[20:], its[20:]),
                              "note": "srbd13 receding-horizon loop (mpc.MpcLoop = dsrbd_example.py:82-185 without ROS), B=1, "
                                      "N=30, walking forward, warm start = previous solution, 200 ticks after 20 warm-up; "
                                      "tick = host scheduler + sddp_advance (device-side shift of parameters and warm start; last "
                                      "parameter column and state over PCIe) + sddp_solve_resident + unpack + one simulator step"}
    # the reference's own example loops (its real problem sizes, ns = 20, T = 1 s): dsrbd_example.py (srbd37) and
    # dlip_example.py (lip30, configs[0]); 10 warm-up + 100 timed ticks each
    out["ms_per_mpc_tick_reference_models"] = {}
    for mname, ns in (("srbd37", 20), ("lip30", 20), ("srbd37", 60), ("srbd61", 20)):
        lp = MpcLoop(mname, ns, warm_start="device")
        tms, its = [], []
        nt = 50 if mname == "srbd61" else (110 if ns == 20 else 60)
        for i in range(nt):
            t1 = time.perf_counter()
            lp.tick("walking", (1.0, 0.0))
            tms.append(1e3 * (time.perf_counter() - t1))
            its.append(int(lp.solver.stats["iters"]))
        out["ms_per_mpc_tick_reference_models"][mname + ("" if ns == 20 else f"_n{ns}")] = {
            "median": float(np.median(tms[10:])), "p99": float(np.percentile(tms[10:], 99)),
            "solve_median": float(np.median(lp.solve_ms[10:])), "mean_iters": float(np.mean(its[10:])),
            "cpu": cpu_tick_baseline(mname, traced(mname, ns, nt)[10:], its[10:])}
    # the 4-wavefront kernel as a batch: BASELINE configs[4] (srbd37, N = 60) and the reference's own problem (srbd37, ns = 20,
    # dsrbd_example.py:30-31), cold start with open defects, one launch.  Two workgroups per CU (waves_per_simd = 2: 512 slots,
    # the tiles of two instances fit a CU's LDS since round 3), four queue rounds per launch; the one-per-CU build beside it
    w2 = dict(opts, waves_per_simd=2)
    out["srbd37_n60_batch"] = mw_batch("srbd37", 60, 1024, w2, workload, DdpEngine)
    out["srbd37_n20_batch"] = mw_batch("srbd37", 20, 2048, w2, workload, DdpEngine)
    out["srbd37_n20_batch_one_workgroup_per_cu"] = mw_batch("srbd37", 20, 1024, opts, workload, DdpEngine)
    # BASELINE configs[0]'s model as a batch
    out["lip30_n20_batch"] = mw_batch("lip30", 20, 4096, w2, workload, DdpEngine)
    # the reference problem at the contact configuration its code defaults to (contact_model = 4: nx 61, nu 48; prb.py:39-41):
    # one workgroup per CU (its tiles take 145 KB of the CU's 160)
    out["srbd61_n20_batch"] = mw_batch("srbd61", 20, 1024, opts, workload, DdpEngine)
    out["srbd61_n60_batch"] = mw_batch("srbd61", 60, 512, opts, workload, DdpEngine, reps=2)
    return out


if __name__ == "__main__":
    main()
