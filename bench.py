#!/usr/bin/env python3
"""bench.py -- SRBD-DDP solves/sec on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: B = 1024 independent SRBD MPC instances per
GPU (BASELINE configs[2]; 8 GPUs x 1024 = configs[3]), N = 30 knots, nx = 13, nu = 6, each solved from a cold
warm start (x = x0 at every node, u = static input) to convergence with the reference example's solver options
(dsrbd_example.py:55-58).  Inputs are resident in HBM before the timed region; a step = reset of the batch's initial state and
warm start (D2D) + its instances entering the engine's work queue (srbd_horizon_amd/fleet.py).  The queue is solved by ONE
launch per `--queue-depth` steps (and at the end of the timed region): the resident wavefronts of the device (2 per SIMD =
2048) pull instances until the queue is empty (+ the RCCL all-gather of the solution records when N > 1), on ONE stream.

Why a queue: a batch ends with its slowest instance (93 DDP iterations; the mean is 16) and 1024 instances do not fill 2048
wavefront slots, so one launch per batch leaves most SIMD time idle.  The queue is ordered longest-previous-solve-first
(sddp_options.queue_order, include/sddp.h): in this bench every step re-solves the same synthetic batch, so the order hint
from the priming pass is exact -- the figure with plain index order is reported beside it (`index_order_solves_per_s`), and
the strictly sequential one-batch-per-launch figure as `one_batch_in_flight_*`.  Weak scaling.

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


def cpu_baseline(N, B, budget_s=12.0):
    """The oracle's plain-C restatement (oracle/c/sddp_oracle.c, kind "port") timed on the host cores of this box on a
    bounded sample of the same workload: the bench batch itself, OpenMP over instances."""
    from oracle import cport, ddp as oddp, models as omodels
    from srbd_horizon_amd import workload
    cst = omodels.RobotConsts()
    opts = oddp.DdpOptions(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)
    threads = max(1, min(16, os.cpu_count() or 1))          # a 1-GPU box's CPU share is 16 cores
    n1 = 64
    batch = workload.make_batch("srbd13", N, np.arange(B))
    t0 = time.perf_counter()
    cport.solve_batch(cst, opts, batch["x0"][:n1], batch["params"][:n1], batch["xs"][:n1], batch["us"][:n1], threads=1)
    r1 = n1 / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    n = iters = reps = 0
    while True:
        _, _, st = cport.solve_batch(cst, opts, batch["x0"], batch["params"], batch["xs"], batch["us"], threads=threads)
        n += B
        iters += int(st[:, 1].sum())
        reps += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "solves/s", "cores": threads, "kind": "port",
            "sample": f"the bench batch ({B} instances, seeds 0..{B - 1}) solved {reps}x = {n} solves, {iters} DDP iterations in {dt:.1f} s "
                      f"with {threads} OpenMP threads (gcc -O3 -march=native); 1 thread: {r1:.1f} solves/s on the first {n1} instances; "
                      f"host has {os.cpu_count()} cores"}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=192)
    ap.add_argument("--warmup", type=int, default=48)
    ap.add_argument("--batch", type=int, default=1024, help="MPC instances per GPU and step")
    ap.add_argument("--horizon", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline and the single-instance extras")
    ap.add_argument("--queue-depth", type=int, default=64, help="steps (batches) one engine handle holds = most steps per launch")
    ap.add_argument("--waves-per-simd", type=int, default=2, help="kernel build: 1 = one wavefront per SIMD, 2 = two")
    ap.add_argument("--queue-order", type=int, default=1, help="1: longest previous solve first (sddp_options.queue_order), 0: index order")
    ap.add_argument("--no-extras", action="store_true", help="skip the index-order and one-batch-in-flight measurements")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from srbd_horizon_amd import workload
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
    Q = max(1, min(args.queue_depth, max(args.steps, 1)))
    nx, nu, npar = 13, 6, 19
    seeds = rank * B + np.arange(B)                        # instances are sharded contiguously across ranks
    batch = workload.make_batch("srbd13", N, seeds)
    opts = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)      # dsrbd_example.py:55-58
    wps = args.waves_per_simd
    d_x0 = torch.from_numpy(batch["x0"]).to(dev)
    d_xs = torch.from_numpy(batch["xs"]).to(dev)
    d_us = torch.from_numpy(batch["us"]).to(dev)
    d_P = torch.from_numpy(batch["params"]).to(dev)
    d_P_all = d_P.repeat(Q, 1, 1).contiguous()             # the queue's parameter tensor, resident: every step's batch has the same plan

    def make_queue(order):
        e = DdpEngine("srbd13", N, Q * B, opts=dict(opts, waves_per_simd=wps, queue_order=order))
        e.use_torch_stream(torch.cuda.current_stream())
        e.enable_timing(True)
        return e, FleetQueue(e, d_P_all, B, Q, collective=collective)

    eng, fleet = make_queue(args.queue_order)

    def barrier():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(fl, n_steps):
        for _ in range(n_steps):
            fl.submit(d_x0, d_xs, d_us)                    # one step: one batch enters the queue (launch when the handle is full)
        fl.flush()

    def timed(fl, n_steps):
        barrier()
        t0 = time.perf_counter()
        run_steps(fl, n_steps)
        barrier()
        return time.perf_counter() - t0

    # warm-up: the W steps asked for, and at least one pass over every block of the handle the timed region will use, so that
    # each of its instances has been solved once (code paths, caches, and the queue-order history)
    priming = max(max(args.warmup, 0), min(Q, args.steps))
    run_steps(fleet, priming)
    barrier()
    eng.synchronize()
    eng.kernel_time_stats(reset=True)
    l0 = fleet.launches
    elapsed = timed(fleet, args.steps)
    if collective:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    eng.synchronize()
    ksum, kcnt = eng.kernel_time_stats(reset=True)
    launches = fleet.launches - l0
    slots, last_grid, last_queued = eng.queue_info()

    x, u, st = eng.fetch()
    st = st[:B]                                            # every block of the handle holds the same batch
    iters, rollouts = st["iters"].astype(np.int64), st["rollouts"].astype(np.int64)
    kms = ksum / max(kcnt, 1)
    abytes_batch = algorithmic_bytes(N, nx, nu, npar, iters, rollouts, B)
    abytes_launch = abytes_batch * args.steps / max(launches, 1)          # average launch of the timed region
    achieved = abytes_launch / (kms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(args.steps, Q) if (B == 1024 and N == 30 and world == 1) else (None, None)
    out = {
        "metric": "SRBD-DDP solves/sec (N=30, nx=13, nu=6)", "value": world * B * args.steps / elapsed, "unit": "solves/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"SRBD N={N} nx=13 nu=6, batch={B} independent MPC instances per GPU and step "
                               "(BASELINE configs[2]; x8 GPUs = configs[3]), cold start, whole line-search ladder "
                               "(alpha=1..1e-12, 40 candidates) rolled out per iteration",
                   "batch_per_gpu": B, "horizon_N": N, "solver_opts": opts, "algorithm": "MS-DDP, Gauss-Newton Hessians + exact torque term",
                   "queue_depth_steps": Q, "launches_timed": launches, "resident_slots": slots, "grid_last_launch": last_grid,
                   "waves_per_simd": wps, "queue_order": "longest previous solve first" if args.queue_order else "index",
                   "priming_steps": priming, "streams": 1,
                   "collective": "all_gather(solution records) per launch" if collective else "none"},
        "mean_iters": float(np.mean(iters)), "max_iters": int(np.max(iters)), "max_iters_hit_frac": float(np.mean(st["status"] == 1)),
        "converged_frac": float(np.mean(st["converged"] == 1)), "line_search_stalled_frac": float(np.mean(st["status"] == 4)),
        "mean_rollouts": float(np.mean(rollouts)),
        "iterations_per_s": world * float(np.sum(iters)) * args.steps / elapsed,
        # secondary: ~0.85 Mflop of fp64 per DDP iteration at (N, nx, nu) = (30, 13, 6) (dense backward
        # sweep 25.2 kflop/knot + model evaluation + one rollout), against the MI355X fp64 vector peak of 78.6 TFLOP/s
        "fp64_algorithmic_tflops": world * float(np.sum(iters)) * args.steps / elapsed * 0.85e6 * (N / 30.0) / 1e12,
        "fp64_vector_peak_frac": float(np.sum(iters)) * args.steps / elapsed * 0.85e6 * (N / 30.0) / 78.6e12,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": ("solve_kernel_w2" if wps >= 2 else "solve_kernel") + "<SrbdModel<2,false>>", "kernel_ms": kms,
                     "launches": int(kcnt), "algorithmic_bytes_per_launch": abytes_launch,
                     "note": "achieved = algorithmic bytes of the average timed launch (SURVEY 8(d) bytes per solve x the instances of "
                             "the launch) / its HIP-event duration on the launch stream; one launch at a time on one stream; traffic is "
                             "not measured in this run: it is read from the committed rocprofv3 PMC passes named in traffic_source"},
    }
    if rank == 0 and world == 1 and not args.no_extras:
        n_x = min(args.steps, Q)
        if args.queue_order:
            # the same queue in plain index order (no history hint), over one full handle
            e_ix, f_ix = make_queue(0)
            run_steps(f_ix, n_x)
            el = timed(f_ix, n_x)
            out["index_order_solves_per_s"] = B * n_x / el
            del f_ix, e_ix
        # strictly one batch per launch (the next step starts after the previous one's slowest instance has finished), the kernel
        # build with the full register file per instance: the latency of one batch
        e_lat = DdpEngine("srbd13", N, B, opts=dict(opts, waves_per_simd=1))
        e_lat.use_torch_stream(torch.cuda.current_stream())
        f_lat = FleetQueue(e_lat, d_P, B, 1)
        n1 = min(args.steps, 6)
        run_steps(f_lat, 1)
        el1 = timed(f_lat, n1)
        out["one_batch_in_flight_solves_per_s"] = B * n1 / el1
        out["one_batch_in_flight_ms_per_step"] = 1e3 * el1 / n1
        del f_lat, e_lat
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out.update(single_instance_extras(N, opts, workload, DdpEngine))
        out["ms_per_fleet_tick"] = fleet_tick(N, B, opts, workload, DdpEngine)
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
        out["cpu_baseline"] = cpu_baseline(N, B)
    if collective and world == 1:
        out["collective_rehearsal"] = True
    if rank == 0:
        print(json.dumps(out))
    if collective:
        dist.destroy_process_group()


def fleet_tick(N, B, opts, workload, DdpEngine, ticks=24):
    """ms / MPC tick of a FLEET: B robots, each warm-started from its previous solution advanced by one knot (sddp_advance), one
    sddp_solve_resident per tick (results fetched to the host every tick), against the C port on the host threads for the same
    sequence of problems.  The per-robot figure (`ms_per_mpc_tick`, B = 1) is one wavefront of the chip; this is the chip."""
    from oracle import cport, ddp as oddp, models as omodels
    b = workload.make_batch("srbd13", N, np.arange(B))
    e = DdpEngine("srbd13", N, B, opts=dict(opts, waves_per_simd=1))
    e.set_initial_state(b["x0"]); e.set_x_warmstart(b["xs"]); e.set_u_warmstart(b["us"])
    e.set_params(b["params"])
    x, u = e.solve_resident()                                  # cold solve: every robot's first tick
    P = b["params"].copy()
    cst, o = omodels.RobotConsts(), oddp.DdpOptions(**opts)
    threads = max(1, min(16, os.cpu_count() or 1))
    gms, cms, its, same = [], [], [], []
    for t in range(ticks):
        p_last, x0 = P[:, -1].copy(), x[:, 1].copy()          # the plan's last column repeats; the robot is where the plan said
        xs_ws = np.concatenate([x[:, 1:], x[:, -1:]], axis=1); xs_ws[:, 0] = x0
        us_ws = np.concatenate([u[:, 1:], u[:, -1:]], axis=1)
        P = np.concatenate([P[:, 1:], p_last[:, None]], axis=1)
        t1 = time.perf_counter()
        e.advance(p_last, x0)
        x, u = e.solve_resident()
        gms.append(1e3 * (time.perf_counter() - t1))
        its.append(float(e.stats["iters"].mean()))
        if t >= ticks - 6:                                     # CPU: the last ticks only (bounded sample)
            t1 = time.perf_counter()
            _, _, st = cport.solve_batch(cst, o, x0, P, xs_ws, us_ws, threads=threads)
            cms.append(1e3 * (time.perf_counter() - t1))
            same.append(float(np.mean(st[:, 1].astype(int) == e.stats["iters"])))
    return {"batch": B, "gpu_ms_per_tick_median": float(np.median(gms[4:])), "gpu_ms_per_tick_p99": float(np.percentile(gms[4:], 99)),
            "mean_iters": float(np.mean(its[4:])), "cpu_ms_per_tick_median": float(np.median(cms)), "cpu_threads": threads,
            "same_iters_as_gpu_frac": float(np.mean(same)),
            "note": "srbd13 N=30, every robot warm-started from its previous solution advanced by one knot; GPU tick = sddp_advance + "
                    "sddp_solve_resident incl. the PCIe copies of p_last / x0 in and x / u / stats out; CPU = the C port, OpenMP over robots"}


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
    for mname, ns in (("srbd37", 20), ("lip30", 20), ("srbd37", 60)):
        lp = MpcLoop(mname, ns, warm_start="device")
        tms, its = [], []
        nt = 110 if ns == 20 else 60
        for i in range(nt):
            t1 = time.perf_counter()
            lp.tick("walking", (1.0, 0.0))
            tms.append(1e3 * (time.perf_counter() - t1))
            its.append(int(lp.solver.stats["iters"]))
        out["ms_per_mpc_tick_reference_models"][mname + ("" if ns == 20 else f"_n{ns}")] = {
            "median": float(np.median(tms[10:])), "p99": float(np.percentile(tms[10:], 99)),
            "solve_median": float(np.median(lp.solve_ms[10:])), "mean_iters": float(np.mean(its[10:])),
            "cpu": cpu_tick_baseline(mname, traced(mname, ns, nt)[10:], its[10:])}
    # BASELINE configs[4] as a batch: srbd37, N = 60, multiple shooting from a cold start with open defects, one launch
    # (4 wavefronts per instance, one instance per CU resident: 256 slots, the rest queue)
    B5 = 512
    b5 = workload.make_batch("srbd37", 60, np.arange(B5))
    e5 = DdpEngine("srbd37", 60, B5, opts=opts)
    t5 = []
    for _ in range(3):
        e5.set_initial_state(b5["x0"]); e5.set_x_warmstart(b5["xs"]); e5.set_u_warmstart(b5["us"])
        e5.set_params(b5["params"]); e5.synchronize()
        t1 = time.perf_counter()
        e5.solve_resident()
        t5.append(time.perf_counter() - t1)
    out["srbd37_n60_batch"] = {"solves_per_s": B5 / min(t5), "batch": B5, "mean_iters": float(np.mean(e5.stats["iters"])),
                               "converged_frac": float(np.mean(e5.stats["converged"] == 1)), "slots": e5.queue_info()[0],
                               "note": "configs[4]: srbd37 N=60 cold start, host-pointer result fetch included"}
    return out


if __name__ == "__main__":
    main()
