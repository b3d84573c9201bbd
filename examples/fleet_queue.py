#!/usr/bin/env python3
"""One engine handle as a queue for a fleet of robots (include/sddp.h: sddp_load_range_device / sddp_solve_range_device).

    python examples/fleet_queue.py [--robots 8192] [--blocks 8] [--ticks 3]

`--robots` cold-started SRBD MPC instances (BASELINE configs[3] at the default) live in ONE handle; they are loaded block by
block and solved by one launch per tick: the wavefronts resident on the GPU pull instances from a queue, ordered by each
instance's iteration count in the previous tick (sddp_options.queue_order).  Prints solves/s per tick and the queue geometry.
Needs a GPU: the engine has no CPU fallback.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srbd_horizon_amd import workload  # noqa: E402
from srbd_horizon_amd.engine import DdpEngine  # noqa: E402
from srbd_horizon_amd.fleet import FleetQueue  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--robots", type=int, default=8192)
    ap.add_argument("--blocks", type=int, default=8)
    ap.add_argument("--ticks", type=int, default=3)
    ap.add_argument("--horizon", type=int, default=30)
    args = ap.parse_args()
    B, N = args.robots // args.blocks, args.horizon
    dev = torch.device("cuda", 0)
    opts = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3, waves_per_simd=2)     # dsrbd_example.py:55-58
    eng = DdpEngine("srbd13", N, args.blocks * B, opts=opts)
    eng.use_torch_stream(torch.cuda.current_stream())
    blocks = []
    for k in range(args.blocks):                                   # every block has its own robots (seeds)
        b = workload.make_batch("srbd13", N, k * B + np.arange(B))
        blocks.append({n: torch.from_numpy(b[n]).to(dev) for n in ("x0", "xs", "us", "params")})
    P_all = torch.cat([b["params"] for b in blocks]).contiguous()
    fleet = FleetQueue(eng, P_all, B, args.blocks)
    for t in range(args.ticks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in blocks:
            fleet.submit(b["x0"], b["xs"], b["us"])                 # cold start again: the worst case for the scheduler
        fleet.flush()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        x, u, st = eng.fetch()
        slots, grid, queued = eng.queue_info()
        print(f"tick {t}: {args.blocks * B} solves in {1e3 * dt:.1f} ms = {args.blocks * B / dt / 1e3:.0f} k solves/s | queue of {queued} on "
              f"{grid} slots | iterations mean {st['iters'].mean():.1f} max {st['iters'].max()} | converged {st['converged'].mean():.4f}"
              + ("   (first tick: index order, no history yet)" if t == 0 else ""))


if __name__ == "__main__":
    main()
