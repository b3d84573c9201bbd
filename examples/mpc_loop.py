#!/usr/bin/env python3
"""The reference's example loops without ROS: python/dsrbd_example.py:82-185 (model srbd37, the default) and
python/dlip_example.py:89-160 (model lip30), or the reduced metric model (srbd13), on the MI355X engine.

    python examples/mpc_loop.py [--model srbd37|srbd61|lip30|srbd13] [--ticks 200] [--motion walking|standing|jumping]
                                [--vx 1.0] [--vy 0.0] [--host-shift] [--barrier W]

Prints per-tick solve time (what the reference publishes on `solution_time`, dsrbd_example.py:134-136), iterations, and the
record it would hand to CartesIO (cartesio.py:58-79) for the last tick.  Needs a GPU: the engine has no CPU fallback.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srbd_horizon_amd.mpc import MpcLoop  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="srbd37", choices=["srbd37", "srbd61", "lip30", "srbd13"])
    ap.add_argument("--ns", type=int, default=None, help="knots (default 20 as the examples; 30 for srbd13)")
    ap.add_argument("--ticks", type=int, default=200)
    ap.add_argument("--motion", default="walking", choices=["walking", "standing", "jumping"])
    ap.add_argument("--vx", type=float, default=1.0, help="joystick axis in [-1, 1] (dsrbd_example.py:119-122)")
    ap.add_argument("--vy", type=float, default=0.0)
    ap.add_argument("--host-shift", action="store_true", help="shift parameters / warm start on the host (default: on the GPU)")
    ap.add_argument("--barrier", type=float, default=0.0, help="friction-cone barrier weight (0 = reference behaviour)")
    args = ap.parse_args()
    ns = args.ns or (30 if args.model == "srbd13" else 20)
    loop = MpcLoop(args.model, ns, warm_start="shift" if args.host_shift else "device") if args.barrier <= 0 else None
    if loop is None:
        # the barrier is a model constant: build the problem with it switched on
        from srbd_horizon_amd import prb as _prb
        _prb.DEFAULT_PARAMS.update(friction_barrier_weight=args.barrier, friction_barrier_sharpness=5.0)
        loop = MpcLoop(args.model, ns, warm_start="shift" if args.host_shift else "device")
    its, conv = [], []
    for _ in range(args.ticks):
        ok, sol = loop.tick(args.motion, (args.vx, args.vy))
        its.append(int(loop.solver.stats["iters"]))
        conv.append(ok)
    ms = np.array(loop.solve_ms[min(10, args.ticks // 2):])
    print(f"{args.model} ns={ns} {args.motion} ({args.vx:+.1f},{args.vy:+.1f}): {args.ticks} ticks, solve ms median {np.median(ms):.3f} "
          f"p99 {np.percentile(ms, 99):.3f}, iterations mean {np.mean(its):.2f} max {max(its)}, converged {np.mean(conv):.3f}")
    rec = loop.reference_record(sol)
    print("com", np.round(rec["com"], 4), "base_link", np.round(rec["base_link"], 4))
    for frame, p in rec["contacts"].items():
        print(frame, np.round(p, 4))


if __name__ == "__main__":
    main()
