#!/usr/bin/env python3
"""List-scheduling model of the solve launch as a work queue: what a queue ORDER and what PACKING several
instances into one wavefront would buy on the driver-shaped run (20 steps x 1024 cold instances in one launch, 2048 slots).

Data: tools/data/bench_queue_iters.npz = DDP iteration count of every instance of seed blocks 0..19 (the timed region of
`bench.py --steps 20`), from the C oracle (identical to the GPU's on > 99 % of the instances), and each instance's rank in the
descending order of its initial cost (the key of sddp_options.queue_order = 2).  Unit of time: one instance-iteration of the
shipped one-wavefront kernel = sweep share S (derivatives, Riccati sweep, accept) + rollout share R (cycle stamps).

    python tools/queue_sim.py            # table printed in DESIGN.md

Model of packing G instances per wavefront: a wave-iteration = one sweep per resident instance (the tiles of
ONE instance fit the wave's LDS share, so sweeps stay sequential) + ONE rollout pass shared by all of them (G x 64/G step
lengths); sub-slots refill from the queue at wave-iteration boundaries; an instance needs `iters` iterations + a closing sweep.
"""
import heapq as hq
import os

import numpy as np

S, R = 0.72, 0.28


def simulate(iters, order, G=1, waves=2048):
    iters = np.asarray(iters)
    order = list(order)
    nb, pos = len(order), 0
    state, ev, end = [], [], 0.0
    for w in range(waves):
        act = []
        while len(act) < G and pos < nb:
            act.append(int(iters[order[pos]]) + 1); pos += 1        # sweeps left (iterations + the closing sweep)
        state.append(act)
        if act:
            hq.heappush(ev, (0.0, w))
    while ev:
        t, w = hq.heappop(ev)
        act = state[w]
        t += S * len(act) + (R if any(a > 1 for a in act) else 0.0)
        act = [a - 1 for a in act if a > 1]
        while len(act) < G and pos < nb:
            act.append(int(iters[order[pos]]) + 1); pos += 1
        state[w] = act
        if act:
            hq.heappush(ev, (t, w))
        else:
            end = max(end, t)
    return end


def main():
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "bench_queue_iters.npz"))
    it, rank = d["iters"].astype(int), d["j0_rank"]
    nb = len(it)
    ideal = float(np.sum(it * (S + R) + S)) / 2048
    print(f"{nb} instances, iterations mean {it.mean():.2f} max {it.max()}, 2048 slots; ideal (perfectly balanced) makespan {ideal:.1f}, "
          f"longest instance alone {it.max() * (S + R) + S:.1f}")
    orders = (("index order", np.arange(nb)), ("largest initial cost first (queue_order 2)", np.argsort(rank)),
              ("longest first, exact foreknowledge (replay)", np.argsort(-it, kind="stable")))
    base = simulate(it, orders[0][1])
    print(f"{'order':46s} " + " ".join(f"{'G=' + str(G):>14s}" for G in (1, 2, 3, 4)))
    for name, order in orders:
        row = []
        for G in (1, 2, 3, 4):
            m = simulate(it, order, G)
            row.append(f"{m:6.1f} ({base / m:4.2f}x)")
        print(f"{name:46s} " + " ".join(f"{r:>14s}" for r in row))
    print("(makespan in instance-iterations of the shipped kernel; in brackets: throughput relative to index order, G = 1)")


if __name__ == "__main__":
    main()
