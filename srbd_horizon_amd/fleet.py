"""A fleet of independent MPC instances fed to ONE engine handle as a work queue ("work queue").

The engine holds ``depth`` batches of ``batch`` instances each.  ``submit`` loads one batch (initial state, warm start) into
the next free block of the handle; ``flush`` solves everything submitted since the last flush in ONE launch: the resident
workgroups of the device pull instances from a queue until it is empty, so the slowest instances overlap with all the others
instead of ending a launch alone (one batch = 1024 instances never fills the 2048 resident wavefronts of an MI355X, and a
launch lasts as long as its slowest instance).  With more than one rank (instances sharded contiguously across the GPUs of a
node) the flush closes with the one collective of the path: an all-gather of the solution records of
the solved blocks (``dist.pack_records`` / ``dist.all_gather_records``: RCCL over xGMI on GPUs, gloo in the CPU test).

This is the step function of ``bench.py``; ``tests/test_fleet_gloo.py`` drives the same class at world size 2 on CPU with an
engine stand-in.  The engine only needs ``load_range_device``, ``solve_range_device`` and ``fetch_device_views``.
"""
from __future__ import annotations

from . import dist as sdist


class FleetQueue:
    def __init__(self, engine, params_all, batch: int, depth: int, collective: bool = False):
        self.eng, self.P, self.batch, self.depth, self.collective = engine, params_all, int(batch), int(depth), bool(collective)
        if tuple(params_all.shape[:1]) != (self.batch * self.depth,):
            raise ValueError("params_all must hold depth * batch instances")
        self.pending = 0            # batches loaded and not yet solved
        self.launches = 0
        self.gathered = None        # records of every rank's last flush, rank-major [world * pending * batch, words]
        self.x, self.u, self.sf, self.si = engine.fetch_device_views()

    def submit(self, x0, xs, us):
        """One step: one batch of instances enters the queue (device tensors of `batch` instances)."""
        if self.pending == self.depth:
            self.flush()
        self.eng.load_range_device(self.pending * self.batch, self.batch, x0, xs, us)
        self.pending += 1

    def flush(self):
        """Solve the pending batches in one launch (asynchronous); with a process group, all-gather their solution records."""
        n = self.pending * self.batch
        if n == 0:
            return 0
        self.eng.solve_range_device(self.P, 0, n)
        self.launches += 1
        if self.collective:
            local = sdist.pack_records(self.x[:n], self.u[:n], self.sf[:n, 0], self.si[:n, 10])      # cost, iters
            self.gathered = sdist.all_gather_records(local)
        self.pending = 0
        return n
