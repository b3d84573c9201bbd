"""A fleet of independent MPC instances fed to ONE engine handle as a work queue ("work queue").

The engine holds ``depth`` batches of ``batch`` instances each.  ``submit`` loads one batch (initial state, warm start and,
optionally, its parameters) into the next free block of the handle; ``flush`` solves everything submitted since the last
flush in ONE launch: the resident workgroups of the device pull instances from a queue until it is empty, so the slowest
instances overlap with all the others instead of ending a launch alone (one batch = 1024 instances never fills the 2048
resident wavefronts of an MI355X, and a launch lasts as long as its slowest instance).

With more than one rank (instances sharded contiguously across the GPUs of a node) a flush closes with
the one collective of the path, an all-gather of the solution records of the solved blocks (RCCL over xGMI on GPUs, gloo in the
CPU test).  The collective is ASYNCHRONOUS and double buffered: the records of launch k are packed into pack buffer k mod 2
on the engine's stream (behind the solve), ``all_gather_into_tensor(async_op=True)`` fills gather buffer k mod 2, and the next
batches are loaded and launch k + 1 starts while it is on the wire.  A buffer pair is waited for before it is reused (at
flush k + 2) and by ``wait()`` -- which the caller runs before reading ``gathered`` and at the end of a timed region.  All
buffers are allocated once, in ``__init__`` (``depth * batch`` records per rank), never per flush.

``submit`` raises when the handle is full: the results of the pending batches live in the handle's buffers and the next load
would overwrite them, so the caller decides when to launch and when the results have been consumed (``flush`` / ``wait``).

This is the step function of ``bench.py``; ``tests/test_fleet_gloo.py`` drives the same class at world size 2 on CPU with an
engine stand-in.  The engine only needs ``load_range_device``, ``solve_range_device`` and ``fetch_device_views`` (and, on a
GPU, ``use_torch_stream``: packing and the collective are torch operations on torch's current stream, so the engine is bound to
that stream here).
"""
from __future__ import annotations

from . import _lib
from . import dist as sdist


class FleetQueue:
    def __init__(self, engine, params_all, batch: int, depth: int, collective: bool = False, gather: str = "full"):
        """gather: what the closing all-gather carries per instance -- "full": x | u | cost | iterations, the whole plan
        (4 680 B at (30, 13, 6)); "first_knot": u_0 | x_1 | cost | iterations (168 B), what a fleet in
        closed loop needs from the other ranks."""
        self.eng, self.P, self.batch, self.depth, self.collective = engine, params_all, int(batch), int(depth), bool(collective)
        if gather not in ("full", "first_knot"):
            raise ValueError("gather must be 'full' or 'first_knot'")
        self.gather = gather
        if tuple(params_all.shape[:1]) != (self.batch * self.depth,):
            raise ValueError("params_all must hold depth * batch instances")
        self.pending = 0            # batches loaded and not yet solved
        self.launches = 0
        self.gather_bytes = 0       # bytes this rank contributed to the collectives so far
        self.x, self.u, self.sf, self.si = engine.fetch_device_views()
        if self.x.is_cuda:
            import torch
            # solve, pack and collective must be ordered on ONE stream: torch's current one
            engine.use_torch_stream(torch.cuda.current_stream())
        self._work = [None, None]   # outstanding collective per buffer pair
        self._n = [0, 0]            # instances per rank in that collective
        self._k = 0                 # buffer pair of the next flush
        self._last = None           # buffer pair of the most recent flush
        self._pack = self._out = None
        if self.collective:
            import torch
            import torch.distributed as dist
            self.world = dist.get_world_size()
            words = sdist.record_words(self.x.shape[1] - 1, self.x.shape[2], self.u.shape[2], gather)
            cap = self.batch * self.depth
            mk = lambda rows: torch.empty((rows, words), dtype=torch.float64, device=self.x.device)
            self._pack = [mk(cap), mk(cap)]
            self._out = [mk(self.world * cap), mk(self.world * cap)]

    @property
    def full(self) -> bool:
        return self.pending == self.depth

    def submit(self, x0, xs, us, params=None, classes=None, n_classes=0):
        """One step: one batch of instances enters the queue (device tensors of `batch` instances; `params` [batch, N+1, np]
        replaces that block's parameters; `classes` [batch] int32 labels them for queue_order 3, sddp.h).  Raises when the handle
        is full: flush first."""
        if self.pending == self.depth:
            raise RuntimeError("FleetQueue is full: flush() (and consume the results) before submitting another batch")
        lo = self.pending * self.batch
        if params is not None:
            self.P[lo:lo + self.batch].copy_(params)
        if classes is not None:
            self.eng.set_instance_classes_range_device(lo, self.batch, classes, n_classes)
        self.eng.load_range_device(lo, self.batch, x0, xs, us)
        self.pending += 1

    def flush(self):
        """Solve the pending batches in one launch (asynchronous); with a process group, start the all-gather of their solution
        records (asynchronous too: `wait()` before reading `gathered`)."""
        n = self.pending * self.batch
        if n == 0:
            return 0
        self.eng.solve_range_device(self.P, 0, n)
        self.launches += 1
        if self.collective:
            import torch.distributed as dist
            k = self._k
            self._wait(k)                                                    # this pair's previous collective (two flushes ago)
            if hasattr(self.eng, "pack_records_device"):                   # the HIP engine: ONE kernel behind the solve
                local = self.eng.pack_records_device(self._pack[k][:n], 0, n, self.gather)
            else:                                                            # engine stand-in of the CPU tests
                local = sdist.pack_records_into(self._pack[k][:n], self.x[:n], self.u[:n], self.sf[:n, _lib.STATS_F64_COST],
                                                self.si[:n, _lib.STATS_I32_ITERS], self.gather)
            self._work[k] = dist.all_gather_into_tensor(self._out[k][:self.world * n], local, async_op=True)
            self._n[k] = n
            self.gather_bytes += local.numel() * local.element_size()
            self._last = k
            self._k ^= 1
        self.pending = 0
        return n

    def _wait(self, k):
        if self._work[k] is not None:
            self._work[k].wait()
            self._work[k] = None

    def wait(self):
        """Every outstanding collective has completed (on torch's current stream for RCCL) when this returns."""
        self._wait(0)
        self._wait(1)

    @property
    def gathered(self):
        """Records of every rank's last flush, rank-major [world * n, words] (a view of the gather buffer: valid until the flush
        after next).  Waits for that collective."""
        if self._last is None:
            return None
        self._wait(self._last)
        return self._out[self._last][:self.world * self._n[self._last]]
