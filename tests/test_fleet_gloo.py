"""bench.py's N > 1 step on CPU: the SAME `fleet.FleetQueue` (submit -> flush = one launch + pack_records + all-gather) at
world size 2 with gloo.  The HIP engine is GPU-only, so each rank's engine is a stand-in with the engine's three methods that
solves its shard with the plain-C oracle; what is tested is sharding by rank (bench.py's seeds), the flush bookkeeping, record
packing from the engine's stats views, the collective, and that the gathered records equal an unsharded solve."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cport, ddp as oddp, models as omodels
from srbd_horizon_amd import _lib, dist as sdist, workload
from srbd_horizon_amd.fleet import FleetQueue

OPTS = dict(max_iters=100, alpha_converge_threshold=1e-12, beta=1e-3)
N, NX, NU, NP = 30, 13, 6, 19


class OracleEngine:
    """CPU stand-in with the engine's queue surface (load_range_device / solve_range_device / fetch_device_views)."""

    def __init__(self, B):
        self.B = B
        self.x0 = torch.zeros(B, NX, dtype=torch.float64)
        self.x = torch.zeros(B, N + 1, NX, dtype=torch.float64)
        self.u = torch.zeros(B, N, NU, dtype=torch.float64)
        self.stats = np.zeros(B, dtype=_lib.STATS_DTYPE)
        self.sf = torch.from_numpy(self.stats.view(np.float64).reshape(B, 7))
        self.si = torch.from_numpy(self.stats.view(np.int32).reshape(B, 14))

    def load_range_device(self, first, count, x0=None, x=None, u=None):
        s = slice(first, first + count)
        if x0 is not None: self.x0[s] = x0
        if x is not None: self.x[s] = x
        if u is not None: self.u[s] = u

    def solve_range_device(self, params, first, count):
        s = slice(first, first + count)
        xo, uo, so = cport.solve_batch(omodels.RobotConsts(), oddp.DdpOptions(**OPTS), self.x0[s].numpy(), params[s].numpy(),
                                       self.x[s].numpy(), self.u[s].numpy(), threads=2)
        self.x[s] = torch.from_numpy(xo)
        self.u[s] = torch.from_numpy(uo)
        self.stats["cost"][s] = so[:, 0]
        self.stats["iters"][s] = so[:, 1].astype(np.int32)

    def fetch_device_views(self):
        return self.x, self.u, self.sf, self.si


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, depth, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seeds = rank * B + np.arange(B)                                    # bench.py: instances sharded contiguously across ranks
    batch = workload.make_batch("srbd13", N, seeds)
    t = {k: torch.from_numpy(batch[k]) for k in ("x0", "xs", "us", "params")}
    fleet = FleetQueue(OracleEngine(depth * B), t["params"].repeat(depth, 1, 1).contiguous(), B, depth, collective=True)
    for _ in range(steps):
        fleet.submit(t["x0"], t["xs"], t["us"])
    fleet.flush()
    last = steps - depth * ((steps - 1) // depth)                      # batches in the last launch
    q.put((rank, fleet.launches, last, fleet.gathered.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_fleet_step_world2_gathers_every_ranks_records():
    world, B, depth, steps = 2, 6, 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, depth, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    # unsharded reference: all world * B instances in one process
    batch = workload.make_batch("srbd13", N, np.arange(world * B))
    xo, uo, so = cport.solve_batch(omodels.RobotConsts(), oddp.DdpOptions(**OPTS), batch["x0"], batch["params"], batch["xs"], batch["us"])
    for rank, launches, last, rec in res:
        assert launches == 2 and last == 1                             # 3 steps on a depth-2 handle: a full launch, then one batch
        assert rec.shape == (world * last * B, sdist.record_words(N, NX, NU))
        x, u, cost, iters = sdist.unpack_records(rec, N, NX, NU)
        np.testing.assert_array_equal(x, xo)                           # rank-major = instance order: rank r's block is seeds r*B..
        np.testing.assert_array_equal(u, uo)
        np.testing.assert_array_equal(cost, so[:, 0])
        np.testing.assert_array_equal(iters, so[:, 1])
    np.testing.assert_array_equal(res[0][3], res[1][3])                # every rank ends with the same gathered tensor
