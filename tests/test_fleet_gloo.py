"""bench.py's N > 1 step on CPU: the SAME `fleet.FleetQueue` (submit -> flush = one launch + record packing + asynchronous,
double-buffered all-gather) at world size 2 with gloo, over three launches of distinct instances.  The HIP engine is GPU-only, so each rank's engine is a stand-in with the engine's three methods that
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
        self.sf = torch.from_numpy(self.stats.view(np.float64).reshape(B, _lib.STATS_F64_WORDS))
        self.si = torch.from_numpy(self.stats.view(np.int32).reshape(B, _lib.STATS_I32_WORDS))

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


def _seeds(rank, steps, step, B):
    return (rank * steps + step) * B + np.arange(B)                    # bench.py: every step of every rank solves its own instances


def _worker(rank, world, port, B, depth, steps, q, gather="full"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P0 = torch.zeros(depth * B, N + 1, NP, dtype=torch.float64)
    fleet = FleetQueue(OracleEngine(depth * B), P0, B, depth, collective=True, gather=gather)
    out = []
    held = None                                                        # launch k's gathered view, read again after launch k + 1 started
    for s in range(steps):
        batch = workload.make_batch("srbd13", N, _seeds(rank, steps, s, B))
        t = {k: torch.from_numpy(batch[k]) for k in ("x0", "xs", "us", "params")}
        if fleet.full:
            with pytest.raises(RuntimeError):                          # a full handle refuses the next batch: no silent flush
                fleet.submit(t["x0"], t["xs"], t["us"], t["params"])
            fleet.flush()
            if held is not None:
                out.append(("after_next_launch", held.numpy().copy()))
            held = fleet.gathered
            out.append(("launch", held.numpy().copy()))
        fleet.submit(t["x0"], t["xs"], t["us"], t["params"])
    fleet.flush()
    if held is not None:
        out.append(("after_next_launch", held.numpy().copy()))
    fleet.wait()
    out.append(("launch", fleet.gathered.numpy().copy()))
    q.put((rank, fleet.launches, fleet.gather_bytes, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("gather", ["full", "first_knot"])
def test_fleet_step_world2_gathers_every_ranks_records(gather):
    world, B, depth, steps = 2, 4, 2, 5                                # 5 steps on a depth-2 handle: launches of 2, 2 and 1 batches
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, depth, steps, q, gather)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    W = sdist.record_words(N, NX, NU, gather)
    assert W == (N + 1) * NX + N * NU + 2 if gather == "full" else W == NU + NX + 2 == 21
    # unsharded reference of each launch: rank-major, each rank's block = its steps of that launch in order
    launches = [[0, 1], [2, 3], [4]]
    refs = []
    for st in launches:
        seeds = np.concatenate([_seeds(r, steps, s, B) for r in range(world) for s in st])
        batch = workload.make_batch("srbd13", N, seeds)
        xo, uo, so = cport.solve_batch(omodels.RobotConsts(), oddp.DdpOptions(**OPTS), batch["x0"], batch["params"], batch["xs"], batch["us"])
        refs.append((xo, uo, so))
    for rank, n_launch, gbytes, out in res:
        assert n_launch == 3
        assert gbytes == steps * B * W * 8                             # every record leaves the rank exactly once
        kinds = [k for k, _ in out]
        assert kinds == ["launch", "after_next_launch", "launch", "after_next_launch", "launch"]
        li = 0
        for kind, rec in out:
            if kind == "after_next_launch":                            # the previous launch's records, re-read after the next launch
                xo, uo, so = refs[li - 1]                              # and its collective were started: the other buffer pair
            else:
                xo, uo, so = refs[li]
                li += 1
            assert rec.shape == (xo.shape[0], W)
            if gather == "full":
                x, u, cost, iters = sdist.unpack_records(rec, N, NX, NU)
                np.testing.assert_array_equal(x, xo)
                np.testing.assert_array_equal(u, uo)
            else:                                                          # u_0 | x_1 | cost | iterations
                np.testing.assert_array_equal(rec[:, :NU], uo[:, 0])
                np.testing.assert_array_equal(rec[:, NU:NU + NX], xo[:, 1])
                cost, iters = rec[:, NU + NX], rec[:, NU + NX + 1]
            np.testing.assert_array_equal(cost, so[:, 0])
            np.testing.assert_array_equal(iters, so[:, 1])
    for a, b in zip(res[0][3], res[1][3]):
        np.testing.assert_array_equal(a[1], b[1])                      # every rank ends with the same gathered tensors
