"""The N > 1 path on CPU: world_size 2 with the gloo backend (instance sharding + the one all-gather of solution records).
The solve itself is GPU-only, so the shards carry synthetic records; what is tested is partition + collective + ordering."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from srbd_horizon_amd import dist as sdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, N, nx, nu, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sdist.shard_range(total, rank, world)
    ids = np.arange(lo, hi)
    rng = [np.random.default_rng(int(i)) for i in ids]
    x = np.stack([r.standard_normal((N + 1, nx)) for r in rng])
    u = np.stack([r.standard_normal((N, nu)) for r in rng])
    cost, iters = ids.astype(float) * 2.0, ids % 7
    rec = torch.from_numpy(sdist.pack_records(x, u, cost, iters))
    sizes = [sdist.shard_range(total, r, world)[1] - sdist.shard_range(total, r, world)[0] for r in range(world)]
    full = sdist.all_gather_records(rec, sizes)
    xg, ug, cg, ig = sdist.unpack_records(full.numpy(), N, nx, nu)
    ok = full.shape == (total, sdist.record_words(N, nx, nu))
    for i in range(total):
        r = np.random.default_rng(i)
        ok &= np.array_equal(xg[i], r.standard_normal((N + 1, nx))) and np.array_equal(ug[i], r.standard_normal((N, nu)))
        ok &= cg[i] == 2.0 * i and ig[i] == i % 7
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 9])
def test_shard_and_all_gather_world2(total):
    world, N, nx, nu = 2, 30, 13, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, N, nx, nu, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_shard_ranges_partition_the_batch():
    for total in (1, 7, 8, 1024, 8192):
        for world in (1, 2, 4, 8):
            edges = [sdist.shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    assert sdist.record_words(30, 13, 6) == 585            # 4 680 B per instance
