"""Instance-level data parallelism: MPC instances are independent, so a batch shards into
contiguous blocks, one per rank/GPU, with no collective inside a solve.  The only exchange is one all-gather of the
per-instance solution record [(N+1)*nx + N*nu + 2] doubles (trajectory, cost, iterations) in the batched config
(RCCL over xGMI on GPUs: torch.distributed backend "nccl"; "gloo" on CPU for the tests).
"""
from __future__ import annotations

import numpy as np


def shard_range(total: int, rank: int, world: int):
    """Contiguous block of instance indices owned by `rank` (first `total % world` ranks get one extra)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def record_words(N: int, nx: int, nu: int, mode: str = "full") -> int:
    """doubles per instance record: "full" x | u | cost | iterations; "first_knot" u_0 | x_1 | cost | iterations"""
    return (N + 1) * nx + N * nu + 2 if mode == "full" else nu + nx + 2


def pack_records(x, u, cost, iters):
    """[B, (N+1)*nx + N*nu + 2] float64 records from a solved shard (numpy or torch inputs of matching kind)."""
    if isinstance(x, np.ndarray):
        B = x.shape[0]
        return np.concatenate([x.reshape(B, -1), u.reshape(B, -1), np.asarray(cost, dtype=np.float64).reshape(B, 1),
                               np.asarray(iters, dtype=np.float64).reshape(B, 1)], axis=1)
    import torch
    B = x.shape[0]
    return torch.cat([x.reshape(B, -1), u.reshape(B, -1), cost.reshape(B, 1).to(torch.float64),
                      iters.reshape(B, 1).to(torch.float64)], dim=1).contiguous()


def pack_records_into(out, x, u, cost, iters, mode: str = "full"):
    """pack_records into a preallocated [B, words] torch tensor (host-side form of csrc pack_records_kernel, which the HIP engine
    uses: engine.pack_records_device); -> out"""
    B = x.shape[0]
    if mode == "first_knot":
        nu, nx = u.shape[2], x.shape[2]
        out[:, :nu].copy_(u[:, 0])
        out[:, nu:nu + nx].copy_(x[:, 1])
        out[:, nu + nx].copy_(cost)
        out[:, nu + nx + 1].copy_(iters)
        return out
    nxw, nuw = x.shape[1] * x.shape[2], u.shape[1] * u.shape[2]
    out[:, :nxw].copy_(x.reshape(B, nxw))
    out[:, nxw:nxw + nuw].copy_(u.reshape(B, nuw))
    out[:, nxw + nuw].copy_(cost)
    out[:, nxw + nuw + 1].copy_(iters)             # int32 -> float64
    return out


def unpack_records(rec, N: int, nx: int, nu: int):
    B = rec.shape[0]
    nxw, nuw = (N + 1) * nx, N * nu
    x = rec[:, :nxw].reshape(B, N + 1, nx)
    u = rec[:, nxw:nxw + nuw].reshape(B, N, nu)
    return x, u, rec[:, nxw + nuw], rec[:, nxw + nuw + 1]


def all_gather_records(local, world_sizes=None):
    """One collective: every rank ends with the records of all instances, in instance order.
    `local` is a torch tensor [B_local, W]; equal shard sizes use all_gather_into_tensor (one fused RCCL call)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if world_sizes is None or len(set(world_sizes)) == 1:
        out = torch.empty((world * local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    # uneven shards: pad to the largest shard, gather once, trim (collectives want equal sizes)
    nmax = max(world_sizes)
    padded = torch.zeros((nmax, local.shape[1]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    out = torch.empty((world * nmax, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * nmax:r * nmax + n] for r, n in enumerate(world_sizes)], dim=0)
