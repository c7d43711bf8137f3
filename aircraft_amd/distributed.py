"""Multi-GPU layer of the hot path: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

MPC instances / random restarts are independent, so the batch shards across ranks with NO data-path
collective; the single exchange is one all-gather of every rank's best-K trajectory records after a sweep
(SURVEY.md §8e).  A record at H=100 is ((H+1)*13 + H*7 + 1) * 4 B = 8 056 B, so the gather is latency-bound
(K*8 KB per rank): one flat all-gather is the right shape for point-to-point xGMI — no ring ordering, no bucketing.
"""
from __future__ import annotations

import ctypes as C
from typing import Tuple

from . import _lib


def _torch():
    import torch

    return torch


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of `total` instances owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def trajectory_cost(system, X, goal, w_track: float = 1.0, w_goal: float = 10.0):
    """cost[b] = w_track * sum_k |p_k - goal|^2 + w_goal * |p_H - goal|^2 on the device (HIP kernel k_traj_cost).
    X (H+1, 13, B) float32 cuda; goal: 3 floats."""
    torch = _torch()
    lib = system._sync()
    assert X.is_cuda and X.dtype == torch.float32 and X.is_contiguous() and X.shape[1] == system.num_states
    H, B = X.shape[0] - 1, X.shape[2]
    cost = torch.empty((B,), device=X.device, dtype=torch.float32)
    g = (C.c_float * 3)(*[float(v) for v in (goal.tolist() if hasattr(goal, "tolist") else goal)])
    _lib.check(lib.ac_traj_cost_f32(system._handle, X.data_ptr(), B, H, g, C.c_float(w_track), C.c_float(w_goal),
                                    cost.data_ptr(), system._stream()), "ac_traj_cost_f32")
    return cost


def pack_records(cost, X, U, k: int):
    """Best-k (lowest cost) instances of this rank as rows [cost, X(H+1,13) flat, U(H,7) flat]."""
    torch = _torch()
    k = min(int(k), cost.numel())
    vals, idx = torch.topk(cost, k, largest=False, sorted=True)
    Xb = X.index_select(2, idx).permute(2, 0, 1).reshape(k, -1)  # (k, (H+1)*13)
    Ub = U.index_select(2, idx).permute(2, 0, 1).reshape(k, -1)  # (k, H*7)
    return torch.cat([vals[:, None], Xb, Ub], dim=1).contiguous()


def unpack_records(rec, H: int):
    """rows -> (cost (n,), X (n, H+1, 13), U (n, H, 7))"""
    n = rec.shape[0]
    nx = (H + 1) * 13
    return rec[:, 0], rec[:, 1 : 1 + nx].reshape(n, H + 1, 13), rec[:, 1 + nx :].reshape(n, H, 7)


def all_gather_records(rec, group=None):
    """ONE all-gather of the per-rank record block (k, R) -> (world*k, R); identity when not distributed."""
    torch = _torch()
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return rec
    world = dist.get_world_size(group)
    if rec.is_cuda and dist.get_backend(group) == "gloo":  # gloo rehearsal of the GPU path: stage through the host
        host = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype)
        dist.all_gather_into_tensor(host, rec.cpu(), group=group)
        return host.to(rec.device)
    out = torch.empty((world * rec.shape[0], rec.shape[1]), device=rec.device, dtype=rec.dtype)
    dist.all_gather_into_tensor(out, rec, group=group)
    return out


def gather_best(X, U, goal, k: int = 1, system=None, cost=None, group=None, w_track: float = 1.0,
                w_goal: float = 10.0):
    """Best-k records of every rank, gathered on all ranks and sorted by cost: (cost, X, U) with
    world*k rows.  `cost` may be supplied; otherwise it is evaluated on the device (needs `system`)
    or, for host tensors in the CPU test-suite, with the same formula in torch."""
    torch = _torch()
    if cost is None:
        if system is not None and X.is_cuda:
            cost = trajectory_cost(system, X, goal, w_track, w_goal)
        else:
            d = X[:, 0:3, :] - torch.as_tensor(goal, dtype=X.dtype, device=X.device)[None, :, None]
            sq = (d * d).sum(dim=1)  # (H+1, B)
            cost = w_track * sq.sum(dim=0) + w_goal * sq[-1]
    rec = all_gather_records(pack_records(cost, X, U, k), group=group)
    order = torch.argsort(rec[:, 0])
    return unpack_records(rec[order], U.shape[0])
