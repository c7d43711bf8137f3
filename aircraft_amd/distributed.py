"""Multi-GPU layer of the hot path: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

MPC instances / random restarts are independent, so the batch shards across ranks with NO data-path
collective; the single exchange is one all-gather of every rank's best-K trajectory records after a sweep
(SURVEY.md §8e).  A record at H=100 is ((H+1)*13 + H*7 + 1) * 4 B = 8 056 B, so the gather is latency-bound
(K*8 KB per rank): one flat all-gather is the right shape for point-to-point xGMI — no ring ordering, no bucketing.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

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


def _need_device(what, t, system):
    if not t.is_cuda or system is None:
        raise _lib.AircraftHipError(
            f"{what}: device tensors and `system` (the handle that owns the kernels and the stream) are required — "
            "aircraft_amd has no CPU fallback (the CPU test-suite injects the restatement of oracle/records_oracle.py "
            "through gather_best(pack=, merge=))")


def pack_records(cost, X, U, k: int, system=None):
    """Best-k (lowest cost; NaN counts as +inf, equal costs by instance index) instances of this rank as rows
    [cost, X(H+1,13) flat, U(H,7) flat], ascending: ONE launch of the K6 select + pack kernel (`ac_best_records_f32`)
    straight on the rollout-shaped buffers."""
    torch = _torch()
    _need_device("pack_records", X, system)
    k = min(int(k), cost.numel())
    H, B = U.shape[0], U.shape[2]
    R = 1 + (H + 1) * 13 + H * 7
    lib = system._sync()
    assert cost.is_cuda and cost.dtype == torch.float32 and cost.is_contiguous() and cost.numel() == B
    assert X.is_contiguous() and U.is_contiguous() and X.dtype == torch.float32 and U.dtype == torch.float32
    assert X.shape == (H + 1, 13, B) and U.shape[1] == _lib.NUM_CONTROLS
    rec = torch.empty((k, R), device=X.device, dtype=torch.float32)
    _lib.check(lib.ac_best_records_f32(system._handle, cost.data_ptr(), X.data_ptr(), U.data_ptr(), B, H, k,
                                       rec.data_ptr(), system._stream()), "ac_best_records_f32")
    return rec


def merge_records(rec, system=None):
    """Rows sorted by cost (column 0; NaN -> +inf; stable): the K * world merge after the all-gather
    (`ac_merge_records_f32`)."""
    torch = _torch()
    _need_device("merge_records", rec, system)
    lib = system._sync()
    assert rec.is_contiguous() and rec.dtype == torch.float32
    out = torch.empty_like(rec)
    _lib.check(lib.ac_merge_records_f32(system._handle, rec.data_ptr(), rec.shape[0], rec.shape[1], out.data_ptr(),
                                        system._stream()), "ac_merge_records_f32")
    return out


def unpack_records(rec, H: int):
    """rows -> (cost (n,), X (n, H+1, 13), U (n, H, 7))"""
    n = rec.shape[0]
    nx = (H + 1) * 13
    return rec[:, 0], rec[:, 1 : 1 + nx].reshape(n, H + 1, 13), rec[:, 1 + nx :].reshape(n, H, 7)


def all_gather_records(rec, group=None, single_rank_collective: bool = False):
    """ONE all-gather of the per-rank record block (k, R) -> (world*k, R); identity when not distributed.
    single_rank_collective: issue the collective even in a one-rank group (the identity then) — how the GPU test-suite runs
    the RCCL branch on a one-GPU box."""
    torch = _torch()
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not single_rank_collective):
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
                w_goal: float = 10.0, timing: Optional[dict] = None, pack=None, merge=None):
    """Best-k records of every rank, gathered on all ranks and sorted by cost: (cost, X, U) with world*k rows.
    Three launches and one collective: K6 cost kernel (unless `cost` is supplied), K6 select + pack kernel, ONE
    all_gather_into_tensor, merge kernel.  `timing` (a dict) receives `gather_ms`: HIP-event time of the whole exchange
    on this rank's stream (it synchronises — a measurement aid, not for captured loops).
    `pack` / `merge`: replacements for the two kernels with the same signatures — how the CPU test-suite's gloo ranks
    (no GPU) drive this function's collective logic with the NumPy restatement; the product never passes them."""
    torch = _torch()
    ev = None
    if timing is not None and X.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if cost is None:
        _need_device("gather_best without `cost`", X, system)
        cost = trajectory_cost(system, X, goal, w_track, w_goal)
    rec = (pack or pack_records)(cost, X, U, k, system=system)
    rec = all_gather_records(rec, group=group)
    rec = (merge or merge_records)(rec, system=system)
    if ev is not None:
        ev[1].record()
        ev[1].synchronize()
        timing["gather_ms"] = ev[0].elapsed_time(ev[1])
    return unpack_records(rec, U.shape[0])
