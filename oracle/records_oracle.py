"""NumPy restatement of the K6 selection kernels (aircraft_amd/csrc/ac_select.hpp) — TEST INFRASTRUCTURE.

Only tests/ may import this file: the GPU tests compare the kernels with it, and the CPU test-suite's gloo ranks (no
GPU) inject it into aircraft_amd.distributed.gather_best(pack=, merge=) to exercise the collective logic.
There is no reference counterpart (the reference solves one instance, no selection, no exchange): build-side.
"""
import numpy as np
import torch


def trajectory_cost(X, goal, w_track=1.0, w_goal=10.0):
    """cost[b] = w_track * sum_k |p_k - goal|^2 + w_goal * |p_H - goal|^2  (k_traj_cost), float64."""
    Xn = np.asarray(X, dtype=np.float64)
    d = Xn[:, 0:3, :] - np.asarray(goal, dtype=np.float64)[None, :, None]
    sq = (d * d).sum(axis=1)
    return w_track * sq.sum(axis=0) + w_goal * sq[-1]


def best_order(cost, k):
    """Indices of the k lowest costs: NaN counts as +inf, equal costs are ordered by index (k_best_records)."""
    c = np.asarray(cost, dtype=np.float32)
    c = np.where(np.isnan(c), np.float32(np.inf), c)
    return np.lexsort((np.arange(c.size), c))[:k], c


def pack_records(cost, X, U, k, system=None):
    """rows [cost, X(H+1,13) flat, U(H,7) flat] of the k best instances, ascending; torch host tensors in and out."""
    Xn, Un = X.cpu().numpy(), U.cpu().numpy()
    k = min(int(k), Xn.shape[2])
    idx, c = best_order(cost.cpu().numpy(), k)
    rec = np.concatenate([c[idx, None], Xn[:, :, idx].transpose(2, 0, 1).reshape(k, -1),
                          Un[:, :, idx].transpose(2, 0, 1).reshape(k, -1)], axis=1)
    return torch.from_numpy(np.ascontiguousarray(rec, dtype=np.float32))


def merge_records(rec, system=None):
    """rows sorted by column 0 (NaN -> +inf), stable (k_merge_records)."""
    r = rec.cpu().numpy().copy()
    r[:, 0] = np.where(np.isnan(r[:, 0]), np.float32(np.inf), r[:, 0])
    return torch.from_numpy(np.ascontiguousarray(r[np.argsort(r[:, 0], kind="stable")]))


def accept(Jc, J0, Xc, Uc, X, U):
    """Line-search acceptance (k_ilqr_accept): returns (Jout, improved, X_new, U_new) as numpy arrays."""
    Jc, J0 = np.asarray(Jc, dtype=np.float32), np.asarray(J0, dtype=np.float32)
    B = J0.size
    na = Jc.size // B
    J = np.where(np.isfinite(Jc), Jc, np.float32(np.inf)).reshape(na, B)
    arg = J.argmin(axis=0)  # first minimum: ties -> lowest a
    best = J[arg, np.arange(B)]
    with np.errstate(invalid="ignore"):
        imp = best < J0
    col = arg * B + np.arange(B)
    Xn = np.where(imp[None, None, :], np.asarray(Xc)[:, :, col], np.asarray(X))
    Un = np.where(imp[None, None, :], np.asarray(Uc)[:, :, col], np.asarray(U))
    return np.where(imp, best, J0), imp, Xn, Un
