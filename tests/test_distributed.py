"""world_size-2 gloo tests of the multi-GPU layer (sharding + the single all-gather of best records) on CPU."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pytest

from tests import helpers  # noqa: F401  (puts oracle/ on sys.path)
import records_oracle
from aircraft_amd._lib import AircraftHipError
from aircraft_amd.distributed import gather_best, pack_records, shard_bounds, unpack_records


def test_shard_bounds_partition():
    for total in (0, 1, 7, 4096, 16384, 16385):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(16384, 3, 8) == (6144, 8192)  # cfg4: 2048 instances per GPU


def test_record_roundtrip():
    H, B = 5, 9
    X = torch.randn(H + 1, 13, B); U = torch.randn(H, 7, B); cost = torch.arange(B, dtype=torch.float32).flip(0)
    with pytest.raises(AircraftHipError):  # the product's select + pack is a HIP kernel: host tensors are refused, loudly
        pack_records(cost, X, U, k=3)
    rec = records_oracle.pack_records(cost, X, U, k=3)
    assert rec.shape == (3, 1 + (H + 1) * 13 + H * 7)
    c, Xb, Ub = unpack_records(rec, H)
    assert torch.equal(c, torch.tensor([0.0, 1.0, 2.0]))
    assert torch.equal(Xb[0], X[:, :, B - 1]) and torch.equal(Ub[2], U[:, :, B - 3])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, H, B_total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7)
    Xall = torch.randn(H + 1, 13, B_total, generator=g); Uall = torch.randn(H, 7, B_total, generator=g)
    lo, hi = shard_bounds(B_total, rank, world)
    goal = torch.tensor([1.0, -2.0, 0.5])
    Xs, Us = Xall[:, :, lo:hi].contiguous(), Uall[:, :, lo:hi].contiguous()
    # no GPU in this process: the two kernels are replaced by their NumPy restatement; the sharding, the ONE all-gather
    # and the record layout are the product's
    c = torch.from_numpy(records_oracle.trajectory_cost(Xs.numpy(), goal.numpy()).astype(np.float32))
    cost, Xb, Ub = gather_best(Xs, Us, goal, k=2, cost=c, pack=records_oracle.pack_records, merge=records_oracle.merge_records)
    torch.save((cost, Xb, Ub), os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_best_two_ranks_gloo(tmp_path):
    H, B_total, world = 4, 10, 2
    mp.spawn(_worker, args=(world, _free_port(), H, B_total, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt"); r1 = torch.load(tmp_path / "r1.pt")
    for a, b in zip(r0, r1):
        assert torch.equal(a, b)  # every rank ends with the same gathered, sorted records
    cost, Xb, Ub = r0
    assert cost.shape == (4,) and Xb.shape == (4, H + 1, 13) and Ub.shape == (4, H, 7)
    assert torch.all(cost[1:] >= cost[:-1])
    # single-process reference: best-2 of each shard, merged
    g = torch.Generator().manual_seed(7)
    Xall = torch.randn(H + 1, 13, B_total, generator=g); Uall = torch.randn(H, 7, B_total, generator=g)
    d = Xall[:, 0:3, :] - torch.tensor([1.0, -2.0, 0.5])[None, :, None]
    sq = (d * d).sum(1); c = sq.sum(0) + 10.0 * sq[-1]
    want = sorted(sum([sorted(c[lo:hi].tolist())[:2] for lo, hi in (shard_bounds(B_total, r, world) for r in range(world))], []))
    assert np.allclose(cost.numpy(), want, rtol=1e-6)
    best = int(torch.argmin(c))
    assert torch.equal(Xb[0], Xall[:, :, best]) and torch.equal(Ub[0], Uall[:, :, best])
