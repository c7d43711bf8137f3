"""GPU side of the multi-GPU layer: the K6 cost / top-K kernel against the torch formula, the record
packing on device tensors, and a 2-rank rehearsal of bench.py on ONE GPU (both ranks on cuda:0, collectives over gloo)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import ROOT, make_aircraft, near_trim_problem

pytestmark = pytest.mark.gpu


def _torch_cost(X, goal, w_track, w_goal):
    import torch

    d = X[:, 0:3, :].double() - torch.as_tensor(goal, dtype=torch.float64, device=X.device)[None, :, None]
    sq = (d * d).sum(dim=1)  # (H+1, B)
    return w_track * sq.sum(dim=0) + w_goal * sq[-1]


@pytest.mark.parametrize("B,H", [(1, 1), (37, 5), (300, 50), (2048, 100)])
def test_traj_cost_kernel_matches_the_torch_formula(gpu, B, H):
    """ac_traj_cost_f32 (k_traj_cost) on a real rollout: cost[b] = w_track sum_k |p_k - g|^2 + w_goal |p_H - g|^2."""
    import torch

    from aircraft_amd.distributed import trajectory_cost

    ac = make_aircraft("poly", normalise=True)
    X0, U = near_trim_problem(B, H, seed=5)  # trajectories that stay finite (random ones tumble and overflow)
    X = ac.rollout(torch.from_numpy(X0).float().to(gpu), torch.from_numpy(U).float().to(gpu), 0.01)
    goal = (150.0, 10.0, -190.0)
    for wt, wg in ((1.0, 10.0), (0.25, 0.0), (0.0, 3.0)):
        cost = trajectory_cost(ac, X, goal, wt, wg)
        want = _torch_cost(X, goal, wt, wg)
        assert cost.shape == (B,) and cost.dtype == torch.float32 and torch.isfinite(want).all()
        rel = ((cost.double() - want).abs() / want.abs().clamp_min(1e-12)).max().item()
        assert rel < 2e-6, rel  # fp32 running sum of H+1 positive terms vs float64


def test_traj_cost_status_codes_and_nan(gpu):
    import ctypes as C

    import torch

    from aircraft_amd import _lib

    lib = _lib.load()
    ac = make_aircraft("default")
    ac._sync()
    h, st = ac._handle, ac._stream()
    X = torch.zeros((4, 13, 8), device=gpu)
    X[2, 1, 3] = float("nan")
    cost = torch.zeros(8, device=gpu)
    g = (C.c_float * 3)(0, 0, 0)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    assert lib.ac_traj_cost_f32(h, p(X), 8, 3, g, C.c_float(1), C.c_float(1), p(cost), st) == 0
    torch.cuda.synchronize()
    c = cost.cpu().numpy()
    assert np.isnan(c[3]) and np.isfinite(np.delete(c, 3)).all()  # NaNs propagate, per instance
    assert lib.ac_traj_cost_f32(h, None, 8, 3, g, C.c_float(1), C.c_float(1), p(cost), st) == -1
    assert lib.ac_traj_cost_f32(h, p(X), -1, 3, g, C.c_float(1), C.c_float(1), p(cost), st) == -1
    assert lib.ac_traj_cost_f32(h, None, 0, 3, None, C.c_float(1), C.c_float(1), None, st) == 0  # empty batch


def test_gather_best_on_device_records(gpu):
    """cost kernel -> top-K -> record packing on device tensors (single process: the gather is the identity)."""
    import torch

    from aircraft_amd.distributed import gather_best, pack_records, trajectory_cost, unpack_records

    ac = make_aircraft("poly", normalise=True)
    B, H, k = 512, 20, 4
    X0, U = near_trim_problem(B, H, seed=9)
    Ud = torch.from_numpy(U).float().to(gpu)
    X = ac.rollout(torch.from_numpy(X0).float().to(gpu), Ud, 0.01)
    goal = torch.tensor([150.0, 10.0, -190.0], device=gpu)
    cost = trajectory_cost(ac, X, goal)
    rec = pack_records(cost, X, Ud, k, system=ac)
    assert ac.last_launch()[0] == "k_best_records"
    assert rec.is_cuda and rec.shape == (k, 1 + (H + 1) * 13 + H * 7)
    c, Xb, Ub = unpack_records(rec, H)
    order = torch.argsort(_torch_cost(X, goal.tolist(), 1.0, 10.0))[:k]
    assert torch.equal(torch.argsort(cost)[:k], order)
    for i, b in enumerate(order.tolist()):
        assert torch.equal(Xb[i], X[:, :, b]) and torch.equal(Ub[i], Ud[:, :, b]) and c[i] == cost[b]
    c2, Xb2, Ub2 = gather_best(X, Ud, goal, k=k, system=ac)
    assert torch.equal(c2, c) and torch.equal(Xb2, Xb) and torch.equal(Ub2, Ub)


@pytest.mark.parametrize("B,H,k", [(512, 20, 4), (16384, 100, 8), (2048, 100, 1), (9, 3, 8), (8, 1, 8), (70001, 2, 5)])
def test_best_records_kernel_equals_torch_topk(gpu, B, H, k):
    """K6's second half (ac_best_records_f32: per-rank best-K select + record pack in one launch) bit-equal to
    torch.topk + index_select + permute + cat on the same buffers; NaN costs count as +inf; equal costs are ordered by
    instance index (checked against the NumPy restatement, torch.topk leaves ties unspecified); then the merge kernel
    against a stable argsort."""
    import torch
    import records_oracle
    from aircraft_amd.distributed import merge_records, pack_records

    ac = make_aircraft("default")
    g = torch.Generator(device="cpu").manual_seed(B + H)
    X = torch.randn(H + 1, 13, B, generator=g).to(gpu); U = torch.randn(H, 7, B, generator=g).to(gpu)
    cost = torch.rand(B, generator=g).to(gpu) * 100
    cost[3 % B] = float("nan")                      # a crashed rollout never wins ...
    if B > 600:
        cost[500] = float("-inf"); cost[77] = float("inf")
    rec = pack_records(cost, X, U, k, system=ac)
    san = torch.nan_to_num(cost, nan=float("inf"), posinf=float("inf"), neginf=float("-inf"))
    vals, idx = torch.topk(san, k, largest=False, sorted=True)
    want = torch.cat([vals[:, None], X.index_select(2, idx).permute(2, 0, 1).reshape(k, -1),
                      U.index_select(2, idx).permute(2, 0, 1).reshape(k, -1)], dim=1)
    assert torch.equal(rec, want)                   # bit-equal, including the record layout
    assert not torch.isnan(rec[:, 0]).any()         # ... unless K reaches it, and then as +inf
    # ties: quantised costs, many equal values -> index order decides (the restatement's order)
    tied = (cost * 0.05).floor().nan_to_num(nan=7.0)
    rec_t = pack_records(tied, X, U, k, system=ac)
    want_t = records_oracle.pack_records(tied.cpu(), X.cpu(), U.cpu(), k)
    assert torch.equal(rec_t.cpu(), want_t)
    # merge: K * world gathered rows, sorted by cost, stable; NaN last
    rows = torch.cat([rec_t, rec, rec_t.flip(0)]).contiguous()
    rows[1, 0] = float("nan")
    merged = merge_records(rows, system=ac)
    assert ac.last_launch()[0] == "k_merge_records"
    assert torch.equal(merged.cpu(), records_oracle.merge_records(rows.cpu()))
    # argument checks: K > 8, K > B, aliasing
    lib, hnd, st = ac._sync(), ac._handle, ac._stream()
    assert lib.ac_best_records_f32(hnd, cost.data_ptr(), X.data_ptr(), U.data_ptr(), B, H, 9, rec.data_ptr(), st) == -1
    assert lib.ac_best_records_f32(hnd, cost.data_ptr(), X.data_ptr(), U.data_ptr(), 2, H, 3, rec.data_ptr(), st) == -1
    assert lib.ac_best_records_f32(hnd, None, None, None, 0, H, 0, None, st) == 0
    assert lib.ac_merge_records_f32(hnd, rows.data_ptr(), rows.shape[0], rows.shape[1], rows.data_ptr(), st) == -1


@pytest.mark.parametrize("B,H,na", [(24, 30, 3), (1024, 50, 3), (300, 7, 8), (1, 1, 1)])
def test_line_search_accept_kernel(gpu, B, H, na):
    """ac_ilqr_accept_f32 (per-instance argmin over the candidates + conditional in-place copy) against the NumPy
    restatement: NaN / inf candidates never win, ties take the lowest alpha index, non-improving instances keep their
    iterate bit for bit."""
    import ctypes as C
    import torch
    import records_oracle

    ac = make_aircraft("default")
    lib = ac._sync()
    g = torch.Generator(device="cpu").manual_seed(11 * B + na)
    Xc = torch.randn(H + 1, 13, na * B, generator=g).to(gpu); Uc = torch.randn(H, 7, na * B, generator=g).to(gpu)
    X = torch.randn(H + 1, 13, B, generator=g).to(gpu); U = torch.randn(H, 7, B, generator=g).to(gpu)
    Jc = (torch.rand(na * B, generator=g) * 4).round().to(gpu)   # few distinct values: ties across alphas
    J0 = torch.full((B,), 2.0, device=gpu)
    Jc[0] = float("nan")
    if B > 5:
        Jc[1] = float("-inf"); Jc[2] = float("inf"); J0[4] = float("nan")
    wJ, wimp, wX, wU = records_oracle.accept(Jc.cpu().numpy(), J0.cpu().numpy(), Xc.cpu().numpy(), Uc.cpu().numpy(),
                                             X.cpu().numpy(), U.cpu().numpy())
    Jout = torch.empty(B, device=gpu); imp = torch.zeros(B, device=gpu, dtype=torch.bool)
    rc = lib.ac_ilqr_accept_f32(ac._handle, Jc.data_ptr(), J0.data_ptr(), Xc.data_ptr(), Uc.data_ptr(), na, B, H,
                                X.data_ptr(), U.data_ptr(), Jout.data_ptr(), imp.data_ptr(), ac._stream())
    assert rc == 0
    assert np.array_equal(imp.cpu().numpy(), wimp) and (B == 1 or 0 < wimp.sum() < B)
    assert np.array_equal(Jout.cpu().numpy(), wJ, equal_nan=True)
    assert np.array_equal(X.cpu().numpy(), wX) and np.array_equal(U.cpu().numpy(), wU)


def test_rccl_branch_of_the_record_exchange_on_one_rank(gpu):
    """The `nccl` (= RCCL) branch of all_gather_records — the only branch an 8-GPU run takes — has never run on the
    one-GPU boxes of this pool.  A one-rank RCCL group makes the collective the identity but runs the real call: device
    tensors of the record layout through dist.all_gather_into_tensor on the RCCL backend, in a process of its own."""
    code = """
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from tests.helpers import make_aircraft
from aircraft_amd.distributed import all_gather_records, merge_records, pack_records, unpack_records
ac = make_aircraft("default")
g = torch.Generator().manual_seed(3)
H, B, k = 20, 300, 4
X = torch.randn(H + 1, 13, B, generator=g).cuda(); U = torch.randn(H, 7, B, generator=g).cuda(); cost = torch.rand(B, generator=g).cuda()
rec = pack_records(cost, X, U, k, system=ac)
out = all_gather_records(rec, single_rank_collective=True)
assert dist.get_backend() == "nccl" and out.is_cuda and out.data_ptr() != rec.data_ptr() and torch.equal(out, rec)
c, Xb, Ub = unpack_records(merge_records(out, system=ac), H)
order = torch.argsort(cost)[:k]
assert torch.equal(c, cost[order]) and torch.equal(Xb[0], X[:, :, order[0]])
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK")
""" % (ROOT, str(29500 + os.getpid() % 2000))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def _run_bench(extra, env_extra, timeout=600):
    env = dict(os.environ)
    env.update(env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "512",
                        "--no-cpu-baseline", "--no-extras", *extra], env=env, capture_output=True, text=True,
                       timeout=timeout)
    return r


def test_bench_two_rank_rehearsal_on_one_gpu(gpu):
    """`python bench.py --gpus 2` with no launcher starts two fresh ranks itself (here both on cuda:0, gloo collectives)
    and reports n_gpus = ranks_seen = 2 with weak- and strong-scaled figures; the N = 1 line has the same keys."""
    r = _run_bench(["--gpus", "2"], {"AIRCRAFT_BENCH_ONE_GPU": "1", "AIRCRAFT_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2 and res["scaling"] == "weak"
    assert res["config"]["units_per_step"] == 2 * 512 * 50 and res["config"]["batch_per_gpu"] == 512
    s = res["strong"]
    assert s["scaling"] == "strong" and s["batch_total"] == 512 and s["batch_per_gpu"] == 256
    assert s["units_per_step"] == 512 * 50 and s["value"] > 0 and res["value"] > 0
    assert res["roofline"]["frac"] > 0 and "cpu_baseline" not in res  # baseline leg is rank-0-at-N=1 only


def test_bench_refuses_a_mislabelled_run(gpu):
    """One rank launched as if it were one of two (--gpus 2 under WORLD_SIZE=1) must fail, not print n_gpus 1."""
    # the child sees ONE device whatever the box has (on a 2/4/8-GPU node the self-launch would legitimately succeed):
    # without the rehearsal switches one GPU cannot host two ranks: every rank exits 3, the parent relays it
    r = _run_bench(["--gpus", "2"], {"HIP_VISIBLE_DEVICES": "0", "ROCR_VISIBLE_DEVICES": "0"})
    assert r.returncode != 0 and "need" in r.stderr
    env = {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"}
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-extras"], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr and not r.stdout.strip()
