"""GPU tests at BASELINE.json's full sizes (cfg3: B=4096, H=50, 4x128 MLP; cfg4 shard: B=2048, H=100) through
size-independent properties, plus an oracle spot check on a random sample of the full-size result."""
import numpy as np
import pytest

from tests.helpers import (block_rel_err, check_against_conditioning, conditioning, f32_exact, make_aircraft, make_oracle,
                           parity_report, unit_max_rel, unit_rowblock_rel)
from aircraft_amd.synthetic import synthetic_controls, synthetic_states, synthetic_units

pytestmark = pytest.mark.gpu


def _problem(B, H, gpu, seed=42):
    import torch

    rng = np.random.default_rng(seed)
    Xh = f32_exact(synthetic_states(B * (H + 1), rng).reshape(13, H + 1, B).transpose(1, 0, 2))
    Uh = f32_exact(synthetic_controls(H, B, rng))
    return Xh, Uh, torch.from_numpy(np.ascontiguousarray(Xh, dtype=np.float32)).to(gpu), \
        torch.from_numpy(np.ascontiguousarray(Uh, dtype=np.float32)).to(gpu)


@pytest.fixture(scope="module")
def cfg3(gpu):
    from aircraft_amd.control import MultipleShooting

    ac = make_aircraft("nn", hidden=(128, 128, 128, 128))
    ms = MultipleShooting(system=ac, dt=0.01, num_nodes=50, opts={"quaternion": "integration"})
    Xh, Uh, X, U = _problem(4096, 50, gpu)
    F, A, Bm, c = ms.linearise(X, U)
    return dict(ac=ac, ms=ms, Xh=Xh, Uh=Uh, X=X, U=U, F=F, A=A, Bm=Bm, c=c)


def test_cfg3_oracle_spot_check(cfg3):
    """512 random (instance, node) units of the 204 800 against the float64 oracle."""
    rng = np.random.default_rng(0)
    k = rng.integers(0, 50, 512); b = rng.integers(0, 4096, 512)
    Xs = np.ascontiguousarray(cfg3["Xh"][k, :, b].T); Us = np.ascontiguousarray(cfg3["Uh"][k, :, b].T)
    Xr, Ar, Br, cr = make_oracle(cfg3["ac"]).step_sens(Xs, Us, 0.01)
    F = cfg3["F"].cpu().numpy()[k, :, b].T
    A = cfg3["A"].cpu().numpy()[k, :, :, b].transpose(1, 2, 0)
    Bm = cfg3["Bm"].cpu().numpy()[k, :, :, b].transpose(1, 2, 0)
    c = cfg3["c"].cpu().numpy()[k, :, b].T
    assert block_rel_err(F, Xr) < 1e-5
    for key, g, w in (("A", A, Ar), ("B", Bm, Br), ("c", c, cr)):  # every sampled unit on its own
        e, eb = unit_max_rel(g, w), unit_rowblock_rel(g, w)
        parity_report("cfg3_spot_check", block=key, units=512, unit_rel_max=float(e.max()), rowblock_rel_max=float(eb.max()))
        assert e.max() < 1e-5 and eb.max() < 1e-4, (key, float(e.max()), float(eb.max()))


def test_cfg3_second_order_blocks_at_full_size(cfg3):
    """The nlp_hess_l blocks of all 204 800 units in one call (ac_shoot_hess_f32: 3 200 tasks on a persistent grid of one
    workgroup per CU — every workgroup runs 12-13 tasks through its scratch slots and the hand-sequenced weight ring): 48 units
    spread over the first, middle and last tasks against central differences of the oracle's exact float64 Jacobians, and the
    same units evaluated on their own (one task per workgroup) bit for bit."""
    import torch

    from tests.helpers import oracle_step_hessian

    ac, ms, X, U = cfg3["ac"], cfg3["ms"], cfg3["X"], cfg3["U"]
    H, B = 50, 4096
    rng = np.random.default_rng(4)
    lam_h = f32_exact(rng.normal(size=(H, 13, B)))
    Lam = torch.from_numpy(np.ascontiguousarray(lam_h, dtype=np.float32)).to(X.device)
    Hz = ms.hessian(X, U, Lam)
    torch.cuda.synchronize()
    assert Hz.shape == (H, 21, 21, B) and bool(torch.isfinite(Hz).all())
    # sampled units: flat unit u = k * B + b (node-major); tasks are 64 consecutive units
    flat = np.concatenate([np.arange(0, 16), 64 * 1599 + np.arange(20, 36), H * B - 16 + np.arange(0, 16)])
    k, b = flat // B, flat % B
    Xs = np.ascontiguousarray(cfg3["Xh"][k, :, b].T); Us = np.ascontiguousarray(cfg3["Uh"][k, :, b].T)
    ls = np.ascontiguousarray(lam_h[k, :, b].T)
    got = Hz.cpu().numpy()[k, :, :, b].transpose(1, 2, 0).astype(np.float64)
    want = oracle_step_hessian(make_oracle(ac), Xs, Us, 0.01, ls)
    num = np.sqrt(((got - want) ** 2).sum(axis=(0, 1))); den = np.sqrt((want ** 2).sum(axis=(0, 1)))
    rel = float((num / np.maximum(den, 1e-30)).max())
    parity_report("cfg3_hess_full_size", units=len(flat), rel_block_max=rel)
    assert rel < 5e-4
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(X.device)  # noqa: E731
    alone = ac.step_hess(dev(Xs), dev(Us), 0.01, dev(ls))
    assert torch.equal(alone, torch.from_numpy(np.ascontiguousarray(Hz.cpu().numpy()[k, :, :, b].transpose(1, 2, 0))).to(X.device))


def test_cfg3_deterministic_and_shard_invariant(cfg3):
    """Same inputs -> same bits; and a unit's result does not depend on which batch it is evaluated in (the
    multi-GPU sharding relies on it): the full batch equals its two halves evaluated separately."""
    import torch

    ms, X, U = cfg3["ms"], cfg3["X"], cfg3["U"]
    F2, A2, B2, c2 = ms.linearise(X, U)
    assert torch.equal(F2, cfg3["F"]) and torch.equal(A2, cfg3["A"]) and torch.equal(B2, cfg3["Bm"])
    for lo, hi in ((0, 2048), (2048, 4096)):
        Fh, Ah, Bh, ch = ms.linearise(X[:, :, lo:hi].contiguous(), U[:, :, lo:hi].contiguous())
        assert torch.equal(Fh, cfg3["F"][:, :, lo:hi]) and torch.equal(Ah, cfg3["A"][:, :, :, lo:hi])
        assert torch.equal(Bh, cfg3["Bm"][:, :, :, lo:hi]) and torch.equal(ch, cfg3["c"][:, :, lo:hi])


def test_cfg3_structure_and_norm(cfg3):
    import torch

    F, A, Bm = cfg3["F"], cfg3["A"], cfg3["Bm"]
    assert torch.isfinite(F).all() and torch.isfinite(A).all() and torch.isfinite(Bm).all()
    assert float((F[:, 6:10].norm(dim=1) - 1).abs().max()) < 1e-6  # q normalised (quaternion == 'integration')
    eye = torch.eye(13, device=A.device)[:, :3]
    assert torch.equal(A[:, :, :3, :], eye[None, :, :, None].expand(50, 13, 3, 4096))  # dF/dp = [I; 0]
    assert not Bm[:, :, 3:6, :].any()  # dF/dthrust = 0
    # the normalised quaternion is orthogonal to its own tangent: q+^T dq+/d(anything) = 0
    qT = F[:, 6:10]
    assert float(torch.einsum("kib,kijb->kjb", qT, A[:, 6:10]).abs().max()) < 2e-5
    assert float(torch.einsum("kib,kijb->kjb", qT, Bm[:, 6:10]).abs().max()) < 2e-5


def test_cfg3_jacobian_is_the_derivative_of_the_gpu_step(cfg3):
    """Linearity check at full size, no oracle: F(x + d) - F(x - d) ~= 2 A d along a random direction."""
    import torch

    ms, X, U = cfg3["ms"], cfg3["X"], cfg3["U"]
    g = torch.Generator(device="cpu").manual_seed(3)
    d = torch.randn(13, generator=g).to(X.device) * torch.tensor([1, 1, 1, 0.05, 0.05, 0.05, 2e-3, 2e-3, 2e-3, 2e-3, 0.02, 0.02, 0.02], device=X.device)
    dX = d[None, :, None].expand(50, 13, 4096)
    Fp = ms.propagate((X[:50] + dX).contiguous(), U)
    Fm = ms.propagate((X[:50] - dX).contiguous(), U)
    lin = 2 * torch.einsum("kijb,j->kib", cfg3["A"], d)
    num = (Fp - Fm)
    scale = lin.abs().amax(dim=(0, 2), keepdim=True).clamp_min(1e-6)
    # fp32 central difference: truncation + cancellation noise, so a coarse bound — but a wrong Jacobian is O(1) off
    assert float(((num - lin).abs() / scale).quantile(0.999)) < 5e-2
    assert float(((num - lin).abs() / scale).median()) < 2e-3


def test_cfg3_rollout_chains_the_step(cfg3):
    """Full-size rollout (cooperative kernel): every node is the step of its predecessor to fp32 rounding."""
    import torch

    ms, X, U = cfg3["ms"], cfg3["X"], cfg3["U"]
    traj = ms.rollout(X[0].contiguous(), U)
    assert traj.shape == (51, 13, 4096) and torch.equal(traj[0], X[0])
    Fchain = ms.propagate(traj, U)
    fin = torch.isfinite(traj[1:]).all(dim=1) & torch.isfinite(Fchain).all(dim=1)
    d = (Fchain - traj[1:]).abs()
    scale = traj[1:].abs().clamp_min(1.0)
    assert float((d / scale)[fin[:, None, :].expand_as(d)].max()) < 5e-6
    assert float((traj[-1, 6:10].norm(dim=0) - 1).abs()[fin[-1]].max()) < 1e-6


def test_cfg3_full_rollout_against_the_oracle(cfg3):
    """The whole cfg3 rollout — 4096 instances x 50 nodes, cooperative register-resident kernel — against the float64
    oracle on EVERY instance (2e5 float64 steps: seconds on the host), no mask: an instance is within 1e-5 at every node
    or within 8 x the deviation the float64 reference itself shows under a one-ulp perturbation of x0."""
    ms, X, U = cfg3["ms"], cfg3["X"], cfg3["U"]
    traj = ms.rollout(X[0].contiguous(), U).cpu().numpy()
    orc = make_oracle(cfg3["ac"])
    X0, Uh = cfg3["Xh"][0], cfg3["Uh"]
    ref, cond = conditioning(orc, np.ascontiguousarray(X0), np.ascontiguousarray(Uh), 0.01, draws=2)
    check_against_conditioning("cfg3_full_rollout[4096x50]", traj, ref, cond, 1e-5, min_frac=0.9)


def test_cfg4_shard_shape(gpu):
    """cfg4: one rank's shard of B=16384 (2048 instances), H=100: shapes, finiteness, oracle sample."""
    from aircraft_amd.control import MultipleShooting
    from aircraft_amd.distributed import shard_bounds

    lo, hi = shard_bounds(16384, 5, 8)
    assert hi - lo == 2048
    ac = make_aircraft("nn", hidden=(128, 128, 128, 128))
    ms = MultipleShooting(system=ac, dt=0.01, num_nodes=100, opts={"quaternion": "integration"})
    Xh, Uh, X, U = _problem(2048, 100, gpu, seed=5)
    F, A, Bm, c = ms.linearise(X, U, want_c=False)
    assert F.shape == (100, 13, 2048) and A.shape == (100, 13, 13, 2048) and c is None
    rng = np.random.default_rng(1)
    k = rng.integers(0, 100, 128); b = rng.integers(0, 2048, 128)
    Xr, Ar, Br, _ = make_oracle(ac).step_sens(np.ascontiguousarray(Xh[k, :, b].T), np.ascontiguousarray(Uh[k, :, b].T), 0.01)
    assert block_rel_err(F.cpu().numpy()[k, :, b].T, Xr) < 1e-5
    assert unit_max_rel(A.cpu().numpy()[k, :, :, b].transpose(1, 2, 0), Ar).max() < 1e-5
    assert unit_max_rel(Bm.cpu().numpy()[k, :, :, b].transpose(1, 2, 0), Br).max() < 1e-5


def test_cfg4_shard_solve_cost_topk_and_gather(gpu):
    """cfg4 on one rank: its shard of the 16 384 random restarts (2048 instances), H = 100, 4x128 surrogate — a batched
    solve of 2 iterations, the K6 trajectory-cost kernel, top-K selection and the record exchange (identity at world
    size 1).  The winning records are dynamically consistent: the float64 oracle's rollout of the winning controls
    reproduces the winning states."""
    import torch

    from aircraft_amd.control import ILQR, QuadraticCost
    from aircraft_amd.distributed import gather_best, shard_bounds, trajectory_cost
    from aircraft_amd.synthetic import TRIM_STATE

    lo, hi = shard_bounds(16384, 3, 8)
    B, H, k = hi - lo, 100, 4
    ac = make_aircraft("nn", hidden=(128, 128, 128, 128))
    cost = QuadraticCost.goal((50.0, 2.0), w_goal=1.0, height=-200.0, w_height=1.0, w_lateral_speed=0.5, r=0.5, reg=1.0)
    solver = ILQR(system=ac, dt=0.01, num_nodes=H, cost=cost, alphas=(1.0, 0.5, 0.1))
    rng = np.random.default_rng(1234)
    U_all = np.zeros((H, 7, 16384), dtype=np.float32)
    U_all[:, :3] = np.clip(np.cumsum(rng.normal(0, 0.3, (H, 3, 16384)), axis=0), -5, 5)
    x0 = torch.from_numpy(np.repeat(TRIM_STATE[:, None], B, axis=1).astype(np.float32)).to(gpu)
    U0 = torch.from_numpy(np.ascontiguousarray(U_all[:, :, lo:hi])).to(gpu)
    X, U, hist = solver.solve(x0, U0, iters=2)
    h = hist.cpu().numpy()
    assert X.shape == (H + 1, 13, B) and U.shape == (H, 7, B) and h.shape == (3, B)
    fin = np.isfinite(h).all(axis=0)
    assert fin.mean() > 0.95
    assert (np.diff(h[:, fin], axis=0) <= 1e-6 * np.abs(h[:-1, fin]) + 1e-6).all()  # monotone per instance
    assert np.median(h[-1, fin] / h[0, fin]) < 0.9
    # the one exchange: best-k by the solver's own cost, and by the K6 goal-distance kernel
    c, Xb, Ub = gather_best(X, U, None, k=k, cost=hist[-1].contiguous(), system=ac)
    assert c.shape == (k,) and Xb.shape == (k, H + 1, 13) and Ub.shape == (k, H, 7)
    order = torch.argsort(torch.nan_to_num(hist[-1], nan=float("inf")))[:k]
    assert torch.equal(c, hist[-1][order]) and torch.equal(Xb[0], X[:, :, order[0]]) and torch.equal(Ub[0], U[:, :, order[0]])
    goal = torch.tensor([50.0, 2.0, -200.0], device=gpu)
    ck = trajectory_cost(ac, X, goal)
    d = X[:, 0:3, :].double() - goal.double()[None, :, None]
    want = (d * d).sum(1).sum(0) + 10.0 * (d[-1] * d[-1]).sum(0)
    okc = torch.isfinite(want)
    assert float(((ck.double() - want).abs() / want.abs().clamp_min(1e-9))[okc].max()) < 5e-6
    c2, Xb2, Ub2 = gather_best(X, U, goal, k=k, system=ac)
    assert torch.equal(c2, ck[torch.argsort(torch.nan_to_num(ck, nan=float("inf")))[:k]])
    # dynamic consistency of the winners against the float64 oracle
    Uw = np.ascontiguousarray(Ub.cpu().numpy().astype(np.float64).transpose(1, 2, 0))  # (H, 7, k)
    ref = make_oracle(ac).rollout(np.repeat(TRIM_STATE[:, None], k, axis=1), Uw, 0.01)
    got = Xb.cpu().numpy().astype(np.float64).transpose(1, 2, 0)
    parity_report("cfg4_winners_vs_oracle", err=block_rel_err(got, ref))
    assert block_rel_err(got, ref) < 1e-5


def test_hipgraph_capture_of_the_inner_step(gpu):
    """cfg5: the rollout + linearise step of the receding-horizon loop is hipGraph-capturable (no allocation or
    synchronisation inside the entry points) and replays bit-identically."""
    import torch
    from aircraft_amd.control import MultipleShooting

    ac = make_aircraft("nn", hidden=(128, 128, 128, 128))
    B, H = 1024, 50
    ms = MultipleShooting(system=ac, dt=0.01, num_nodes=H, opts={"quaternion": "integration"})
    _, _, X, U = _problem(B, H, gpu, seed=9)
    x0 = X[0].contiguous().clone()
    traj = torch.empty((H + 1, 13, B), device=gpu)
    out = (torch.empty((H, 13, B), device=gpu), torch.empty((H, 13, 13, B), device=gpu),
           torch.empty((H, 13, 7, B), device=gpu), None)

    def step():
        ms.rollout(x0, U, out=traj)
        ms.linearise(traj, U, out=out)

    step()
    torch.cuda.synchronize()
    ref = [t.clone() for t in (traj, out[0], out[1], out[2])]
    from aircraft_amd.control.moving_horizon import quiet_capture

    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
        torch.cuda.synchronize()
        with quiet_capture(g, s):  # (garbage collector paused: a finaliser's hipFree would invalidate the capture)
            step()
    torch.cuda.current_stream().wait_stream(s)
    for t in (traj, out[0], out[1], out[2]):
        t.zero_()
    g.replay()
    torch.cuda.synchronize()
    for got, want in zip((traj, out[0], out[1], out[2]), ref):
        assert torch.equal(got, want)
    # replay with a new initial state written into the captured buffer
    x0.copy_(X[7])
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(traj[0], X[7]) and not torch.equal(traj[1], ref[0][1])


@pytest.mark.parametrize("hidden", [None, (128, 128, 128, 128)])
def test_forward_64_units_per_wave_path(gpu, hidden):
    """Large n takes the four-slab forward kernels (lane = unit): step, derivative and getters against the oracle on a
    sample, and bit-equality with the 16-units-per-wave kernels that smaller batches use."""
    import torch
    from tests.helpers import synthetic_units

    ac = make_aircraft("nn", hidden=hidden, normalise=True, stall_scaling=True)
    n = 300001  # ragged
    X, U = synthetic_units(n, seed=41, flaps=True); X = f32_exact(X); U = f32_exact(U)
    Xd = torch.from_numpy(X).float().to(gpu); Ud = torch.from_numpy(U).float().to(gpu)
    orc = make_oracle(ac)
    idx = np.random.default_rng(2).integers(0, n, 400); idx[0], idx[1] = 0, n - 1
    out = ac.state_update(Xd, Ud, 0.01)
    assert ac.last_launch()[0].startswith("k_nn_fwd4")
    assert block_rel_err(out.cpu().numpy()[:, idx], orc.state_update(X[:, idx], U[:, idx], 0.01)) < 1e-5
    small = torch.cat([ac.state_update(Xd[:, i:i + 50000].contiguous(), Ud[:, i:i + 50000].contiguous(), 0.01)
                       for i in range(0, n, 50000)], dim=1)
    assert ac.last_launch()[0].startswith("k_nn_fwd<")
    assert torch.equal(out, small)  # same arithmetic whichever kernel evaluates a unit
    xd = ac.state_derivative(Xd, Ud).cpu().numpy()
    assert block_rel_err(xd[:, idx], orc.state_derivative(X[:, idx], U[:, idx])) < 2e-5
    ref = orc.aero(X[:, idx], U[:, idx])
    got = ac.coefficients(Xd, Ud).cpu().numpy()[:, idx]
    assert np.abs(got - ref[7:13]).max() / np.abs(ref[7:13]).max() < 2e-5
    dt = np.random.default_rng(0).uniform(1e-3, 1e-2, n)
    out2 = ac.state_update(Xd, Ud, torch.from_numpy(dt).float().to(gpu)).cpu().numpy()
    assert block_rel_err(out2[:, idx], orc.state_update(X[:, idx], U[:, idx], f32_exact(dt)[idx])) < 1e-5


@pytest.mark.parametrize("hidden", [(128, 128, 128, 128), (64, 64, 64)])
def test_rollout_one_wave_kernel_for_very_large_batches(gpu, hidden):
    """B >= 65 536 instances roll out in k_nn_rollout (one wave per 16 instances, the forward engine with its edge layers on
    the vector ALUs) instead of the cooperative kernels: a sample against the oracle, and the same instances through the
    cooperative kernel (a smaller batch) to rounding."""
    import torch
    from aircraft_amd.synthetic import near_trim_problem

    ac = make_aircraft("nn", hidden=hidden, normalise=True)
    B, H = 65536 + 48, 3
    X0s, Us = near_trim_problem(64, H, seed=9)
    reps = -(-B // 64)
    X0 = np.tile(f32_exact(X0s), (1, reps))[:, :B]; U = np.tile(f32_exact(Us), (1, 1, reps))[:, :, :B]
    X0d = torch.from_numpy(np.ascontiguousarray(X0, dtype=np.float32)).to(gpu)
    Ud = torch.from_numpy(np.ascontiguousarray(U, dtype=np.float32)).to(gpu)
    big = ac.rollout(X0d, Ud, 0.01)
    assert ac.last_launch()[0] == "k_nn_rollout"
    small = ac.rollout(X0d[:, :64].contiguous(), Ud[:, :, :64].contiguous(), 0.01)
    assert ac.last_launch()[0].startswith("k_nn_rollout_")
    want = make_oracle(ac).rollout(X0[:, :64], U[:, :, :64], 0.01)
    for sl in (slice(0, 64), slice(B - 48 - 64, B - 48)):  # first and last whole period
        assert block_rel_err(big[:, :, sl].cpu().numpy(), want) < 1e-5
    assert block_rel_err(big[:, :, :64].cpu().numpy(), small.cpu().numpy()) < 5e-6
    assert torch.equal(big[:, :, :64], big[:, :, 64 * 1000:64 * 1001])  # same inputs, same outputs, wherever they sit


@pytest.mark.parametrize("model,hidden", [("default", None), ("nn", (32, 32))])
def test_indexing_beyond_2_31_elements(gpu, model, hidden):
    """Maximum sizes: 13 M units make dF/dx a 2.2e9-element array (8.8 GB), past 32-bit element indices.  The inputs
    are a small set repeated, so every output must repeat with the same period — checked at the start, the middle and
    the very end of the arrays, and against the small call itself."""
    import torch

    free, _ = torch.cuda.mem_get_info()
    if free < 40e9:
        pytest.skip("needs ~20 GB of device memory")
    ac = make_aircraft(model, hidden=hidden, normalise=True)
    P, reps = 4096, 3175  # 13 004 800 units
    n = P * reps
    X, U = synthetic_units(P, seed=77)
    Xs = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).to(gpu)
    Us = torch.from_numpy(np.ascontiguousarray(U, dtype=np.float32)).to(gpu)
    ref = ac.step_sens(Xs, Us, 0.01)
    Xb, Ub = Xs.repeat(1, reps), Us.repeat(1, reps)
    assert Xb.shape == (13, n)
    out = ac.step_sens(Xb, Ub, 0.01)
    assert out[1].numel() > 2 ** 31
    for got, want in zip(out, ref):
        for start in (0, (reps // 2) * P, (reps - 1) * P):
            assert torch.equal(got[..., start:start + P], want), start
    # the forward step and the second-order blocks on the same batch (their own index arithmetic)
    xn = ac.state_update(Xb, Ub, 0.01)
    assert torch.equal(xn[:, -P:], ac.state_update(Xs, Us, 0.01))
    del out, xn
    torch.cuda.empty_cache()
    m = 5_200_000  # 441 x 5.2 M = 2.29e9 elements (9.2 GB)
    lam = torch.ones((13, P), device=gpu)
    Hb = ac.step_hess(Xb[:, :m].contiguous(), Ub[:, :m].contiguous(), 0.01, lam.repeat(1, m // P + 1)[:, :m].contiguous())
    assert Hb.numel() > 2 ** 31
    Hs = ac.step_hess(Xs, Us, 0.01, lam)
    last = (m // P - 1) * P
    assert torch.equal(Hb[..., last:last + P], Hs) and torch.equal(Hb[..., :P], Hs)


@pytest.mark.parametrize("hidden", [(128, 128, 128, 128), (64, 64, 64), None])
@pytest.mark.parametrize("n", [100, 16384 + 1, 16384 + 5000, 16384 + 8192, 16384 + 8193, 3 * 16384 + 2048])
def test_remainder_wave_pair_kernel_is_bit_identical(gpu, hidden, n):
    """A remainder of at most half a round of workgroups runs in k_nn_step_sens_pair (two waves per 16 units: value slab +
    tangents 0-1 | tangents 2-4, value activations and outputs exchanged through LDS).  Its results must be bit-identical to k_nn_step_sens': the same
    units computed in a batch padded to whole rounds (no remainder) give exactly the same x+, A, B, c."""
    import torch

    ac = make_aircraft("nn", hidden=hidden, normalise=True)
    X, U = synthetic_units(n, seed=n % 1000)
    Xd = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float32)).to(gpu)
    Ud = torch.from_numpy(np.ascontiguousarray(U, dtype=np.float32)).to(gpu)
    ragged = ac.step_sens(Xd, Ud, 0.01)
    if n % 16384:
        name = ac.last_launch()[0]
        # (the folded 5-32-6 net is small enough for the two-waves-per-SIMD kernel, which takes every batch size itself)
        want_name = "k_nn_step_sens_w2" if hidden is None else ("k_nn_step_sens_pair" if n < 16384 else "k_nn_step_sens")
        assert name == want_name
    n_pad = -(-n // 16384) * 16384
    reps = -(-n_pad // n)
    Xp = Xd.repeat(1, reps)[:, :n_pad].contiguous()
    Up = Ud.repeat(1, reps)[:, :n_pad].contiguous()
    whole = ac.step_sens(Xp, Up, 0.01)
    for got, want in zip(ragged, whole):
        assert torch.equal(got, want[..., :n])


@pytest.mark.parametrize("substeps", [1, 10])
def test_cfg2_full_size_mfma_off_every_unit_against_the_oracle(gpu, substeps):
    """BASELINE configs[1] at its full size: B = 256 instances x H = 50 nodes = 12 800 units, 5-64-64-64-6 surrogate,
    use_mfma = 0 (the tiled v_pk_fma_f32 engine), through MultipleShooting.linearise — EVERY unit's x+, A, B, c against the
    float64 oracle (one RK4 step per node, and the same dt in 10 sub-steps)."""
    from aircraft_amd.control import MultipleShooting

    ac = make_aircraft("nn", hidden=(64, 64, 64), use_mfma=False, substeps=substeps)
    B, H = 256, 50
    ms = MultipleShooting(system=ac, dt=0.01, num_nodes=H, opts={"quaternion": "integration"})
    Xh, Uh, X, U = _problem(B, H, gpu, seed=2)
    F, A, Bm, c = ms.linearise(X, U)
    assert ac.last_launch()[0].startswith("k_nn_step_sens_tiled")
    Xs = np.ascontiguousarray(Xh[:H].transpose(1, 0, 2).reshape(13, H * B))
    Us = np.ascontiguousarray(Uh.transpose(1, 0, 2).reshape(7, H * B))
    Xr, Ar, Br, cr = make_oracle(ac).step_sens(Xs, Us, 0.01)
    n = H * B
    Fg = F.cpu().numpy().transpose(1, 0, 2).reshape(13, n)
    e = block_rel_err(Fg, Xr)
    parity_report(f"cfg2_full[{substeps}]", block="F", units=n, block_rel_max=e)
    assert e < 1e-5
    for key, g, w in (("A", A.cpu().numpy().transpose(1, 2, 0, 3).reshape(13, 13, n), Ar),
                      ("B", Bm.cpu().numpy().transpose(1, 2, 0, 3).reshape(13, 7, n), Br),
                      ("c", c.cpu().numpy().transpose(1, 0, 2).reshape(13, n), cr)):
        eu, eb = unit_max_rel(g, w), unit_rowblock_rel(g, w)
        parity_report(f"cfg2_full[{substeps}]", block=key, units=n, unit_rel_max=float(eu.max()), rowblock_rel_max=float(eb.max()))
        assert eu.max() < 1e-5 and eb.max() < 1e-4, (key, float(eu.max()), float(eb.max()))


@pytest.mark.parametrize("model", ["cfg3_4x128", "poly"])
def test_cfg5_full_size_graph_replay_of_1000_solves(gpu, model):
    """BASELINE configs[4] at its full size: the receding-horizon closed loop at B = 1024, H = 50 (2 iLQR iterations per
    solve, overlap 30), one cycle captured into a hipGraph and replayed 1000 times — 200 s of flight per instance.  The first
    20 cycles equal the eager loop bit for bit; the executed history of all 1000 is finite and continuous on EVERY instance.
    Problem: gliders near trim under a goal-free regulator (hold heading and wings level, sink freely) — a fixed goal point
    is behind every glider after a second.  With the cubic-polynomial aerodynamics (the model of every reference driver) the
    loop settles into the steady glide; the random-weight 4x128 surrogate has no physics to settle into, but stays bounded."""
    import torch
    from aircraft_amd.control import ILQR, QuadraticCost, RecedingHorizon
    from aircraft_amd.synthetic import cruise_problem

    B, H = 1024, 50
    ac = make_aircraft("nn", hidden=(128, 128, 128, 128)) if model == "cfg3_4x128" else make_aircraft(model)
    cost = QuadraticCost.cruise()
    mk = lambda: ILQR(system=ac, dt=0.01, num_nodes=H, cost=cost, alphas=(1.0, 0.5, 0.1))  # noqa: E731
    x0 = torch.from_numpy(np.ascontiguousarray(cruise_problem(B, seed=11), dtype=np.float32)).to(gpu)
    U0 = torch.zeros((H, 7, B), device=gpu)
    eager = RecedingHorizon(mk(), overlap=30, iterations=2).allocate(x0, U0)
    he = eager.run(20, record=True)
    graph = RecedingHorizon(mk(), overlap=30, iterations=2).allocate(x0, U0).capture()
    hg = graph.run(20, record=True)
    assert he.shape == (20 * 20 + 1, 13, B)
    bits = lambda t: t.contiguous().view(torch.int32)  # noqa: E731  (bit for bit)
    assert torch.equal(bits(he), bits(hg)) and torch.equal(bits(eager.x0), bits(graph.x0)) and torch.equal(bits(eager.U), bits(graph.U))
    # ... and the remaining 980 replays, the executed states of every cycle kept on the device (1000 x 20 x 13 x 1024 fp32 = 1 GB)
    rest = graph.run(980, record=True)
    assert rest.shape == (980 * 20 + 1, 13, B) and torch.equal(bits(rest[0]), bits(hg[-1]))
    hist = torch.cat([hg, rest[1:]])
    del rest
    fin = torch.isfinite(hist).all(dim=1).all(dim=0)  # per instance
    step = (hist[1:, 0:3] - hist[:-1, 0:3]).norm(dim=1)
    speed = hist[:-1, 3:6].norm(dim=1)
    jump = float((step - speed * 0.01).abs().max())
    qerr = float((hist[:, 6:10].norm(dim=1) - 1).abs().max())
    w_end = float(hist[-1, 10:13].norm(dim=0).max())
    parity_report("cfg5_full_replay", model=model, instances=B, finite_instances=int(fin.sum()), solves=1000,
                  worst_position_jump_m=jump, worst_quaternion_norm_error=qerr, worst_final_body_rate=w_end)
    assert bool(fin.all()), int(fin.sum())
    assert jump < 0.05     # continuity: consecutive executed states are one dt apart, never a jump
    assert qerr < 1e-5
    if model == "poly":    # the real aerodynamics settle into the steady glide (every instance the same one)
        assert w_end < 0.05 and float(hist[-1, 3:6].norm(dim=0).std()) < 0.05


@pytest.mark.parametrize("model", ["poly", "default", "real"])
def test_reference_models_at_the_headline_size(gpu, model):
    """The models the reference itself ships, at cfg3's size (B = 4096 x H = 50 = 204 800 units, 3 200 workgroups whose four
    direction waves exchange primal work through LDS behind barriers): every unit of a random sample against the float64
    oracle, two runs bit-identical, halves of the batch bit-identical to the whole (other workgroups, other rounds), and the
    implicit rows (derivative kernel, same exchange) against the same oracle."""
    import torch
    from aircraft_amd.control import MultipleShooting

    ac = make_aircraft("nn" if model == "real" else model)
    ms = MultipleShooting(system=ac, dt=0.01, num_nodes=50, opts={"quaternion": "integration"})
    Xh, Uh, X, U = _problem(4096, 50, gpu, seed=7)
    F, A, Bm, c = ms.linearise(X, U)
    rng = np.random.default_rng(3)
    k = rng.integers(0, 50, 1024); b = rng.integers(0, 4096, 1024)
    k[:4] = (0, 0, 49, 49); b[:4] = (0, 4095, 0, 4095)
    Xs = np.ascontiguousarray(Xh[k, :, b].T); Us = np.ascontiguousarray(Uh[k, :, b].T)
    orc = make_oracle(ac)
    Xr, Ar, Br, cr = orc.step_sens(Xs, Us, 0.01)
    assert block_rel_err(F.cpu().numpy()[k, :, b].T, Xr) < 1e-5
    for name, got, want in (("A", A.cpu().numpy()[k, :, :, b].transpose(1, 2, 0), Ar), ("B", Bm.cpu().numpy()[k, :, :, b].transpose(1, 2, 0), Br)):
        e = unit_max_rel(got, want)
        parity_report(f"headline_size[{model}]", block=name, units=int(e.size), unit_rel_max=float(e.max()))
        assert e.max() < 1e-5, (model, name, int(e.argmax()), float(e.max()))
    F2, A2, B2, c2 = ms.linearise(X, U)
    assert torch.equal(F2, F) and torch.equal(A2, A) and torch.equal(B2, Bm)
    for lo, hi in ((0, 2048), (2048, 4096), (1000, 1100)):
        Fh, Ah, Bh, ch = ms.linearise(X[:, :, lo:hi].contiguous(), U[:, :, lo:hi].contiguous())
        assert torch.equal(Fh, F[:, :, lo:hi]) and torch.equal(Ah, A[:, :, :, lo:hi]) and torch.equal(Bh, Bm[:, :, :, lo:hi])
    if model != "real":
        # derivative kernel: f and its Jacobians at the nodes
        xd, Fx, Fu = ac.state_derivative_sens(X[:50].permute(1, 0, 2).reshape(13, -1).contiguous(), U.permute(1, 0, 2).reshape(7, -1).contiguous())
        flat = k * 4096 + b
        fr, Fxr, Fur = orc.state_derivative_sens(Xs, Us)
        assert block_rel_err(xd.cpu().numpy()[:, flat], fr) < 2e-5
        assert unit_max_rel(Fx.cpu().numpy()[:, :, flat], Fxr).max() < 1e-5
        assert unit_max_rel(Fu.cpu().numpy()[:, :, flat], Fur).max() < 1e-5
