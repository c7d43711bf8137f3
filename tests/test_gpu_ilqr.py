"""GPU tests of the batched iLQR sweep against the NumPy float64 restatement in oracle/ilqr_oracle.py."""
import numpy as np
import pytest

from tests.helpers import block_rel_err, f32_exact, make_aircraft, make_oracle, rel_fro, synthetic_units

pytestmark = pytest.mark.gpu


def dev(a, gpu):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(gpu)


def setup(gpu, model="poly", hidden=None, B=24, H=30, use_mfma=True):
    """B gliders released near trim from the same point, flying roughly along +x; the goal is a point a little to the
    side of where they would coast to, so the optimal controls are small and the closed loop stays in the envelope."""
    from aircraft_amd.control import ILQR, QuadraticCost
    from aircraft_amd.synthetic import quat_from_euler, quat_rotate

    ac = make_aircraft(model, hidden=hidden, use_mfma=use_mfma)
    T = H * 0.01
    cost = QuadraticCost.goal((60.0 * T, 0.5), w_goal=1.0, height=-200.0, w_height=1.0, w_lateral_speed=0.5, r=0.5, reg=1.0)
    cost.q = [0, 0, 1e-2, 0, 0, 0, 0, 0, 0, 0, 0.2, 0.2, 0.2]
    cost.x_ref = [0, 0, -200.0] + [0] * 10
    il = ILQR(system=ac, dt=0.01, num_nodes=H, cost=cost, alphas=(1.0, 0.5, 0.1))
    rng = np.random.default_rng(3)
    X0 = np.zeros((13, B))
    X0[2] = -200.0
    V = rng.uniform(50, 65, B); al = np.deg2rad(rng.uniform(-1, 1, B)); be = np.deg2rad(rng.uniform(-1, 1, B))
    vb = np.stack([V * np.cos(al) * np.cos(be), V * np.sin(be), V * np.sin(al) * np.cos(be)])
    q = quat_from_euler(np.deg2rad(rng.uniform(-5, 5, B)), np.deg2rad(rng.uniform(-2, 2, B)), np.deg2rad(rng.uniform(-5, 5, B)))
    X0[3:6] = quat_rotate(q, vb); X0[6:10] = q; X0[10:13] = rng.normal(0, 0.02, (3, B))
    U = np.zeros((H, 7, B)); U[:, :3] = rng.normal(0, 0.2, (1, 3, B))
    return ac, il, cost, f32_exact(X0), f32_exact(U)


@pytest.mark.parametrize("model,hidden", [("poly", None), ("nn", (64, 64, 64))])
def test_backward_pass_matches_numpy(gpu, model, hidden):
    import ilqr_oracle as io

    ac, il, cost, X0, U = setup(gpu, model, hidden)
    X = il.rollout(dev(X0, gpu), dev(U, gpu))
    F, A, Bm, _ = il.linearise(X, dev(U, gpu), want_c=False)
    K, kff, dV = il.backward(X, dev(U, gpu), A, Bm)
    Xh = X.cpu().numpy().astype(np.float64); Ah = A.cpu().numpy().astype(np.float64); Bh = Bm.cpu().numpy().astype(np.float64)
    Kr, kr, dVr = io.backward(cost, Xh, U, Ah, Bh)  # same A, B: isolates the Riccati arithmetic
    assert rel_fro(K.cpu().numpy(), Kr) < 2e-3
    assert rel_fro(kff.cpu().numpy(), kr) < 2e-3
    assert rel_fro(dV.cpu().numpy(), dVr) < 2e-3
    assert (dV.cpu().numpy()[0] <= 0).all()  # descent direction


@pytest.mark.parametrize("model,hidden,use_mfma", [("poly", None, True), ("nn", (64, 64, 64), True),
                                                   ("nn", (64, 64, 64), False), ("nn", (32, 32), False)])
def test_policy_rollout_and_cost_match_numpy(gpu, model, hidden, use_mfma):
    import ilqr_oracle as io

    ac, il, cost, X0, U = setup(gpu, model, hidden, B=20, H=25, use_mfma=use_mfma)
    Ud = dev(U, gpu)
    X = il.rollout(dev(X0, gpu), Ud)
    F, A, Bm, _ = il.linearise(X, Ud, want_c=False)
    K, kff, dV = il.backward(X, Ud, A, Bm)
    Xc, Uc = il.forward(dev(X0, gpu), X, Ud, K, kff)
    f64 = lambda t: t.cpu().numpy().astype(np.float64)  # noqa: E731
    Xr, Ur = io.forward(make_oracle(ac), cost, X0, f64(X), U, f64(K), f64(kff), il.alphas, 0.01)
    assert np.isfinite(Xr).all() and np.isfinite(Xc.cpu().numpy()).all()
    assert np.abs(Uc.cpu().numpy() - Ur).max() < 2e-4  # degrees; gains of O(10) times 1e-6 state differences
    assert block_rel_err(Xc.cpu().numpy(), Xr) < 1e-5
    Jc = il.trajectory_cost(Xc, Uc).cpu().numpy()
    assert np.abs(Jc - io.cost(cost, f64(Xc), f64(Uc))).max() / np.abs(Jc).max() < 1e-5
    # alpha = 0 reproduces the nominal trajectory (within the control box)
    X0c, U0c = il.forward(dev(X0, gpu), X, Ud, K, kff, alphas=[0.0])
    assert block_rel_err(X0c.cpu().numpy(), X.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("model,hidden,B,use_mfma", [("poly", None, 256, True), ("nn", (128, 128, 128, 128), 128, True),
                                                     ("nn", (64, 64, 64), 300, False)])
def test_ilqr_cost_decreases(gpu, model, hidden, B, use_mfma):
    import torch

    ac, il, cost, X0, U = setup(gpu, model, hidden, B=B, H=50, use_mfma=use_mfma)
    X, Uo, hist = il.solve(dev(X0, gpu), dev(np.zeros_like(U), gpu), iters=6)
    h = hist.cpu().numpy()
    assert np.isfinite(h).all()
    assert (np.diff(h, axis=0) <= 1e-6 * np.abs(h[:-1]) + 1e-6).all()  # monotone per instance
    assert (h[-1] < h[0]).mean() > 0.9 and np.median(h[-1] / h[0]) < 0.8  # nearly all improve, typically by > 20 %
    # the accepted (X, U) pair is dynamically consistent: X is the rollout of U from x0
    Xchk = il.rollout(dev(X0, gpu), Uo)
    assert block_rel_err(X.cpu().numpy(), Xchk.cpu().numpy()) < 5e-5
    lim = torch.tensor(cost.u_max, device=Uo.device)[None, :, None]
    assert bool((Uo <= lim + 1e-6).all()) and bool((Uo >= torch.tensor(cost.u_min, device=Uo.device)[None, :, None] - 1e-6).all())


def test_solve_writes_reference_trajectory_file(gpu, tmp_path):
    """ILQR.solve(save_to=...) leaves one `iteration_k` group per iteration in the reference's HDF5 layout
    (control/base.py:89-105); the last group is the returned trajectory of the chosen instance."""
    from aircraft_amd import trajectory_io as tio

    if not tio.hdf5_available():
        pytest.skip("no HDF5 C library in this image")
    ac, il, cost, X0, U = setup(gpu, "poly", None, B=32, H=20)
    path = str(tmp_path / "solve.h5")
    X, Uo, hist = il.solve(dev(X0, gpu), dev(np.zeros_like(U), gpu), iters=3, save_to=path, save_instance=5)
    assert tio.list_iterations(path) == [0, 1, 2, 3]
    last = tio.load_trajectory(path, 3)
    assert last.state.shape == (13, 21) and last.control.shape == (7, 20) and last.times.shape == (21,)
    assert np.array_equal(last.state, X[:, :, 5].T.cpu().numpy().astype(np.float64))
    assert np.array_equal(last.control, Uo[:, :, 5].T.cpu().numpy().astype(np.float64))
    assert np.allclose(last.times, il.dt * np.arange(21))
    first = tio.load_trajectory(path, 0)
    assert np.array_equal(first.state[:, 0], last.state[:, 0]) and not first.control.any()


def test_receding_horizon_loop_eager_equals_graph(gpu):
    """The closed loop of main/mhe/mhtt.py:79-124 (solve, keep N-overlap nodes, restart from the last kept state):
    a hipGraph replay of one cycle reproduces the eager loop bit for bit, and the executed trajectory is continuous."""
    import torch
    from aircraft_amd.control import RecedingHorizon

    ac, il, cost, X0, U = setup(gpu, "nn", (64, 64, 64), B=64, H=50)
    x0, U0 = dev(X0, gpu), dev(np.zeros_like(U), gpu)
    eager = RecedingHorizon(il, overlap=30, iterations=2).allocate(x0, U0)
    he = eager.run(4, record=True)
    graph = RecedingHorizon(il, overlap=30, iterations=2).allocate(x0, U0).capture()
    hg = graph.run(4, record=True)
    assert he.shape == (4 * 20 + 1, 13, 64)
    assert torch.equal(he, hg) and torch.equal(eager.x0, graph.x0) and torch.equal(eager.U, graph.U)
    assert torch.isfinite(he).all()
    # continuity: consecutive executed states are one dt apart (positions move by ~v dt, never jump)
    step = (he[1:, 0:3] - he[:-1, 0:3]).norm(dim=1)
    speed = he[:-1, 3:6].norm(dim=1)
    assert float((step - speed * 0.01).abs().max()) < 0.05


def test_costate_and_newton_backward_match_numpy(gpu):
    """Exact-Hessian sweep: costate recursion, then the backward pass with the second-order blocks of
    ac_shoot_hess_f32 added to Qxx / Qux / Quu, against the NumPy restatement fed the same A, B, Hz."""
    import ilqr_oracle as io

    ac, il, cost, X0, U = setup(gpu, "poly", None, B=12, H=16)
    Ud = dev(U, gpu)
    X = il.rollout(dev(X0, gpu), Ud)
    F, A, Bm, _ = il.linearise(X, Ud, want_c=False)
    Lam = il.costate(X, A)
    f64 = lambda t: t.cpu().numpy().astype(np.float64)  # noqa: E731
    assert rel_fro(f64(Lam), io.costate(cost, f64(X), f64(A))) < 1e-5
    Hz = il.hessian(X, Ud, Lam)
    K, kff, dV = il.backward(X, Ud, A, Bm, Hz=Hz)
    Kr, kr, dVr = io.backward(cost, f64(X), U, f64(A), f64(Bm), Hz=f64(Hz))
    assert rel_fro(f64(K), Kr) < 2e-3 and rel_fro(f64(kff), kr) < 2e-3 and rel_fro(f64(dV), dVr) < 2e-3
    # the second-order terms matter here: the Gauss-Newton gains differ
    K0, _, _ = il.backward(X, Ud, A, Bm)
    assert rel_fro(f64(K), f64(K0)) > 1e-3


def test_exact_hessian_solve_reaches_the_same_minimum(gpu):
    from aircraft_amd.control import ILQR

    ac, il, cost, X0, U = setup(gpu, "poly", None, B=128, H=40)
    newton = ILQR(system=ac, dt=0.01, num_nodes=40, cost=cost, alphas=(1.0, 0.5, 0.1), hessian="exact")
    U0 = dev(np.zeros_like(U), gpu)
    _, _, h_gn = il.solve(dev(X0, gpu), U0, iters=8)
    Xn, Un, h_nt = newton.solve(dev(X0, gpu), U0, iters=8)
    h_gn, h_nt = h_gn.cpu().numpy(), h_nt.cpu().numpy()
    assert np.isfinite(h_nt).all()
    assert (np.diff(h_nt, axis=0) <= 1e-6 * np.abs(h_nt[:-1]) + 1e-6).all()  # accepted steps only: monotone
    assert (h_nt[-1] < h_nt[0]).mean() > 0.9
    # same problem, same start, same line search: both sweeps settle in the same minimum (the second-order terms change
    # the path, not the destination)
    assert (np.abs(h_nt[-1] - h_gn[-1]) <= 0.02 * np.abs(h_gn[-1]) + 1e-3).mean() > 0.9
    assert block_rel_err(Xn.cpu().numpy(), newton.rollout(dev(X0, gpu), Un).cpu().numpy()) < 5e-5
    # the MLP surrogate takes the same path (stage tensors from the MFMA engine's second-order mode)
    acn, iln, costn, X0n, Un_ = setup(gpu, "nn", (32, 32), B=16, H=10)
    nn_newton = ILQR(system=acn, dt=0.01, num_nodes=10, cost=costn, alphas=(1.0, 0.5, 0.1), hessian="exact")
    _, _, hn = nn_newton.solve(dev(X0n, gpu), dev(np.zeros_like(Un_), gpu), iters=3)
    hn = hn.cpu().numpy()
    assert np.isfinite(hn).all() and (np.diff(hn, axis=0) <= 1e-6 * np.abs(hn[:-1]) + 1e-6).all()


@pytest.mark.parametrize("model,hidden", [("poly", None), ("nn", (32, 32))])
def test_exact_hessian_loop_is_graph_capturable(gpu, model, hidden):
    """The Newton variant inside the receding-horizon loop: costate + second-order blocks (memset + kernels; for the MLP
    the stage-tensor workspace is sized by the warm-up cycle) captured into a hipGraph reproduce the eager run."""
    from aircraft_amd.control import ILQR, RecedingHorizon

    ac, il, cost, X0, U = setup(gpu, model, hidden, B=64, H=20)
    newton = ILQR(system=ac, dt=0.01, num_nodes=20, cost=cost, alphas=(1.0, 0.5, 0.1), hessian="exact")
    U0 = dev(np.zeros_like(U), gpu)
    he = RecedingHorizon(newton, overlap=12, iterations=2).allocate(dev(X0, gpu), U0).run(4, record=True)
    hg = RecedingHorizon(newton, overlap=12, iterations=2).allocate(dev(X0, gpu), U0).capture().run(4, record=True)
    assert he.shape == (4 * 8 + 1, 13, 64)
    assert block_rel_err(hg.cpu().numpy(), he.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("warm_start", ["shift", "zero"])
@pytest.mark.parametrize("iterations", [0, 2])
def test_receding_horizon_loop_matches_numpy_restatement(gpu, warm_start, iterations):
    """a19: the closed loop against oracle/receding_oracle.py (a float64 restatement of main/mhe/mhtt.py:79-124 around
    the NumPy iLQR sweep and the C++ oracle's dynamics): which nodes are kept, which state restarts the next solve, how the
    controls are shifted / zeroed, what the re-rollout produces.  iterations = 0 isolates the loop itself."""
    import receding_oracle as ro
    from aircraft_amd.control import RecedingHorizon

    B, H, overlap, cycles = 6, 20, 12, 3
    ac, il, cost, X0, U = setup(gpu, "poly", None, B=B, H=H)
    U0 = U if iterations == 0 else np.zeros_like(U)  # a non-trivial control sequence when nothing re-solves it
    loop = RecedingHorizon(il, overlap=overlap, iterations=iterations, warm_start=warm_start).allocate(dev(X0, gpu), dev(U0, gpu))
    hist = loop.run(cycles, record=True).cpu().numpy().astype(np.float64)
    want, x0_w, U_w, choices = ro.receding_horizon(make_oracle(ac), cost, X0, U0, overlap, iterations, cycles, il.alphas, 0.01,
                                                   warm_start=warm_start)
    keep = H - overlap
    assert hist.shape == want.shape == (cycles * keep + 1, 13, B)
    assert np.array_equal(hist[0], X0.astype(np.float32))
    assert block_rel_err(hist, want) < (1e-5 if iterations == 0 else 5e-5), block_rel_err(hist, want)
    assert block_rel_err(loop.x0.cpu().numpy()[None], x0_w[None]) < (1e-5 if iterations == 0 else 5e-5)
    # controls carried into the next cycle (degrees): shifted tail / zeros
    assert np.abs(loop.U.cpu().numpy() - U_w).max() < (1e-6 if iterations == 0 else 2e-3)
    if warm_start == "zero":
        assert not loop.U.cpu().numpy().any()
    if iterations:
        assert all((c >= 0).any() for ch in choices for c in ch)  # the solver did move the iterate in every cycle


def _envelope_numpy(orc, X, lo, hi, w):
    """penalty cost (B,), gradient (H+1, 13, B) and Gauss-Newton curvature (H+1, 13, 13, B) from the oracle's rows / Jx"""
    Hn, _, B = X.shape
    cost = np.zeros(B); grad = np.zeros((Hn, 13, B)); curv = np.zeros((Hn, 13, 13, B))
    for k in range(Hn):
        rows, Jx = orc.envelope(X[k])
        viol = np.where(rows > hi[:, None], rows - hi[:, None], np.where(rows < lo[:, None], rows - lo[:, None], 0.0))
        cost += w * (viol ** 2).sum(axis=0)
        grad[k] = 2 * w * np.einsum("rb,rjb->jb", viol, Jx)
        act = (viol != 0).astype(float)
        curv[k] = 2 * w * np.einsum("rb,rib,rjb->ijb", act, Jx, Jx)
    return cost, grad, curv


def test_envelope_penalty_kernels_match_numpy(gpu):
    """ac_envelope_cost_f32 / ac_envelope_model_f32 — the soft form of AircraftControl.state_constraint
    (control/aircraft.py:44-59) the batched sweep uses — against the oracle's envelope rows and their exact Jacobian."""
    import torch
    from aircraft_amd.control import ILQR
    from tests.helpers import synthetic_problem

    ac, il0, cost, X0, U = setup(gpu, "poly", None, B=40, H=12)
    il = ILQR(system=ac, dt=0.01, num_nodes=12, cost=cost, alphas=(1.0, 0.5), envelope_weight=3.0,
              envelope_bounds=((45.0 ** 2, 60.0 ** 2), (-0.01, 0.01), (-0.02, 0.03), (-1e30, -199.5)))  # tight: many rows active
    Xs, Us = synthetic_problem(40, 12, seed=3)
    X = il.rollout(dev(X0, gpu), dev(0.3 * Us, gpu))
    Xh = X.cpu().numpy().astype(np.float64)
    lo = np.array([b[0] for b in il.envelope_bounds]); hi = np.array([b[1] for b in il.envelope_bounds])
    cw, gw, pw = _envelope_numpy(make_oracle(ac), Xh, lo, hi, 3.0)
    assert (cw > 0).mean() > 0.5  # the test is not vacuous
    J = torch.full((40,), 7.0, device=gpu)
    il.envelope_cost(X, J)
    assert np.abs(J.cpu().numpy() - 7.0 - cw).max() <= 2e-5 * max(cw.max(), 1.0)
    glin = torch.zeros((13, 13, 40), device=gpu); Hz = torch.zeros((12, 21, 21, 40), device=gpu)
    il._envelope_model(X, glin=glin, Hz=Hz)
    assert np.abs(glin.cpu().numpy() - gw).max() <= 5e-5 * max(np.abs(gw).max(), 1.0)
    Hh = Hz.cpu().numpy()
    assert np.abs(Hh[:, :13, :13] - pw[:12]).max() <= 1e-4 * max(np.abs(pw).max(), 1.0)
    assert not Hh[:, 13:].any() and not Hh[:, :, 13:].any()


def test_envelope_al_kernels_match_numpy(gpu):
    """The augmented-Lagrangian form of the envelope rows (ac_envelope_al_cost_f32 / _model_f32 / _update_f32: multipliers per
    node, row and instance) against the NumPy restatement (oracle/ilqr_oracle.py::envelope_al) on the oracle's rows and exact
    Jacobians: cost with random multipliers (also for a batch of line-search candidates that shares them), gradient,
    Gauss-Newton curvature, and the first-order multiplier update."""
    import torch
    import ilqr_oracle as io
    from aircraft_amd.control import ILQR
    from tests.helpers import synthetic_problem

    B, H, w = 40, 12, 3.0
    ac, il0, cost, X0, U = setup(gpu, "poly", None, B=B, H=H)
    il = ILQR(system=ac, dt=0.01, num_nodes=H, cost=cost, alphas=(1.0, 0.5), envelope_weight=w, envelope="al",
              envelope_bounds=((45.0 ** 2, 60.0 ** 2), (-0.01, 0.01), (-0.02, 0.03), (-1e30, -199.5)))
    Xs, Us = synthetic_problem(B, H, seed=3)
    X = il.rollout(dev(X0, gpu), dev(0.3 * Us, gpu))
    Xh = X.cpu().numpy().astype(np.float64)
    lo = np.array([b[0] for b in il.envelope_bounds]); hi = np.array([b[1] for b in il.envelope_bounds])
    ws = il._workspace(B, X.device)
    rng = np.random.default_rng(8)
    lam0 = f32_exact(rng.uniform(0, 1, (H + 1, 8, B)) * (rng.uniform(0, 1, (H + 1, 8, B)) < 0.5)
                     * np.array([50.0, 0.2, 0.05, 2.0] * 2)[None, :, None])
    lam0[:, 7] = 0.0  # (z has no lower bound: -1e30)
    ws["lam"].copy_(dev(lam0, gpu))
    orc = make_oracle(ac)
    cw, gw, pw, sv, rows = io.envelope_al(orc, Xh, lo, hi, w, lam0)
    assert (sv != 0).mean() > 0.15  # not vacuous: many shifted violations are active
    # the beta row's shifts lam / 2w reach 0.033 > its half-width 0.01: at some nodes BOTH shifted bounds are violated, and
    # the two max() terms of L_A must then both count (cost, gradient and curvature)
    both = (rows[:, 1] - hi[1] + lam0[:, 1] / (2 * w) > 0) & (lo[1] - rows[:, 1] + lam0[:, 5] / (2 * w) > 0)
    assert both.mean() > 0.02
    J = torch.full((B,), 7.0, device=gpu)
    il.envelope_cost(X, J)
    assert np.abs(J.cpu().numpy() - 7.0 - cw).max() <= 2e-5 * max(np.abs(cw).max(), 1.0)
    # two line-search candidates per instance, column a * B + b, sharing instance b's multipliers
    X2 = torch.cat([X, X], dim=2).contiguous()
    J2 = torch.zeros((2 * B,), device=gpu)
    il.envelope_cost(X2, J2)
    assert np.abs(J2.cpu().numpy() - np.tile(cw, 2)).max() <= 2e-5 * max(np.abs(cw).max(), 1.0)
    glin = torch.zeros((H + 1, 13, B), device=gpu); Hz = torch.zeros((H, 21, 21, B), device=gpu)
    il._envelope_model(X, glin=glin, Hz=Hz)
    assert np.abs(glin.cpu().numpy() - gw).max() <= 5e-5 * max(np.abs(gw).max(), 1.0)
    assert np.abs(Hz.cpu().numpy()[:, :13, :13] - pw[:H]).max() <= 1e-4 * max(np.abs(pw).max(), 1.0)
    viol = il.update_multipliers(X).cpu().numpy()
    want = io.envelope_al_update(rows, lo, hi, w, lam0)
    got = ws["lam"].cpu().numpy()
    assert (got >= 0).all() and np.abs(got - want).max() <= 2e-5 * max(np.abs(want).max(), 1.0)
    span = np.where((hi - lo) < 1e30, hi - lo, 1.0)
    exc = (np.maximum(rows - hi[None, :, None], lo[None, :, None] - rows) / span[None, :, None]).max(axis=(0, 1))
    assert np.abs(viol - np.maximum(exc, 0)).max() <= 2e-5 * max(exc.max(), 1.0)
    # zero multipliers: the plain penalty, bit for bit
    ws["lam"].zero_()
    Ja, Jp = torch.zeros((B,), device=gpu), torch.zeros((B,), device=gpu)
    il.envelope_cost(X, Ja)
    ILQR(system=ac, dt=0.01, num_nodes=H, cost=cost, alphas=(1.0, 0.5), envelope_weight=w,
         envelope_bounds=il.envelope_bounds).envelope_cost(X, Jp)
    assert torch.equal(Ja, Jp)


def test_envelope_multipliers_enforce_what_the_penalty_leaves_violated(gpu):
    """The hard treatment of the envelope rows (the reference enforces them as NLP constraints, control/aircraft.py:44-59):
    gliders asked to descend to a goal height below a floor they must not cross (the height row: z <= -198.5 m; the goal
    is at -185 m).  A quadratic penalty of moderate weight converges to a trajectory that crosses the floor by about
    multiplier / (2 weight) = 0.38 m; augmented-Lagrangian multipliers of the SAME weight, updated every third sweep, bring
    the crossing down to centimetres (measured: median 0.02 m, worst instance 0.09 m after eleven updates, still falling)."""
    from aircraft_amd.control import ILQR, QuadraticCost
    from tests.helpers import parity_report

    ac, il0, cost, X0, U = setup(gpu, "poly", None, B=48, H=40)
    cost = QuadraticCost.goal((24.0, 0.0), w_goal=1.0, height=-185.0, w_height=40.0, w_lateral_speed=0.1, r=0.02, reg=1.0)
    floor = -198.5
    big = np.deg2rad(20)
    bounds = ((20.0 ** 2, 100.0 ** 2), (-np.deg2rad(10), np.deg2rad(10)), (-big, big), (-1e30, floor))
    res = {}
    for mode in ("penalty", "al"):
        il = ILQR(system=ac, dt=0.01, num_nodes=40, cost=cost, alphas=(1.0, 0.5, 0.25, 0.1, 0.03), envelope_weight=500.0,
                  envelope_bounds=bounds, envelope=mode)
        X, Uo, hist = il.solve(dev(X0, gpu), dev(np.zeros_like(U), gpu), iters=36, al_every=3)
        rows, _ = il.envelope(X)
        res[mode] = (rows[:, 3].amax(dim=0) - floor).cpu().numpy()   # how far below the floor (z is down), metres
        assert np.isfinite(hist.cpu().numpy()).all()
        if mode == "al":
            lam = il._ws["lam"].cpu().numpy()
            assert (lam >= 0).all() and (lam[:, 3] > 0).any() and not lam[:, 7].any()  # the floor's multipliers are the active ones
    parity_report("envelope_al_vs_penalty", penalty_excess_median_m=float(np.median(res["penalty"])),
                  al_excess_median_m=float(np.median(res["al"])), al_excess_max_m=float(res["al"].max()))
    assert np.median(res["penalty"]) > 0.3                       # the penalty alone settles 0.38 m below the floor
    assert np.median(res["al"]) < 0.06 and res["al"].max() < 0.2   # the multipliers enforce it


def test_envelope_penalty_steers_the_solve(gpu):
    """With the envelope as a soft constraint the solve trades goal cost for staying inside: gliders asked to reach a goal
    far below their glide path exceed the alpha bound without the penalty and stay (nearly) inside with it."""
    from aircraft_amd.control import ILQR, QuadraticCost

    ac, il0, cost, X0, U = setup(gpu, "poly", None, B=48, H=40)
    cost = QuadraticCost.goal((24.0, 0.0), w_goal=1.0, height=-185.0, w_height=40.0, w_lateral_speed=0.1, r=0.02, reg=1.0)
    bounds = ((20.0 ** 2, 100.0 ** 2), (-np.deg2rad(10), np.deg2rad(10)), (-np.deg2rad(4), np.deg2rad(4)), (-1e30, 0.0))
    res = {}
    for w in (0.0, 2e6):
        il = ILQR(system=ac, dt=0.01, num_nodes=40, cost=cost, alphas=(1.0, 0.5, 0.25, 0.1), envelope_weight=w,
                  envelope_bounds=bounds)
        X, Uo, hist = il.solve(dev(X0, gpu), dev(np.zeros_like(U), gpu), iters=8)
        rows, _ = il.envelope(X)
        alpha = rows[:, 2].abs().amax(dim=0).cpu().numpy()
        h = hist.cpu().numpy()
        assert np.isfinite(h).all() and (np.diff(h, axis=0) <= 1e-5 * np.abs(h[:-1]) + 1e-5).all()  # monotone incl. the penalty
        res[w] = alpha
    lim = np.deg2rad(4)
    assert (res[0.0] > 1.3 * lim).mean() > 0.5          # unconstrained: most instances leave the alpha bound by > 30 %
    # penalised (quadratic penalty, weight 2e6: measured median 1.06 x the bound, 0.126 / 0.088 rad at 2e4 / 2e5)
    assert (res[2e6] < 1.15 * lim).mean() > 0.9
    assert np.median(res[2e6]) < 0.5 * np.median(res[0.0])


@pytest.mark.parametrize("model,hidden", [("poly", None), ("nn", (64, 64, 64))])
def test_time_as_a_decision_variable_kernels_match_numpy(gpu, model, hidden):
    """The reference carries dt_k per node as a decision variable (control/base.py:276, 339-385: dt_k = 1/progress_k^2 or
    progress_k^2, bounded by dt_bounds) and charges the total time (main/control/control.py:44, 66-67).  Here a control row the
    force model ignores carries dt_k: linearisation at per-node steps with c = dF/d(dt) as that row's column of B, the linear
    time cost in the backward pass and the cost kernel, and the policy rollout that integrates node k with its own clipped
    step — against the NumPy restatement (which takes node k's step from the same row)."""
    import torch
    import ilqr_oracle as io
    from aircraft_amd.control import ILQR

    ac, il0, cost, X0, U = setup(gpu, model, hidden, B=20, H=25)
    il = ILQR(system=ac, dt=0.01, num_nodes=25, cost=cost, alphas=(1.0, 0.5, 0.1), time="variable",
              dt_bounds=(0.006, 0.015), w_time=300.0, r_time=50.0)
    row = il.time_row
    assert row == 3 and il.cost.dt_row == 3 and il.cost.u_lin[3] == 300.0
    rng = np.random.default_rng(4)
    U = U.copy(); U[:, row] = f32_exact(rng.uniform(0.007, 0.014, (25, 20)))   # a different step at every node
    Ud = dev(U, gpu)
    # the nominal trajectory at those steps: the policy rollout with zero gains
    z = lambda *s: torch.zeros(s, device=gpu)  # noqa: E731
    X, U1 = il.forward(dev(X0, gpu), z(26, 13, 20), Ud, z(25, 7, 13, 20), z(25, 7, 20), alphas=[0.0])
    assert torch.equal(U1, Ud)
    f64 = lambda t: t.cpu().numpy().astype(np.float64)  # noqa: E731
    orc = make_oracle(ac)
    Xr = np.zeros((26, 13, 20)); Xr[0] = X0
    for k in range(25):
        Xr[k + 1] = orc.state_update(Xr[k], U[k], U[k, row])
    assert block_rel_err(f64(X), Xr) < 1e-5
    # linearisation at the nodes' own steps; the time row's column of B is c
    ws = il._workspace(20, X.device)
    ws["dt"].copy_(Ud[:, row, :])
    F, A, Bm, c = il.linearise(X, Ud, dt=ws["dt"], want_c=True)
    flatX = np.ascontiguousarray(f64(X)[:25].transpose(1, 0, 2).reshape(13, -1)); flatU = np.ascontiguousarray(U.transpose(1, 0, 2).reshape(7, -1))
    _, Ar, Br, cr = orc.step_sens(flatX, flatU, np.ascontiguousarray(U[:, row].reshape(-1)))
    assert rel_fro(f64(c).transpose(1, 0, 2).reshape(13, -1), cr) < 1e-5 and not f64(Bm)[:, :, row].any()
    Bm[:, :, row, :].copy_(c)
    K, kff, dV = il.backward(X, Ud, A, Bm)
    Kr, kr, dVr = io.backward(il.cost, f64(X), U, f64(A), f64(Bm))
    assert rel_fro(f64(K), Kr) < 2e-3 and rel_fro(f64(kff), kr) < 2e-3 and rel_fro(f64(dV), dVr) < 2e-3
    assert np.abs(f64(K)[:, row]).max() > 0 and np.abs(f64(kff)[:, row]).max() > 0     # the time row takes part in the policy
    Xc, Uc = il.forward(dev(X0, gpu), X, Ud, K, kff)
    Xcr, Ucr = io.forward(orc, il.cost, X0, f64(X), U, f64(K), f64(kff), il.alphas, 0.01)
    assert np.abs(f64(Uc)[:, :3] - Ucr[:, :3]).max() < 2e-4 and np.abs(f64(Uc)[:, row] - Ucr[:, row]).max() < 2e-7
    assert (f64(Uc)[:, row] >= 0.006 - 1e-9).all() and (f64(Uc)[:, row] <= 0.015 + 1e-9).all()   # dt_bounds are a hard box
    assert block_rel_err(f64(Xc), Xcr) < 1e-5
    Jc = il.trajectory_cost(Xc, Uc).cpu().numpy()
    assert np.abs(Jc - io.cost(il.cost, f64(Xc), f64(Uc))).max() / np.abs(Jc).max() < 1e-5


def test_time_as_a_decision_variable_reaches_a_goal_the_fixed_step_cannot(gpu):
    """Gliders at 50-65 m/s, 40 nodes, a goal 30 m ahead: at the fixed step 0.01 s the horizon covers 20-26 m, so the goal term
    stays large; with dt_k free in [0.005, 0.02] (and the total time charged) the solve stretches the steps until the horizon
    reaches the goal — the column c = dF/d(dt) of the sensitivity kernels is what tells it how."""
    from aircraft_amd.control import ILQR, QuadraticCost

    ac, il0, cost, X0, U = setup(gpu, "poly", None, B=48, H=40)
    cost = QuadraticCost.goal((30.0, 0.0), w_goal=1.0, height=-200.0, w_height=1.0, w_lateral_speed=0.1, r=0.5, reg=1.0)
    out = {}
    for mode in ("fixed", "variable"):
        il = ILQR(system=ac, dt=0.01, num_nodes=40, cost=cost, alphas=(1.0, 0.5, 0.25, 0.1), time=mode,
                  dt_bounds=(0.005, 0.02), w_time=5.0, r_time=0.0)
        X, Uo, hist = il.solve(dev(X0, gpu), dev(np.zeros_like(U), gpu), iters=12)
        h = hist.cpu().numpy()
        assert np.isfinite(h).all() and (np.diff(h, axis=0) <= 1e-5 * np.abs(h[:-1]) + 1e-5).all()
        miss = (X[-1, 0] - 30.0).abs().cpu().numpy()
        T = Uo[:, 3].sum(dim=0).cpu().numpy() if mode == "variable" else np.full(48, 0.4)
        out[mode] = (miss, T, Uo[:, 3].cpu().numpy())
    from tests.helpers import parity_report
    parity_report("time_variable_vs_fixed", miss_fixed_median_m=float(np.median(out["fixed"][0])),
                  miss_variable_median_m=float(np.median(out["variable"][0])), total_time_median_s=float(np.median(out["variable"][1])))
    assert np.median(out["fixed"][0]) > 4.0                                  # 30 m is out of reach in 0.4 s
    assert np.median(out["variable"][0]) < 0.25 * np.median(out["fixed"][0])  # the free steps close most of the gap
    assert (out["variable"][1] > 0.42).mean() > 0.9                          # by taking longer ...
    dtv = out["variable"][2]
    assert dtv.min() >= 0.005 - 1e-9 and dtv.max() <= 0.02 + 1e-9            # ... inside dt_bounds


def test_handle_released_inside_a_plain_capture_does_not_invalidate_it(gpu):
    """The last reference to an `Aircraft` dropped, and the cyclic collector run, INSIDE a plain torch.cuda.graph capture
    (no quiet_capture): ac_destroy's hipFree would invalidate the capture; SixDOF.close() parks the handle instead and the
    next call outside the capture destroys it.  The graph replays."""
    import gc
    import torch
    from aircraft_amd.dynamics import base as dyn

    victim = make_aircraft("default")
    keeper = make_aircraft("default", normalise=True)
    X, U = synthetic_units(64, seed=5)
    Xd, Ud = dev(X, gpu), dev(U, gpu)
    victim.state_update(Xd, Ud, 0.01)          # both handles exist on the device
    out = keeper.state_update(Xd, Ud, 0.01)
    ref = out.clone()
    cyc = [victim]; cyc.append(cyc)            # only a cyclic collection can release it
    del victim
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    parked_before = len(dyn._PARKED)
    with torch.cuda.graph(g, stream=side):
        del cyc
        gc.collect()                           # finalises the victim while the stream is capturing
        assert len(dyn._PARKED) >= parked_before + 1   # (the collection may also reach handles of earlier tests)
        lib = keeper._sync()
        from aircraft_amd import _lib as L
        L.check(lib.ac_step_f32(keeper._handle, Xd.data_ptr(), Ud.data_ptr(), 0.01, None, 64, out.data_ptr(),
                                keeper._stream()), "ac_step_f32")
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    keeper.state_update(Xd, Ud, 0.01)          # first call outside the capture drains the parked handle
    assert len(dyn._PARKED) == 0


def _goal_problem(gpu, B=10, H=14, model="poly"):
    from aircraft_amd.control import GoalAcquisition
    from tests.helpers import near_trim_problem

    ac = make_aircraft(model, normalise=True)
    X0, U = near_trim_problem(B, H, seed=6)
    rng = np.random.default_rng(12)
    U = f32_exact(U + rng.normal(0, 0.1, U.shape) * (np.arange(7) < 3)[None, :, None])
    goal = f32_exact(np.stack([rng.uniform(5, 9, B), rng.uniform(-1, 1, B)]))
    il = GoalAcquisition(system=ac, goal=goal, dt=0.01, num_nodes=H, vx_max=58.0, w_al=4.0, alphas=(1.0, 0.5, 0.1), reg=1.0)
    return ac, il, f32_exact(X0), U, goal


def test_goal_acquisition_kernels_match_numpy(gpu):
    """The reference's goal-acquisition loss (Controller.loss, main/control/control.py:44-68) on the device against its
    NumPy restatement (oracle/ilqr_oracle.py: goal_cost / goal_model / goal_multiplier, written from the reference's
    formulas): exact value for an iterate and for a batch of line-search candidates that shares goals and multipliers; the
    quadratic model (node arrays, control gradient of the rate term, its curvature on the (u,u) diagonal of Hz); the
    multiplier update; and the backward pass fed that model against the NumPy Riccati pass."""
    import torch
    import ilqr_oracle as io

    ac, il, X0, U, goal = _goal_problem(gpu)
    B, H = U.shape[2], U.shape[0]
    orc = make_oracle(ac)
    Ud = dev(U, gpu)
    X = il.rollout(dev(X0, gpu), Ud)
    Xh = X.cpu().numpy().astype(np.float64)
    g = il._goal_ws(B, gpu)
    lam = f32_exact(np.random.default_rng(1).uniform(0, 40, B) * (np.arange(B) % 2))
    g["lam"].copy_(dev(lam, gpu))
    gl = io.GoalLoss(w_al=4.0, vx_max=58.0)
    want = io.goal_cost(orc, gl, goal, Xh, U, lam)
    got = il.trajectory_cost(X, Ud).cpu().numpy()
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()
    act = Xh[-1, 3] - 58.0 + lam / 8.0 > 0
    assert act.any() and (~act).any()      # the inequality is active for some instances only
    X2, U2 = torch.cat([X, X], dim=2).contiguous(), torch.cat([Ud, 0.5 * Ud], dim=2).contiguous()
    got2 = il.trajectory_cost(X2, U2).cpu().numpy()
    want2 = np.concatenate([want, io.goal_cost(orc, gl, goal, Xh, 0.5 * U, lam)])
    assert np.abs(got2 - want2).max() <= 2e-5 * np.abs(want2).max()
    ws = il._workspace(B, gpu)
    Hz = ws["Hz"].zero_()
    ug = il._goal_model(X, Ud, Hz)
    nq, nx, ng, ugw, uhw = io.goal_model(orc, gl, goal, Xh, U, lam)
    f64 = lambda t: t.cpu().numpy().astype(np.float64)  # noqa: E731
    assert np.abs(f64(g["nq"]) - nq).max() <= 1e-6 * np.abs(nq).max()
    assert np.abs(f64(g["nx"]) - nx).max() <= 1e-6 * max(np.abs(nx).max(), 1.0)
    assert np.abs(f64(g["ng"]) - ng).max() <= 2e-5 * np.abs(ng).max()
    assert np.abs(f64(ug) - ugw).max() <= 2e-5 * max(np.abs(ugw).max(), 1.0)
    Hh = f64(Hz)
    diag = np.stack([Hh[:, 13 + i, 13 + i] for i in range(7)], axis=1)
    assert np.abs(diag - uhw).max() <= 2e-5 * np.abs(uhw).max()
    Hh[:, np.arange(13, 20), np.arange(13, 20)] = 0.0
    assert not Hh.any()                    # nothing but the (u,u) diagonal is touched
    # backward pass with the control gradient
    F, A, Bm, _ = il.linearise(X, Ud, want_c=False)
    node = (g["nq"], g["nx"], g["ng"])
    K, kff, dV = il.backward(X, Ud, A, Bm, Hz=Hz, node=node, uglin=ug)
    Hzw = np.zeros((H, 21, 21, B)); Hzw[:, np.arange(13, 20), np.arange(13, 20)] = uhw
    Kr, kr, dVr = io.backward(il.cost, Xh, U, f64(A), f64(Bm), node=(nq, nx, ng), Hz=Hzw, uglin=ugw)
    assert rel_fro(f64(K), Kr) < 2e-3 and rel_fro(f64(kff), kr) < 2e-3 and rel_fro(f64(dV), dVr) < 2e-3
    K0, k0, _ = il.backward(X, Ud, A, Bm, Hz=Hz, node=node)
    assert rel_fro(f64(kff), f64(k0)) > 1e-3   # the control gradient matters
    viol = il.update_goal_multiplier(X).cpu().numpy()
    lw, vw = io.goal_multiplier(gl, Xh, lam)
    assert np.abs(g["lam"].cpu().numpy() - lw).max() <= 1e-5 * max(lw.max(), 1.0) and np.abs(viol - vw).max() <= 1e-5 * max(vw.max(), 1.0)


def test_goal_acquisition_sweep_decreases_the_reference_loss(gpu):
    """A few sweeps on the reference's goal-acquisition loss: the exact loss of every instance never increases within a
    block of sweeps (the line search accepts only improvements), it decreases for most, and the solve stays finite."""
    import torch

    ac, il, X0, U, goal = _goal_problem(gpu, B=24, H=30)
    X, Uo, hist = il.solve(dev(X0, gpu), dev(U, gpu), iters=6)
    h = hist.cpu().numpy()
    assert np.isfinite(h).all() and torch.isfinite(X).all()
    assert (np.diff(h, axis=0) <= 1e-3 * np.abs(h[:-1])).all()
    assert (h[-1] < h[0] - 1e-3 * np.abs(h[0])).mean() > 0.8
