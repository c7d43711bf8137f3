"""GPU tests of the track / progress terms of the moving-horizon track tracker against the float64
restatements in oracle/track_oracle.py and oracle/ilqr_oracle.py."""
import numpy as np
import pytest

from tests.helpers import block_rel_err, f32_exact, make_aircraft, make_oracle, rel_fro

pytestmark = pytest.mark.gpu


def dev(a, gpu):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(gpu)


def f64(t):
    return t.cpu().numpy().astype(np.float64)


def arc_points(n=31, R=300.0, sweep=0.5, z0=-200.0, dz=4.0):
    th = np.linspace(0, sweep, n)
    return np.stack([R * np.sin(th), R * (1 - np.cos(th)), z0 + dz * th / sweep], axis=1)


def setup(gpu, B=48, H=20, model="poly", hidden=None, seed=5):
    """B gliders released near trim at the start of a gently curving track, a few metres off it."""
    import track_oracle as to
    from aircraft_amd.control import MHTT, Track
    from aircraft_amd.synthetic import quat_from_euler, quat_rotate

    ac = make_aircraft(model, hidden=hidden)
    P = arc_points()
    track = Track(P)
    mh = MHTT(system=ac, track=track, dt=0.01, num_nodes=H, alphas=(1.0, 0.5, 0.1))
    rng = np.random.default_rng(seed)
    X0 = np.zeros((13, B))
    X0[0] = rng.uniform(-1, 3, B); X0[1] = rng.uniform(-3, 3, B); X0[2] = -200.0 + rng.uniform(-2, 2, B)
    V = rng.uniform(50, 60, B); al = np.deg2rad(rng.uniform(-1, 1, B)); be = np.deg2rad(rng.uniform(-1, 1, B))
    vb = np.stack([V * np.cos(al) * np.cos(be), V * np.sin(be), V * np.sin(al) * np.cos(be)])
    q = quat_from_euler(np.deg2rad(rng.uniform(-5, 5, B)), np.deg2rad(rng.uniform(-2, 2, B)), np.deg2rad(rng.uniform(-5, 5, B)))
    X0[3:6] = quat_rotate(q, vb); X0[6:10] = q; X0[10:13] = rng.normal(0, 0.02, (3, B))
    U = np.zeros((H, 7, B)); U[:, :3] = rng.normal(0, 0.3, (1, 3, B))
    s0 = rng.uniform(0.0, 0.02, B)
    return ac, mh, to.TrackOracle(P), f32_exact(X0), f32_exact(U), f32_exact(s0)


def test_track_eval_matches_host_track(gpu):
    ac, mh, tro, *_ = setup(gpu, B=4, H=4)
    rng = np.random.default_rng(0)
    s = f32_exact(np.concatenate([rng.uniform(0, 1, 500), [0.0, 1.0, -0.3, 1.5]]))
    pos, tan = mh.track_eval(dev(s, gpu))
    want_p, want_t = mh.track.eval(s), mh.track.eval_tangent(s)
    # fp32 knots: a draw that lands on an interior knot would be double-counted by the host (closed segments) — none does
    assert not np.isin(s[:500], mh.track.s_vals).any()
    assert np.abs(f64(pos) - want_p).max() < 2e-4          # metres, on coordinates up to ~200
    assert np.abs(f64(tan) - want_t).max() / np.abs(want_t).max() < 2e-5
    assert not f64(tan)[:, -2:].any() and np.allclose(f64(pos)[:, -1], mh.track.points[-1], atol=1e-4)


def test_reference_dubins_track_on_the_device(gpu):
    """The reference's own geometry (tests/golden/dubins_track.npz: the sampled Dubins path of problem_definition.json) on the
    device: ac_track_eval_f32 against the line-by-line restatement, INCLUDING progress values exactly on a knot, where the
    reference's closed segment intervals return the sum of both one-sided values (twice the knot), and the progress recursion
    and loss of gliders flying along it."""
    import track_oracle as to
    from tests.helpers import golden
    from aircraft_amd.control import MHTT, Track

    g = golden("dubins_track.npz")
    P = g["points"]
    ac = make_aircraft("poly")
    track = Track(P)
    H = 20
    mh = MHTT(system=ac, track=track, dt=0.01, num_nodes=H, alphas=(1.0, 0.5, 0.1))
    tro = to.TrackOracle(P)
    rng = np.random.default_rng(0)
    knots = np.array([0.5, 0.25, 0.75, 0.125, 0.375, 0.625, 0.875])  # k / 200 representable in fp32: hit exactly
    near = np.float32(1 / 200) * np.arange(1, 12, dtype=np.float32)   # fp32 products next to knots that are NOT fp32 numbers
    s = f32_exact(np.concatenate([rng.uniform(0, 1, 400), knots, near, [0.0, 1.0, -0.2, 1.3]]))
    pos, tan = mh.track_eval(dev(s, gpu))
    want_p = np.stack([tro.eval(float(v)) for v in s], axis=1)
    want_t = np.stack([tro.eval_tangent(float(v)) for v in s], axis=1)
    assert np.abs(f64(pos) - want_p).max() < 5e-4            # metres, on coordinates up to ~400 (twice a knot)
    assert np.abs(f64(tan) - want_t).max() / np.abs(want_t).max() < 5e-5
    assert np.allclose(f64(pos)[:, 400], 2 * P[100], atol=5e-4)  # s = 0.5 = knot 100, counted by both segments
    # gliders released on the first straight of the path, flying along it (the path climbs at 45 degrees there: they do not
    # follow it for long, which is immaterial to the recursion being checked)
    B = 32
    X0 = np.zeros((13, B))
    s_start = rng.uniform(0.001, 0.01, B)
    X0[0:3] = np.stack([tro.eval(float(v)) for v in s_start], axis=1) + rng.normal(0, 0.5, (3, B))
    th = np.stack([tro.eval_tangent(float(v)) for v in s_start], axis=1); th /= np.linalg.norm(th, axis=0)
    from aircraft_amd.synthetic import quat_from_euler
    X0[3:6] = 50.0 * th
    X0[6:10] = quat_from_euler(np.zeros(B), -np.arcsin(th[2]), np.arctan2(th[1], th[0]))  # nose along the velocity
    U = np.zeros((H, 7, B))
    X0, s0 = f32_exact(X0), f32_exact(s_start)
    X = mh.rollout(dev(X0, gpu), dev(U, gpu))
    Xh, L = f64(X), mh.track_length
    assert np.isfinite(Xh).all()
    for mode in (0, 1):
        S = mh.progress(X, dev(s0, gpu), mode=mode)
        want = to.progress_initial(tro, L, Xh, s0, mh.dt) if mode == 0 else to.progress_tight(tro, L, Xh, s0, mh.dt)[0]
        assert np.abs(f64(S) - want).max() < 2e-6
    S1 = mh.progress(X, dev(s0, gpu), mode=1)
    J = mh.loss(X, dev(U, gpu), S1)
    wantJ = to.mhtt_loss(tro, L, Xh, U, f64(S1))
    assert np.abs(f64(J) - wantJ).max() / np.abs(wantJ).max() < 2e-5


@pytest.mark.parametrize("mode", [0, 1])
def test_progress_recursion_matches_restatement(gpu, mode):
    import track_oracle as to

    ac, mh, tro, X0, U, s0 = setup(gpu)
    X = mh.rollout(dev(X0, gpu), dev(U, gpu))
    S, sd, te = mh.progress(X, dev(s0, gpu), mode=mode, want_terms=True)
    Xh = f64(X)
    L = mh.track_length
    if mode == 0:
        want = to.progress_initial(tro, L, Xh, s0, mh.dt)
    else:
        want, wsd, wte = to.progress_tight(tro, L, Xh, s0, mh.dt)
        assert np.abs(f64(sd) - wsd).max() / np.abs(wsd).max() < 1e-5
        assert np.abs(f64(te) - wte).max() / np.abs(wte).max() < 1e-4
    assert np.abs(f64(S) - want).max() < 2e-6  # progress is O(0.1); 20 chained fp32 updates
    assert (np.diff(f64(S), axis=0) > 0).all()  # these gliders fly along the track


def test_mhtt_loss_matches_restatement(gpu):
    import track_oracle as to

    ac, mh, tro, X0, U, s0 = setup(gpu)
    X = mh.rollout(dev(X0, gpu), dev(U, gpu))
    S = mh.progress(X, dev(s0, gpu), mode=1)
    J = mh.loss(X, dev(U, gpu), S)
    want = to.mhtt_loss(tro, mh.track_length, f64(X), U, f64(S))
    assert np.abs(f64(J) - want).max() / np.abs(want).max() < 2e-5
    # non-default weights reach the kernel
    from aircraft_amd.control import MHTTWeights
    mh.weights = MHTTWeights(w_tracking=1.0, w_progress=0.0, w_progress_rate=0.0, w_backward=0.0, w_terminal_align=0.0,
                             w_low_velocity=0.0, w_control=0.0)
    _, _, te = mh.progress(X, dev(s0, gpu), mode=1, want_terms=True)
    assert np.allclose(f64(mh.loss(X, dev(U, gpu), S)), f64(te).sum(axis=0), rtol=1e-5)


def test_initialise_matches_reference_shape(gpu):
    """MHTT.initialise(initial_state, current_progress) -> (13 + 7 + 1, N+1): rollout of zero controls, progress by
    velocity projection (moving_horizon.py:203-239)."""
    import track_oracle as to

    ac, mh, tro, X0, U, s0 = setup(gpu, H=25)
    guess = mh.initialise(X0[:, 3], 0.013)
    assert guess.shape == (21, 26) and not guess[13:20].any() and guess[20, 0] == np.float32(0.013)
    orc = make_oracle(ac)
    want = orc.rollout(X0[:, 3:4], np.zeros((25, 7, 1)), 0.01)
    assert block_rel_err(guess[:13].T[:, :, None], want) < 2e-6
    S = to.progress_initial(tro, mh.track_length, want, np.array([np.float32(0.013)]), 0.01)
    assert np.abs(guess[20] - S[:, 0]).max() < 2e-6


def test_node_cost_backward_and_cost_match_numpy(gpu):
    import ilqr_oracle as io

    ac, mh, tro, X0, U, s0 = setup(gpu, B=16, H=12)
    Ud = dev(U, gpu)
    mh.set_progress(s0)
    X = mh.rollout(dev(X0, gpu), Ud)
    F, A, Bm, _ = mh.linearise(X, Ud, want_c=False)
    K, kff, dV = mh.backward(X, Ud, A, Bm)
    ws = mh._mhtt_workspace(16, Ud.device)
    node = tuple(f64(ws[k]) for k in ("nq", "nx", "ng"))
    Kr, kr, dVr = io.backward(mh.cost, f64(X), U, f64(A), f64(Bm), node=node)
    assert rel_fro(f64(K), Kr) < 2e-3 and rel_fro(f64(kff), kr) < 2e-3 and rel_fro(f64(dV), dVr) < 2e-3
    assert (f64(dV)[0] <= 0).all()
    # the model: tracking curvature on position, reference = track point at the frozen progress
    S = f64(ws["S"])
    assert np.allclose(node[0][:-1, :3], 2 * mh.weights.w_tracking) and not node[0][:, 3:].any()
    want_ref = np.stack([mh.track.eval(S[k]) for k in range(12)])
    assert np.abs(node[1][:12, :3] - want_ref).max() < 1e-3
    assert (node[2][:12, 3] < 0).all()  # rewards velocity along the track (+x here)
    # cost of the quadratic model through the ABI, also on a line-search-wide batch (column a*B + b -> b)
    import ctypes as C
    import torch
    from aircraft_amd import _lib
    Xc = torch.cat([X, X + 0.01], dim=2).contiguous(); Uc = torch.cat([Ud, Ud], dim=2).contiguous()
    out = torch.empty(32, device=X.device)
    _lib.check(_lib.load().ac_ilqr_cost_node_f32(ac._handle, C.byref(mh.cost.struct()), ws["nq"].data_ptr(),
                                                 ws["nx"].data_ptr(), ws["ng"].data_ptr(), 16, Xc.data_ptr(),
                                                 Uc.data_ptr(), 32, 12, out.data_ptr(), ac._stream()), "cost_node")
    want = io.cost(mh.cost, f64(Xc), f64(Uc), node=node)
    assert np.abs(f64(out) - want).max() / np.abs(want).max() < 1e-5


@pytest.mark.parametrize("model,hidden,B", [("poly", None, 96), ("nn", (64, 64, 64), 48)])
def test_mhtt_solve_improves_true_loss(gpu, model, hidden, B):
    ac, mh, tro, X0, U, s0 = setup(gpu, B=B, H=40, model=model, hidden=hidden)
    X, Uo, S, hist = mh.solve(dev(X0, gpu), dev(s0, gpu), dev(np.zeros_like(U), gpu), iters=6)
    h = f64(hist)
    assert np.isfinite(h).all()
    assert (np.diff(h, axis=0) <= 1e-6 * np.abs(h[:-1]) + 1e-5).all()  # accepted on the true loss: monotone
    assert (h[-1] < h[0]).mean() > 0.9
    # the accepted pair is dynamically consistent and S is its progress
    Xchk = mh.rollout(dev(X0, gpu), Uo)
    assert block_rel_err(f64(X), f64(Xchk)) < 5e-5
    assert np.abs(f64(S) - f64(mh.progress(Xchk, dev(s0, gpu), mode=1))).max() < 1e-5
    # tracking error at the end of the horizon shrinks for most instances
    _, _, te0 = mh.progress(mh.rollout(dev(X0, gpu), dev(np.zeros_like(U), gpu)), dev(s0, gpu), mode=1, want_terms=True)
    _, _, te1 = mh.progress(X, dev(s0, gpu), mode=1, want_terms=True)
    assert (f64(te1)[-1] < f64(te0)[-1]).mean() > 0.7


def test_mhtt_with_exact_dynamics_hessian(gpu):
    """The Newton variant of the sweep under the per-node MHTT cost model: the costate uses the node arrays, the
    second-order dynamics blocks enter the Riccati pass, acceptance stays on the true loss (monotone)."""
    from aircraft_amd.control import MHTT

    ac, mh, tro, X0, U, s0 = setup(gpu, B=48, H=30)
    newton = MHTT(system=ac, track=mh.track, dt=0.01, num_nodes=30, alphas=(1.0, 0.5, 0.1), hessian="exact")
    X, Uo, S, hist = newton.solve(dev(X0, gpu), dev(s0, gpu), dev(np.zeros_like(U), gpu), iters=4)
    h = f64(hist)
    assert np.isfinite(h).all() and (np.diff(h, axis=0) <= 1e-6 * np.abs(h[:-1]) + 1e-5).all()
    assert (h[-1] < h[0]).mean() > 0.8
    _, _, _, hist_gn = mh.solve(dev(X0, gpu), dev(s0, gpu), dev(np.zeros_like(U), gpu), iters=4)
    assert np.median(np.abs(h[-1] - f64(hist_gn)[-1]) / np.abs(f64(hist_gn)[-1])) < 0.05  # same minimum


def test_receding_horizon_on_the_track_eager_equals_graph(gpu):
    """The closed loop of main/mhe/mhtt.py:79-124 with the track: solve, keep N - overlap nodes, restart from the last
    kept state AND its progress.  Captured into a hipGraph the loop must reproduce the eager run."""
    from aircraft_amd.control import RecedingHorizon

    ac, mh, tro, X0, U, s0 = setup(gpu, B=64, H=30)
    U0 = dev(np.zeros_like(U), gpu)
    mh.set_progress(s0)
    eager = RecedingHorizon(mh, overlap=18, iterations=2).allocate(dev(X0, gpu), U0)
    he = eager.run(5, record=True)
    s_eager = mh.s0.clone()
    mh.set_progress(s0)
    graph = RecedingHorizon(mh, overlap=18, iterations=2).allocate(dev(X0, gpu), U0).capture()
    hg = graph.run(5, record=True)
    assert he.shape == (5 * 12 + 1, 13, 64)
    assert block_rel_err(f64(hg), f64(he)) < 1e-6
    assert np.abs(f64(mh.s0) - f64(s_eager)).max() < 1e-6
    assert (f64(mh.s0) > s0 + 0.15).all()  # 60 executed steps at ~55 m/s on a 150 m track
