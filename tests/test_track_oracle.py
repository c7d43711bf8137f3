"""CPU tests of the track / MHTT-progress restatement (oracle/track_oracle.py) against hand-computable cases, and of
the host-side `Track` (aircraft_amd/control/track.py) against it.  The reference holds no stored data for these
functions (and needs casadi to run), so this is where the restatement is anchored — "parity unpinned" (DESIGN.md)."""
import numpy as np
import pytest

import track_oracle as to
from aircraft_amd.control.track import Track


def straight(n=8, L=140.0, z=-200.0):
    return np.stack([np.linspace(0, L, n), np.zeros(n), np.full(n, z)], axis=1)


def arc(n=31, R=300.0, sweep=0.5, z0=-200.0, dz=4.0):
    th = np.linspace(0, sweep, n)
    return np.stack([R * np.sin(th), R * (1 - np.cos(th)), z0 + dz * th / sweep], axis=1)


def test_straight_track_known_answers():
    L = 140.0
    t = to.TrackOracle(straight(L=L))
    for s in (0.0, 0.1234, 0.5, 0.99, 1.0):
        assert np.allclose(t.eval(s), [L * s, 0, -200.0], atol=1e-10)
        assert np.allclose(t.eval_tangent(s), [L, 0, 0], atol=1e-9)
    assert np.isclose(t.length(), L, rtol=1e-12)  # 8 knots share only the end points with the 100-point grid
    # outside [0, 1]: end knot, zero tangent (initialisation.py:821-823)
    assert np.allclose(t.eval(-0.2), [0, 0, -200.0]) and np.allclose(t.eval(1.3), [L, 0, -200.0])
    assert not t.eval_tangent(1.3).any() and not t.eval_tangent(-0.2).any()


def test_closed_segments_count_an_interior_knot_twice():
    """initialisation.py:818-819 tests `s >= s0 and s <= s1` for every segment, so a point exactly on an interior
    knot is summed by both neighbours.  Restated as is (the device kernels use half-open segments instead)."""
    P = arc(n=5)
    t = to.TrackOracle(P)
    assert np.allclose(t.eval(0.5), 2 * P[2])
    assert np.allclose(t.eval(np.nextafter(0.5, 1.0)), P[2], atol=1e-9)
    assert np.allclose(t.eval(0.0), P[0]) and np.allclose(t.eval(1.0), P[-1])


def test_hermite_interpolates_knots_and_slopes():
    P = arc(n=11)
    t = to.TrackOracle(P)
    eps = 1e-9
    for i in range(1, 10):
        s = i / 10 + eps
        assert np.allclose(t.eval(s), P[i], atol=1e-5)
        central = (P[i + 1] - P[i - 1]) / 0.2  # mean of the two secant slopes on a uniform grid
        assert np.allclose(t.eval_tangent(s), central, rtol=1e-6, atol=1e-5)
    # arc length of a circular arc: R * sweep (plus the small climb)
    assert abs(to.TrackOracle(arc(n=31)).length() - np.hypot(300 * 0.5, 4.0)) < 0.05


def test_progress_recursions_on_a_straight_track():
    L, dt, H = 140.0, 0.01, 6
    t = to.TrackOracle(straight(L=L))
    X = np.zeros((H + 1, 13, 2)); X[:, 2] = -200.0
    X[:, 3, 0] = 50.0; X[:, 0, 0] = 50.0 * dt * np.arange(H + 1)            # on the track, along it
    X[:, 3, 1] = -20.0; X[:, 4, 1] = 30.0; X[:, 0, 1] = 70.0; X[:, 1, 1] = 3.0  # backwards, 3 m to the side
    s0 = np.array([0.0, 0.5 + 1e-3])
    S0 = to.progress_initial(t, L, X, s0, dt)
    assert np.allclose(S0[:, 0], 50.0 * dt * np.arange(H + 1) / L)
    assert np.allclose(S0[:, 1], s0[1] - 20.0 * dt * np.arange(H + 1) / L)
    S1, sd, te = to.progress_tight(t, L, X, s0, dt)
    assert np.allclose(sd[:, 0], 50.0 / L) and np.allclose(sd[:, 1], -20.0 / L)
    assert np.allclose(te[:, 0], 0.0, atol=1e-18)
    assert np.allclose(S1[:, 0], S0[:, 0])  # on the track the position correction vanishes
    # instance 1: pos_err = (70 - L s, 3, 0); correction 0.05 (70 - L s) / L pulls s towards x / L = 0.5
    s = s0[1]
    for k in range(H):
        assert np.isclose(te[k, 1], (70.0 - L * s) ** 2 + 9.0)
        s = s - 20.0 * dt / L + 0.05 * (70.0 - L * s) / L
        assert np.isclose(S1[k + 1, 1], s)
    # the [0, 1] box
    S = to.progress_initial(t, L, X[:, :, :1], np.array([0.999]), dt)
    assert S[-1, 0] == 1.0


def test_mhtt_loss_hand_computed():
    L, dt, H = 100.0, 0.1, 2
    t = to.TrackOracle(straight(n=6, L=L, z=0.0))
    X = np.zeros((H + 1, 13, 1)); U = np.zeros((H, 7, 1))
    X[:, 0, 0] = [0, 4, 8]; X[:, 1, 0] = [1, 0, 2]; X[:, 3, 0] = [40, 40, 0.05]
    U[1, :3, 0] = [1, -2, 0.5]; U[0, 0, 0] = 9.0  # u_0 is pinned in the NLP and not in the loss
    S = np.array([[0.0], [0.04], [0.07]])
    # node1 <- terms at node0: err = 0^2 + 1^2, s_dot = 0.4 ; node2 <- node1: err = (4-4)^2 + 0, s_dot = 0.4
    tracking = 1.0 + 0.0
    progress = 0.04 + 0.07
    rate = 0.4 + 0.4
    slow = (0.1 - 0.05) ** 2  # node 2 only
    effort = 1 + 4 + 0.25      # u_1
    terminal = np.hypot(8 - 100, 2)
    want = 10 * tracking - 5 * progress - 2 * rate + 10 * slow + 20 * terminal + 100 * effort
    assert np.isclose(to.mhtt_loss(t, L, X, U, S)[0], want, rtol=1e-13)
    X[:, 3, 0] = [-30, 40, 0.05]  # flying backwards at node 0
    want += -2 * (-0.3 - 0.4) + 50 * 0.3 ** 2
    assert np.isclose(to.mhtt_loss(t, L, X, U, S)[0], want, rtol=1e-13)


def test_host_track_equals_restatement():
    rng = np.random.default_rng(1)
    P = arc(n=17) + rng.normal(0, 0.3, (17, 3))
    t, T = to.TrackOracle(P), Track(P)
    ss = list(rng.uniform(-0.1, 1.1, 60)) + [0.0, 1.0, 0.5, 0.25, 1 / 16, np.nextafter(0.5, 0)]
    for s in ss:
        assert np.allclose(T.eval(s), t.eval(s), rtol=0, atol=1e-12)
        assert np.allclose(T.eval_tangent(s), t.eval_tangent(s), rtol=1e-12, atol=1e-9)
    assert np.isclose(T.length(), t.length(), rtol=1e-13)
    # vectorised call and the per-segment cubics handed to the device
    sv = np.array(ss[:20])
    assert np.allclose(T.eval(sv), np.stack([t.eval(s) for s in sv], axis=1), atol=1e-12)
    c = T.segment_cubics().astype(np.float64)
    for s in rng.uniform(0, 1, 20):
        i = min(int(s * 16), 15); u = s * 16 - i
        assert np.allclose(c[i] @ [1, u, u * u, u ** 3], t.eval(s), rtol=1e-6, atol=1e-4)
    with pytest.raises(ValueError):
        Track(np.zeros((1, 3)))


def test_node_cost_restatement_reduces_to_the_constant_cost():
    import ilqr_oracle as io
    from aircraft_amd.control import QuadraticCost

    rng = np.random.default_rng(2)
    H, B = 5, 3
    c = QuadraticCost(q=list(rng.uniform(0, 1, 13)), qf=list(rng.uniform(0, 2, 13)), r=[0.3] * 7,
                      x_ref=list(rng.normal(size=13)), x_goal=list(rng.normal(size=13)), reg=0.1)
    X, U = rng.normal(size=(H + 1, 13, B)), rng.normal(size=(H, 7, B))
    A, Bm = rng.normal(size=(H, 13, 13, B)) * 0.3, rng.normal(size=(H, 13, 7, B))
    nq = np.tile(np.asarray(c.q)[None, :, None], (H + 1, 1, B)); nq[H] = np.asarray(c.qf)[:, None]
    nx = np.tile(np.asarray(c.x_ref)[None, :, None], (H + 1, 1, B)); nx[H] = np.asarray(c.x_goal)[:, None]
    ng = np.zeros_like(nq)
    assert np.allclose(io.cost(c, X, U, node=(nq, nx, ng)), io.cost(c, X, U))
    for a, b in zip(io.backward(c, X, U, A, Bm, node=(nq, nx, ng)), io.backward(c, X, U, A, Bm)):
        assert np.allclose(a, b, rtol=1e-10, atol=1e-12)
    # a linear term shifts the gradient only
    ng = rng.normal(size=nq.shape)
    assert np.allclose(io.cost(c, X, U, node=(nq, nx, ng)) - io.cost(c, X, U), (ng * X).sum(axis=(0, 1)))


def test_reference_dubins_geometry():
    """The track of the reference's OWN problem: tests/golden/dubins_track.npz is the sampled 3-D Dubins path that
    DubinsInitialiser builds for data/glider/problem_definition.json (make_fixtures.py::dubins_track runs the reference's
    aircraft.dubins for it).  This pins the INPUT of the track functions to the reference's geometry; their CasADi evaluation
    cannot run here, so the restatement stays the checker for eval / eval_tangent / length."""
    from tests.helpers import golden

    g = golden("dubins_track.npz")
    P, conf = g["points"], g["configurations"]
    assert P.shape == (201, 3)
    # the path starts at the initial position, ends at the last waypoint and passes through the inner ones
    assert np.allclose(P[0], g["initial_state"][:3]) and np.allclose(P[-1], conf[-1, :3], atol=1e-9)
    for wp in conf[1:-1, :3]:
        assert np.linalg.norm(P - wp, axis=1).min() < 1e-9
    # sampled chord length ~ the manoeuvre lengths the reference's Dubins code reports (50 + 76 + 75 samples; each segment's
    # sample count max(50, int(length / 2)))
    chord = np.linalg.norm(np.diff(P, axis=0), axis=1).sum()
    assert abs(chord - g["segment_lengths"].sum()) / g["segment_lengths"].sum() < 2e-2
    t, T = to.TrackOracle(P), Track(P)
    rng = np.random.default_rng(3)
    exact_knots = [0.5, 0.25, 0.75, 0.125, 0.875]   # k / 200 that are binary fractions: hit exactly, counted twice
    for s in list(rng.uniform(-0.05, 1.05, 80)) + exact_knots + [0.0, 1.0, 0.005, np.nextafter(0.5, 1)]:
        assert np.allclose(T.eval(s), t.eval(s), rtol=0, atol=1e-10)
        assert np.allclose(T.eval_tangent(s), t.eval_tangent(s), rtol=1e-12, atol=1e-7)
    k = 100  # s = 0.5 is knot 100: the closed segments [99, 100] and [100, 101] both hold it
    assert np.allclose(T.eval(0.5), 2 * P[k], atol=1e-10) and np.allclose(T.eval(np.nextafter(0.5, 1)), P[k], atol=1e-6)
    assert np.isclose(T.length(), t.length(), rtol=1e-12)
    assert 0.9 * g["segment_lengths"].sum() < T.length() < 1.2 * g["segment_lengths"].sum()
