"""CPU tests: pin the float64 oracle to the reference's own artefacts (no GPU needed).

  * simulation.h5 replay  — the reference's stored rollout (data/trajectories/simulation.h5, written by
    main/dynamics/dynamics.py:134-145): poly model, dt = 0.1, 10 sub-steps, no normalisation, CoM override.
  * ScaledModel golden vectors — produced by importing the reference's own torch module
    (tests/golden/make_fixtures.py); forward values and autograd Jacobians.
  * known answers inside the source (DefaultModel constants) and AD-vs-finite-difference checks for the
    sensitivities, which nothing in the reference pins.
"""
import numpy as np
import pytest

from tests.helpers import (GLIDER, block_rel_err, golden, make_aircraft, make_oracle, near_trim_problem,
                           synthetic_units)


def glider_oracle(model="default", **kw):
    return make_oracle(make_aircraft(model, **kw))


def test_simulation_h5_replay_one_step_from_every_column():
    sim = golden("simulation_h5.npz")
    S, U = sim["state"], sim["control"]
    o = glider_oracle("poly", substeps=10, normalise=False)
    X1 = o.state_update(S[:, :-1], U[:, :-1], 0.1)
    assert block_rel_err(X1, S[:, 1:]) < 1e-12
    # components that are not pure cancellation noise agree to the last bits
    big = np.abs(S[:, 1:]) > 1e-3
    assert (np.abs(X1 - S[:, 1:])[big] / np.abs(S[:, 1:])[big]).max() < 1e-13


def test_simulation_h5_replay_chained():
    sim = golden("simulation_h5.npz")
    S, U = sim["state"], sim["control"]
    o = glider_oracle("poly", substeps=10, normalise=False)
    traj = o.rollout(S[:, :1], np.ascontiguousarray(U[:, :-1].T[:, :, None]), 0.1)[:, :, 0].T
    assert traj.shape == S.shape
    assert block_rel_err(traj, S) < 1e-10
    # SURVEY.md App. B known answers
    assert np.allclose(S[:, 1][[0, 2, 3, 5, 7, 9, 11]],
                       [4.9819627850119135, -199.9356772780644, 49.37371693561093, 2.469763608014319,
                        -0.0863806521559003, 0.9962622977615254, -1.9891161610023425], rtol=0, atol=0)


def test_inverse_is_conjugate_over_norm_squared():
    """The fixture discriminates liecasadi's inverse() = conj/|q|^2 from a bare conjugate: with an
    un-normalised quaternion the two differ, and only the former reproduces the stored rollout."""
    sim = golden("simulation_h5.npz")
    S, U = sim["state"], sim["control"]
    n2 = (S[6:10] ** 2).sum(axis=0)
    assert np.abs(n2 - 1).max() > 1e-8  # the stored quaternions do drift off the unit sphere


def test_scaledmodel_golden_vectors():
    g = golden("scaledmodel_golden.npz")
    o = glider_oracle("nn")
    y, J = o.mlp(g["x"].astype(np.float64))
    assert np.abs(y - g["y_f64"]).max() < 1e-13
    assert np.abs(J - g["jac_f64"]).max() < 1e-12
    # the reference runs the net in fp32 inside l4casadi; the float64 restatement stays within fp32 rounding of it
    assert np.abs(y - g["y_f32"]).max() < 1e-6
    assert np.abs(J - g["jac_f32"]).max() < 5e-6
    # SURVEY.md §8c known answer
    assert np.allclose(g["y_f32"][0], [-0.049651101, 0.006531812, -0.522380352, 0.025876714, -0.249959484,
                                       -0.000759662], atol=2e-8)


def test_default_model_known_answers():
    """DefaultModel constants (coefficient_models.py:41-78) at a hand-computable point."""
    o = glider_oracle("default")
    x = np.zeros((13, 1)); x[3] = 50.0; x[9] = 1.0; x[10:13, 0] = [0.1, -0.2, 0.3]
    u = np.zeros((7, 1)); u[:3, 0] = [2.0, -3.0, 1.0]; u[6] = 0.5
    a = o.aero(x, u)
    eps = 1e-6
    vr = np.array([50 + eps, eps, eps])
    alpha = np.arctan2(vr[2], vr[0] + eps); beta = np.arcsin(vr[1] / np.sqrt(vr @ vr + eps))
    d = np.pi / 180
    C = [-(0.02 + 0.3 * alpha ** 2) - 0.1 * 0.5, -0.98 * beta, -5 * alpha - 0.6 * 0.5,
         0.08 * 4 * 2.0 * d - 0.05 * 0.1, -1.2 * 5 * -3.0 * d - 0.5 * -0.2, -0.1 * 6 * 1.0 * d - 0.05 * 0.3]
    assert np.allclose(a[7:13, 0], C, rtol=1e-14, atol=1e-16)
    qbar = 0.5 * 1.225 * (vr @ vr)
    assert np.isclose(a[6, 0], qbar, rtol=1e-15)
    assert np.allclose(a[13:16, 0], np.array(C[:3]) * qbar * GLIDER["reference_area"], rtol=1e-14)


def test_euler_getters_known_answers():
    """phi/theta/psi (base.py:179-195) invert the ZYX Euler -> quaternion map used by scipy's Rotation
    (utils.py imports it for the initial attitude)."""
    from scipy.spatial.transform import Rotation

    rng = np.random.default_rng(4)
    ang = np.stack([rng.uniform(-3, 3, 50), rng.uniform(-1.4, 1.4, 50), rng.uniform(-3, 3, 50)])  # psi, theta, phi
    q = Rotation.from_euler("ZYX", ang.T).as_quat()  # xyzw
    X = np.zeros((13, 50)); X[6:10] = q.T; X[3] = 30.0
    a = glider_oracle("default").aero(X, np.zeros((7, 50)))
    assert np.abs(a[19] - ang[2]).max() < 1e-12 and np.abs(a[20] - ang[1]).max() < 1e-12
    assert np.abs(a[21] - ang[0]).max() < 1e-12


def test_reference_test_invariants():
    """The intent of the reference's stale unit tests (src/aircraft/tests/test_dynamics.py:44-76)."""
    o = glider_oracle("default", normalise=True)
    X, U = synthetic_units(64, seed=2)
    # omega = 0  =>  q_dot = 0
    X0 = X.copy(); X0[10:13] = 0
    assert np.abs(o.state_derivative(X0, U)[6:10]).max() == 0.0
    # |q| stays 1 after a normalised step with omega != 0
    assert np.abs(np.linalg.norm(o.state_update(X, U, 0.01)[6:10], axis=0) - 1).max() < 1e-15
    # identity attitude => v_frd_rel == v_ned + eps
    Xi = X.copy(); Xi[6:10] = np.array([0, 0, 0, 1.0])[:, None]
    assert np.abs(o.aero(Xi, U)[0:3] - (Xi[3:6] + 1e-6)).max() < 1e-13


@pytest.mark.parametrize("model,hidden", [("default", None), ("linear", None), ("poly", None), ("nn", None),
                                          ("nn", (64, 64, 64))])
@pytest.mark.parametrize("normalise", [False, True])
def test_sensitivities_match_finite_differences(model, hidden, normalise):
    """A, B, c come from exact forward-mode AD of the pinned forward map; central differences confirm them."""
    o = make_oracle(make_aircraft(model, hidden=hidden, normalise=normalise, stall_scaling=True))
    X, U = synthetic_units(12, seed=4, flaps=True)
    dt = 0.01
    Xn, A, B, c = o.step_sens(X, U, dt)
    assert np.array_equal(Xn, o.state_update(X, U, dt))
    for j in range(13):
        h = 1e-6 * max(1.0, np.abs(X[j]).max())
        Xp, Xm = X.copy(), X.copy(); Xp[j] += h; Xm[j] -= h
        fd = (o.state_update(Xp, U, dt) - o.state_update(Xm, U, dt)) / (2 * h)
        assert np.abs(fd - A[:, j]).max() < 2e-6 * max(1.0, np.abs(A[:, j]).max())
    for j in range(7):
        h = 1e-6
        Up, Um = U.copy(), U.copy(); Up[j] += h; Um[j] -= h
        fd = (o.state_update(X, Up, dt) - o.state_update(X, Um, dt)) / (2 * h)
        assert np.abs(fd - B[:, j]).max() < 2e-6 * max(1.0, np.abs(B[:, j]).max())
    h = 1e-7
    fd = (o.state_update(X, U, dt + h) - o.state_update(X, U, dt - h)) / (2 * h)
    assert np.abs(fd - c).max() < 1e-5 * max(1.0, np.abs(c).max())
    # exact structure: dF/dp = [I; 0], dF/dthrust = 0
    assert np.array_equal(A[:, :3], np.broadcast_to(np.eye(13)[:, :3, None], A[:, :3].shape))
    assert not B[:, 3:6].any()


def test_substeps_compose():
    """state_update with n sub-steps of dt/n == n chained single-sub-step updates (dynamics/base.py:463-474)."""
    X, U = synthetic_units(16, seed=8)
    o10 = glider_oracle("poly", substeps=10)
    o1 = glider_oracle("poly", substeps=1)
    x = X.copy()
    for _ in range(10):
        x = o1.state_update(x, U, 0.1 / 10)
    assert np.abs(x - o10.state_update(X, U, 0.1)).max() < 1e-12 * np.abs(x).max()


def test_per_unit_dt_and_rollout_consistency():
    o = glider_oracle("poly", normalise=True)
    X0, U = near_trim_problem(8, 5, seed=1)
    traj = o.rollout(X0, U, 0.01)
    x = X0.copy()
    for k in range(5):
        x = o.state_update(x, U[k], np.full(8, 0.01))
        assert np.array_equal(x, traj[k + 1])


def test_quadrotor_known_answers_and_sensitivities():
    """The Quadrotor plugin (dynamics/quadrotor.py:8-54) in the oracle: hand-computable derivative, free fall, and
    exact AD against central differences."""
    from oracle import Oracle

    body = dict(mass=1.0, reference_area=1.0, span=1.0, chord=1.0, Ixx=1.0, Iyy=1.0, Izz=1.0, Ixz=0.0, com=[0, 0, 0])
    o = Oracle(body, "quad", substeps=1)
    x = np.zeros((13, 1)); x[9] = 1.0
    u = np.zeros((7, 1)); u[:4, 0] = [1, 2, 3, 4]
    xd = o.state_derivative(x, u)[:, 0]
    assert np.allclose(xd[3:6], [0, 0, 10 + 9.81]) and np.allclose(xd[10:13], [1 - 2 - 3 + 4, -1 - 2 + 3 + 4, 0.5 * (1 - 2 + 3 - 4)])
    assert not xd[:3].any() and not xd[6:10].any()
    u[4:] = 7.0  # rows 4-6 are not controls of this plugin
    assert np.array_equal(o.state_derivative(x, u)[:, 0], xd)
    # free fall: RK4 is exact for constant acceleration
    x[3:6, 0] = [1.0, -2.0, 0.5]
    xn = o.state_update(x, np.zeros((7, 1)), 0.1)[:, 0]
    assert np.allclose(xn[:3], [0.1, -0.2, 0.05 + 0.5 * 9.81 * 0.01], atol=1e-15) and np.allclose(xn[3:6], [1, -2, 0.5 + 0.981])
    # off-centre reference point: moments pick up com x F (base.py:253-278)
    body["com"] = [0.1, -0.2, 0.0]
    oc = Oracle(body, "quad", substeps=1)
    assert np.allclose(oc.state_derivative(np.eye(13)[:, 9:10], u)[10:13, 0] - xd[10:13], np.cross([0.1, -0.2, 0.0], [0, 0, 10.0]))
    # sensitivities
    rng = np.random.default_rng(3)
    o2 = Oracle(body, "quad", substeps=2, normalise=True)
    X = rng.normal(size=(13, 6)); X[6:10] /= np.linalg.norm(X[6:10], axis=0)
    U = np.zeros((7, 6)); U[:4] = rng.uniform(-3, 1, (4, 6))
    Xn, A, Bm, c = o2.step_sens(X, U, 0.02)
    h = 1e-6
    for j in range(13):
        d = np.zeros((13, 1)); d[j] = h
        fd = (o2.state_update(X + d, U, 0.02) - o2.state_update(X - d, U, 0.02)) / (2 * h)
        assert np.abs(A[:, j] - fd).max() < 1e-7
    for j in range(7):
        d = np.zeros((7, 1)); d[j] = h
        fd = (o2.state_update(X, U + d, 0.02) - o2.state_update(X, U - d, 0.02)) / (2 * h)
        assert np.abs(Bm[:, j] - fd).max() < 1e-7
    assert not Bm[:, 4:].any()
    fd = (o2.state_update(X, U, 0.02 + h) - o2.state_update(X, U, 0.02 - h)) / (2 * h)
    assert np.abs(c - fd).max() < 1e-6


@pytest.mark.parametrize("model,hidden", [("default", None), ("poly", None), ("linear", None), ("nn", None), ("nn", (64, 64, 64))])
def test_state_derivative_jacobians_match_finite_differences(model, hidden):
    """Fx = df/dx, Fu = df/du (exact forward-mode AD of the pinned f) against central differences of f itself —
    what the implicit defect row and the Baumgarte row of control/base.py:282-304 differentiate."""
    o = make_oracle(make_aircraft(model, hidden=hidden, stall_scaling=True))
    X, U = synthetic_units(40, seed=23, flaps=True)
    xd, Fx, Fu = o.state_derivative_sens(X, U)
    assert np.array_equal(xd, o.state_derivative(X, U))
    assert not Fx[:, :3].any() and np.array_equal(Fx[:3, 3:6], np.broadcast_to(np.eye(3)[:, :, None], (3, 3, 40)))
    assert not Fu[:, 3:6].any()  # thrust has no effect in the reference's force model (aircraft.py:320)
    hs = np.array([1e-3] * 3 + [1e-4] * 3 + [1e-6] * 4 + [1e-5] * 3)
    for j in range(13):
        d = np.zeros_like(X); d[j] = hs[j]
        fd = (o.state_derivative(X + d, U) - o.state_derivative(X - d, U)) / (2 * hs[j])
        assert np.abs(fd - Fx[:, j]).max() <= 2e-6 * max(np.abs(Fx[:, j]).max(), 1.0), j
    for j in (0, 1, 2, 6):
        d = np.zeros_like(U); d[j] = 1e-4
        fd = (o.state_derivative(X, U + d) - o.state_derivative(X, U - d)) / 2e-4
        assert np.abs(fd - Fu[:, j]).max() <= 2e-6 * max(np.abs(Fu[:, j]).max(), 1.0), j


def test_envelope_rows_and_jacobian():
    """The rows of AircraftControl.state_constraint (control/aircraft.py:44-59) = the getters, and their state Jacobian
    against central differences."""
    o = make_oracle(make_aircraft("poly"))
    X, U = synthetic_units(60, seed=29)
    rows, Jx = o.envelope(X)
    a = o.aero(X, U)
    assert np.array_equal(rows[0], (a[0:3] ** 2).sum(axis=0)) and np.array_equal(rows[1], a[5])
    assert np.array_equal(rows[2], a[4]) and np.array_equal(rows[3], X[2])
    assert np.array_equal(Jx[3], np.broadcast_to(np.eye(13)[2][:, None], (13, 60)))  # dz/dx
    assert not Jx[:3, :3].any() and not Jx[:3, 10:].any()  # no dependence on position / body rates
    hs = np.array([1e-3] * 3 + [1e-4] * 3 + [1e-6] * 4 + [1e-5] * 3)
    for j in range(3, 10):
        d = np.zeros_like(X); d[j] = hs[j]
        fd = (o.envelope(X + d)[0] - o.envelope(X - d)[0]) / (2 * hs[j])
        assert np.abs(fd - Jx[:, j]).max() <= 2e-6 * max(np.abs(Jx[:, j]).max(), 1.0), j
    # synthetic units are inside the envelope
    assert (rows[0] >= 400).all() and (rows[0] <= 1e4).all() and (np.abs(rows[1]) <= np.deg2rad(10)).all()
