"""CPU test that pins the NumPy iLQR restatement (the checker of the GPU sweep) to textbook LQR."""
import numpy as np
import scipy.linalg

import ilqr_oracle as io
from aircraft_amd.control import QuadraticCost


def test_backward_pass_converges_to_the_discrete_riccati_solution():
    rng = np.random.default_rng(0)
    A = np.eye(13) + 0.05 * rng.normal(size=(13, 13))
    Bm = 0.1 * rng.normal(size=(13, 7))
    c = QuadraticCost(q=[1.0] * 13, qf=[1.0] * 13, r=[0.5] * 7, reg=0.0)
    H = 400
    X = np.zeros((H + 1, 13, 1)); U = np.zeros((H, 7, 1))
    K, kff, dV = io.backward(c, X, U, np.broadcast_to(A[None, :, :, None], (H, 13, 13, 1)),
                             np.broadcast_to(Bm[None, :, :, None], (H, 13, 7, 1)))
    P = scipy.linalg.solve_discrete_are(A, Bm, np.eye(13), 0.5 * np.eye(7))
    Klqr = -np.linalg.solve(0.5 * np.eye(7) + Bm.T @ P @ Bm, Bm.T @ P @ A)
    assert np.abs(K[0, :, :, 0] - Klqr).max() < 1e-8
    assert np.abs(kff).max() == 0.0 and np.abs(dV).max() == 0.0  # at the optimum (x = x_ref, u = 0) nothing to gain


def test_cost_struct_layout_and_goal_helper():
    import ctypes

    c = QuadraticCost.goal((150.0, 0.0), height=-200.0)
    s = c.struct()
    assert ctypes.sizeof(s) == (13 + 13 + 7 + 13 + 13 + 7 + 7 + 1 + 7 + 1) * 4  # ... + u_lin[7], dt_row
    assert s.dt_row == 0 and not any(s.u_lin)                                      # fixed time, no linear control cost
    assert s.qf[0] == 2000.0 and s.x_goal[0] == 150.0 and s.x_goal[2] == -200.0 and s.u_max[6] == 1.0
    X = np.zeros((3, 13, 2)); X[-1, 0] = [150.0, 149.0]; X[-1, 2] = -200.0
    U = np.zeros((2, 7, 2))
    assert np.allclose(io.cost(c, X, U), [0.0, 1000.0])


def test_goal_loss_restatement_matches_the_reference_formulas_and_its_own_gradient():
    """oracle/ilqr_oracle.py::goal_cost against the reference's loss evaluated term by term on a hand case
    (main/control/control.py:44-68), and goal_model's gradients against central differences of goal_cost: the state gradient
    of every node (terminal quadratics, linear v_x term, speed reward, active inequality) and the control gradient of the
    rate term."""
    from tests.helpers import make_aircraft, make_oracle, near_trim_problem

    ac = make_aircraft("default", normalise=True)
    orc = make_oracle(ac)
    B, H = 3, 6
    X0, U = near_trim_problem(B, H, seed=2)
    X = orc.rollout(X0, U, 0.01)
    rng = np.random.default_rng(5)
    U = U + rng.normal(0, 0.08, U.shape) * (np.arange(7) < 3)[None, :, None]
    goal = np.array([[1.0, 2.0, 3.0], [0.5, -0.5, 0.0]])
    lam = np.array([0.0, 30.0, 5.0])
    g = io.GoalLoss(w_al=4.0, vx_max=60.0)   # v_x(N) ~ 50-65 here: the inequality is active for some instances only
    J = io.goal_cost(orc, g, goal, X, U, lam)
    # term by term, as the reference writes them
    for b in range(B):
        du = U[1:, :, b] - U[:-1, :, b]
        control_loss = 100 * (1 - np.exp(-du ** 2 / 1e-2)).sum()
        goal_loss = 1000 * ((X[-1, :2, b] - goal[:, b]) ** 2).sum()
        height_loss = (X[-1, 2, b] - X[0, 2, b]) ** 2
        vr = orc.aero(X[:-1, :, b].T, np.zeros((7, H)))[:3]
        speed_loss = -((vr * vr).sum(axis=0) / 100).sum() / H
        final_velocity_loss = 1000 * X[-1, 3, b] + 1000 * X[-1, 4, b] ** 2 + 1000 * X[-1, 5, b] ** 2
        s = lam[b] / 8.0
        al = 4.0 * (max(0.0, X[-1, 3, b] - 60.0 + s) ** 2 - s ** 2)
        want = goal_loss + control_loss + height_loss + speed_loss + final_velocity_loss + al
        assert abs(J[b] - want) <= 1e-9 * abs(want)
    nq, nx, ng, ug, uh = io.goal_model(orc, g, goal, X, U, lam)
    # Gauss-Newton curvature of l0: 2 / eps per difference at d = 0 (rows that never move: 100 x 200 per difference), never negative
    assert (uh >= 0).all() and np.allclose(uh[0, 3:], 100 * 2 / 1e-2) and np.allclose(uh[1, 3:], 2 * 100 * 2 / 1e-2)
    gx = nq * (X - nx) + ng
    for k, j, b in [(H, 0, 0), (H, 2, 1), (H, 3, 1), (H, 3, 0), (H, 5, 2), (2, 3, 0), (4, 7, 2), (0, 9, 1)]:
        e = 1e-5 * max(1.0, abs(X[k, j, b]))
        Xp, Xm = X.copy(), X.copy()
        Xp[k, j, b] += e; Xm[k, j, b] -= e
        fd = (io.goal_cost(orc, g, goal, Xp, U, lam)[b] - io.goal_cost(orc, g, goal, Xm, U, lam)[b]) / (2 * e)
        if k == 0 and j == 2:
            continue
        assert abs(fd - gx[k, j, b]) <= 2e-5 * max(1.0, abs(fd)), (k, j, b, fd, gx[k, j, b])
    # (the rate term alone: beside the 1e6-sized goal term a difference quotient in u has no digits left)
    gr = io.GoalLoss(w_goal=0.0, w_height=0.0, w_speed=0.0, w_vx=0.0, w_vyz=0.0)
    assert np.array_equal(io.goal_model(orc, gr, goal, X, U)[3], ug)
    for k, i, b in [(0, 0, 0), (3, 1, 1), (H - 1, 2, 2), (2, 0, 2)]:
        e = 1e-5
        Up, Um = U.copy(), U.copy()
        Up[k, i, b] += e; Um[k, i, b] -= e
        fd = (io.goal_cost(orc, gr, goal, X, Up)[b] - io.goal_cost(orc, gr, goal, X, Um)[b]) / (2 * e)
        assert abs(fd - ug[k, i, b]) <= 1e-5 * max(1.0, abs(fd)), (k, i, b, fd, ug[k, i, b])
    lam2, viol = io.goal_multiplier(g, X, lam)
    assert np.allclose(lam2, np.maximum(0, lam + 8.0 * (X[-1, 3] - 60.0))) and (viol >= 0).all()
