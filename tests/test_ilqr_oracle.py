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
