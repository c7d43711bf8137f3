"""Synthetic, in-envelope inputs for parity tests and bench.py (SURVEY.md §8d spec, seed 42 =
reference config.py:5).  NumPy float64 on the host; callers move them to the device."""
from __future__ import annotations

import numpy as np

# Airframe of data/glider/problem_definition.json:12-24 with the drivers' CoM override
# (main/control/control.py:169-172).
GLIDER = {
    "mass": 4.0, "span": 2.0, "length": 1.2, "chord": 0.124605, "reference_area": 0.238,
    "aero_centre_offset": [0.0131991, -1.78875e-08, 0.00313384],
    "Ixx": 0.155, "Iyy": 0.114, "Izz": 0.262, "Ixz": 0.01, "glide_ratio": 3.0, "r_min": 1.0,
}

TRIM_STATE = np.array([0, 0, -200, 50, 0, 0, 0, 0, 0, 1, 0, 0, 0], dtype=np.float64)  # problem_definition.json:7


def quat_from_euler(roll, pitch, yaw):
    """xyzw quaternion (body -> NED) from aerospace ZYX Euler angles."""
    cr, sr = np.cos(roll / 2), np.sin(roll / 2)
    cp, sp = np.cos(pitch / 2), np.sin(pitch / 2)
    cy, sy = np.cos(yaw / 2), np.sin(yaw / 2)
    return np.stack([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy,
                     cr * cp * cy + sr * sp * sy])


def quat_rotate(q, v):
    """q (x) (v,0) (x) q^-1 for unit xyzw q; q (4,n), v (3,n)."""
    qv, qw = q[:3], q[3]
    t = 2.0 * np.cross(qv, v, axis=0)
    return v + qw * t + np.cross(qv, t, axis=0)


def synthetic_states(n, rng):
    """(13, n) states inside the reference envelope (control/aircraft.py:47-59)."""
    X = np.empty((13, n))
    X[0] = rng.uniform(-100, 100, n)
    X[1] = rng.uniform(-100, 100, n)
    X[2] = rng.uniform(-300, -100, n)
    V = rng.uniform(30, 80, n)
    al = np.deg2rad(rng.uniform(-8, 8, n))
    be = np.deg2rad(rng.uniform(-5, 5, n))
    vb = np.stack([V * np.cos(al) * np.cos(be), V * np.sin(be), V * np.sin(al) * np.cos(be)])
    q = quat_from_euler(np.deg2rad(rng.uniform(-30, 30, n)), np.deg2rad(rng.uniform(-15, 15, n)),
                        np.deg2rad(rng.uniform(-180, 180, n)))
    X[3:6] = quat_rotate(q, vb)
    X[6:10] = q
    X[10:13] = rng.normal(0.0, 0.2, (3, n))
    return X


def synthetic_controls(H, n, rng, flaps=False):
    """(H, 7, n): aileron/elevator/rudder as clipped random walks in [-5, 5] deg (step sigma 0.5);
    thrust 0; flaps 0 unless requested."""
    U = np.zeros((H, 7, n))
    cur = rng.uniform(-2, 2, (3, n))
    for k in range(H):
        cur = np.clip(cur + rng.normal(0, 0.5, (3, n)), -5, 5)
        U[k, :3] = cur
    if flaps:
        U[:, 6] = rng.uniform(0, 1, (1, n))
    return U


def synthetic_units(n, seed=42, flaps=False):
    """n independent (x_k, u_k) pairs: X (13, n), U (7, n)."""
    rng = np.random.default_rng(seed)
    X = synthetic_states(n, rng)
    U = synthetic_controls(1, n, rng, flaps=flaps)[0]
    return X, U


def synthetic_problem(B, H, seed=42):
    """Initial states X0 (13, B) and control sequences U (H, 7, B) of B independent MPC instances."""
    rng = np.random.default_rng(seed)
    return synthetic_states(B, rng), synthetic_controls(H, B, rng)


def near_trim_problem(B, H, seed=42):
    """B instances that start near the glider's trim (zero controls, CoM override) and are steered by small
    control random walks (|delta| <= 2 deg): trajectories that stay inside the flight envelope for H steps."""
    rng = np.random.default_rng(seed)
    X = np.empty((13, B))
    X[0] = rng.uniform(-100, 100, B)
    X[1] = rng.uniform(-100, 100, B)
    X[2] = rng.uniform(-300, -100, B)
    V = rng.uniform(45, 80, B)
    al = np.deg2rad(rng.uniform(-2, 2, B))
    be = np.deg2rad(rng.uniform(-2, 2, B))
    vb = np.stack([V * np.cos(al) * np.cos(be), V * np.sin(be), V * np.sin(al) * np.cos(be)])
    q = quat_from_euler(np.deg2rad(rng.uniform(-15, 15, B)), np.deg2rad(rng.uniform(-5, 5, B)),
                        np.deg2rad(rng.uniform(-180, 180, B)))
    X[3:6] = quat_rotate(q, vb)
    X[6:10] = q
    X[10:13] = rng.normal(0.0, 0.05, (3, B))
    U = np.zeros((H, 7, B))
    cur = np.zeros((3, B))
    for k in range(H):
        cur = np.clip(cur + rng.normal(0, 0.2, (3, B)), -2, 2)
        U[k, :3] = cur
    return X, U


def cruise_problem(B, seed=42):
    """B gliders near trim heading roughly north (+x) at 200 m: initial states of a closed loop that can run for minutes
    (the receding-horizon regulator of QuadraticCost.cruise holds heading and wings level while the glider sinks)."""
    rng = np.random.default_rng(seed)
    X = np.zeros((13, B))
    X[0] = rng.uniform(-20, 20, B)
    X[1] = rng.uniform(-20, 20, B)
    X[2] = rng.uniform(-220, -180, B)
    V = rng.uniform(45, 60, B)
    al = np.deg2rad(rng.uniform(-1, 1, B))
    be = np.deg2rad(rng.uniform(-1, 1, B))
    vb = np.stack([V * np.cos(al) * np.cos(be), V * np.sin(be), V * np.sin(al) * np.cos(be)])
    q = quat_from_euler(np.deg2rad(rng.uniform(-10, 10, B)), np.deg2rad(rng.uniform(-3, 3, B)),
                        np.deg2rad(rng.uniform(-10, 10, B)))
    X[3:6] = quat_rotate(q, vb)
    X[6:10] = q
    X[10:13] = rng.normal(0.0, 0.03, (3, B))
    return X
