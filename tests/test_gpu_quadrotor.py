"""GPU parity of the Quadrotor plugin (reference dynamics/quadrotor.py:8-54) — the second SixDOF plugin on the shared
rigid-body kernels — against the float64 oracle."""
import numpy as np
import pytest

from tests.helpers import block_rel_err, f32_exact, rel_fro, unit_max_rel

pytestmark = pytest.mark.gpu


def dev(a, gpu):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(gpu)


def quad_units(n, seed=0):
    from aircraft_amd.synthetic import quat_from_euler

    rng = np.random.default_rng(seed)
    X = np.zeros((13, n))
    X[0:3] = rng.uniform(-10, 10, (3, n)); X[3:6] = rng.uniform(-5, 5, (3, n))
    X[6:10] = quat_from_euler(rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.uniform(-3, 3, n))
    X[10:13] = rng.uniform(-2, 2, (3, n))
    U = rng.uniform(-4, 1, (4, n))  # thrusts (NED: lift is negative z)
    return f32_exact(X), f32_exact(U)


def make(substeps=1, normalise=False, com=(0.0, 0.0, 0.0)):
    from aircraft_amd import Quadrotor
    from oracle import Oracle

    q = Quadrotor()
    q.physical_integration_substeps = substeps
    q.normalise = normalise
    q.com = np.asarray(com, dtype=np.float64)
    orc = Oracle(q.airframe_dict(), "quad", None, substeps=substeps, normalise=normalise, epsilon=q.epsilon, gravity=q.gravity)
    return q, orc


def pad7(U):
    return np.concatenate([U, np.zeros((3,) + U.shape[1:])], axis=0)


def test_plugin_surface(gpu):
    q, _ = make()
    assert q.num_states == 13 and q.num_controls == 4
    f = q.state_update
    assert (f.size1_in(0), f.size1_in(1), f.size1_in(2)) == (13, 4, 1)
    assert q.mass == 1.0 and np.array_equal(q.inertia_tensor, np.eye(3)) and q.physical_integration_substeps == 1
    # hover: sum T = -m g along body z cancels gravity (NED)
    x = np.zeros(13); x[9] = 1.0
    xd = q.state_derivative(x, np.full(4, -9.81 / 4))
    assert np.abs(xd).max() < 1e-6


@pytest.mark.parametrize("substeps,normalise,com", [(1, False, (0, 0, 0)), (3, True, (0.02, -0.01, 0.03))])
def test_quadrotor_step_and_derivative(gpu, substeps, normalise, com):
    q, orc = make(substeps, normalise, com)
    X, U = quad_units(777, seed=3)
    xd = q.state_derivative(dev(X, gpu), dev(U, gpu)).cpu().numpy()
    want = orc.state_derivative(X, pad7(U))
    assert np.abs(xd - want).max() / np.abs(want).max() < 2e-6
    xn = q.state_update(dev(X, gpu), dev(U, gpu), 0.02).cpu().numpy()
    assert block_rel_err(xn, orc.state_update(X, pad7(U), 0.02)) < 2e-6
    # numpy in -> numpy out, single column; a ready 7-row control buffer is accepted as is
    one = q.state_update(X[:, 0], U[:, 0], 0.02)
    assert one.shape == (13,) and block_rel_err(one[:, None], orc.state_update(X[:, :1], pad7(U[:, :1]), 0.02)) < 2e-6
    same = q.state_update(dev(X, gpu), dev(pad7(U), gpu), 0.02).cpu().numpy()
    assert np.array_equal(same, xn)


def test_quadrotor_rollout(gpu):
    q, orc = make()
    X, U0 = quad_units(200, seed=4)
    rng = np.random.default_rng(5)
    U = f32_exact(U0[None] + rng.normal(0, 0.2, (40, 4, 200)))
    traj = q.rollout(dev(X, gpu), dev(U, gpu), 0.01).cpu().numpy()
    want = orc.rollout(X, np.concatenate([U, np.zeros((40, 3, 200))], axis=1), 0.01)
    assert traj.shape == (41, 13, 200)
    assert block_rel_err(traj, want) < 1e-5


@pytest.mark.parametrize("substeps,normalise", [(1, True), (2, False)])
def test_quadrotor_sensitivities(gpu, substeps, normalise):
    q, orc = make(substeps, normalise, com=(0.01, 0.0, -0.02))
    X, U = quad_units(300, seed=6)
    Xn, A, Bm, c = q.step_sens(dev(X, gpu), dev(U, gpu), 0.02)
    Xr, Ar, Br, cr = orc.step_sens(X, pad7(U), 0.02)
    assert Bm.shape == (13, 4, 300)
    assert block_rel_err(Xn.cpu().numpy(), Xr) < 2e-6
    assert rel_fro(A.cpu().numpy(), Ar) < 1e-5
    assert rel_fro(Bm.cpu().numpy(), Br[:, :4]) < 1e-5 and not Br[:, 4:].any()
    assert rel_fro(c.cpu().numpy(), cr) < 1e-5
    assert unit_max_rel(A.cpu().numpy(), Ar).max() < 2e-5 and unit_max_rel(Bm.cpu().numpy(), Br[:, :4]).max() < 2e-5  # per unit
    # a caller-provided 7-column buffer gets zeros in the three unused columns
    import torch
    out = (torch.empty(13, 300, device=gpu), torch.empty(13, 13, 300, device=gpu), torch.full((13, 7, 300), 9.0, device=gpu),
           torch.empty(13, 300, device=gpu))
    q.step_sens(dev(X, gpu), dev(U, gpu), 0.02, out=out)
    assert not out[2][:, 4:].any() and torch.equal(out[2][:, :4], Bm)


def test_quadrotor_getters_and_shooting(gpu):
    from aircraft_amd.control import MultipleShooting

    q, orc = make()
    X, U = quad_units(128, seed=8)
    ref = orc.aero(X, pad7(U))
    F = q.forces_frd(dev(X, gpu), dev(U, gpu)).cpu().numpy()
    M = q.moments_frd(dev(X, gpu), dev(U, gpu)).cpu().numpy()
    assert np.abs(F - ref[13:16]).max() < 1e-5 and np.abs(M - ref[16:19]).max() < 1e-5
    assert np.allclose(F[2], U.sum(axis=0), atol=1e-5) and not F[:2].any()
    assert np.abs(q.phi(dev(X, gpu)).cpu().numpy() - ref[19]).max() < 1e-5
    # multiple shooting on the plugin (device buffers carry 7 control rows)
    ms = MultipleShooting(system=q, dt=0.01, num_nodes=12, opts={"quaternion": "integration"})
    assert ms.control_dim == 4 and q.normalise is True
    Uh = f32_exact(np.tile(pad7(U)[None], (12, 1, 1)))
    Xt = ms.rollout(dev(X, gpu), dev(Uh, gpu))
    Xn, A, Bm, c = ms.linearise(Xt, dev(Uh, gpu))
    orc.p.normalise = 1
    k = 5
    Xr, Ar, Br, cr = orc.step_sens(Xt[k].cpu().numpy().astype(np.float64), Uh[k], 0.01)
    assert block_rel_err(Xn[k].cpu().numpy(), Xr) < 2e-6 and rel_fro(A[k].cpu().numpy(), Ar) < 1e-5
    assert rel_fro(Bm[k].cpu().numpy(), Br) < 1e-5
    assert float(ms.defects(Xt, dev(Uh, gpu)).abs().max()) < 1e-5  # rollout carries f64 between steps, shooting restarts from fp32
