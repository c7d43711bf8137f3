"""GPU tests of the second-order step sensitivities: Hessian of lambda . F(x, u, dt) per unit against
central differences of the oracle's exact float64 Jacobians."""
import numpy as np
import pytest

from tests.helpers import f32_exact, make_aircraft, make_oracle, oracle_step_hessian, synthetic_units

pytestmark = pytest.mark.gpu


def dev(a, gpu):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(gpu)


def rel_block(got, want):
    """max over units of ||got - want||_F / ||want||_F"""
    num = np.sqrt(((got - want) ** 2).sum(axis=(0, 1)))
    den = np.sqrt((want ** 2).sum(axis=(0, 1)))
    return float((num / np.maximum(den, 1e-30)).max())


def units(n, seed):
    X, U = synthetic_units(n, seed=seed, flaps=True)
    rng = np.random.default_rng(seed + 100)
    lam = f32_exact(rng.normal(size=(13, n)))
    return X, U, lam


@pytest.mark.parametrize("model,normalise", [("default", False), ("default", True), ("linear", True), ("poly", False),
                                             ("poly", True)])
def test_hessian_matches_finite_differences_of_exact_jacobians(gpu, model, normalise):
    ac = make_aircraft(model, normalise=normalise, stall_scaling=(model == "default"))
    orc = make_oracle(ac)
    X, U, lam = units(96, seed=31)
    Hm = ac.step_hess(dev(X, gpu), dev(U, gpu), 0.01, dev(lam, gpu)).cpu().numpy().astype(np.float64)
    want = oracle_step_hessian(orc, X, U, 0.01, lam)
    assert Hm.shape == (21, 21, 96)
    assert rel_block(Hm, want) < 2e-4
    # structure: symmetric (every (a, b) pair is computed on its own lane), zero rows for position and the thrust controls
    assert np.abs(Hm - Hm.transpose(1, 0, 2)).max() <= 1e-5 * np.abs(Hm).max()
    for z in (0, 1, 2, 16, 17, 18):
        assert not Hm[z].any() and not Hm[:, z].any()
    assert np.abs(want[[0, 1, 2, 16, 17, 18]]).max() < 1e-6 * np.abs(want).max()


@pytest.mark.parametrize("hidden,normalise,use_mfma", [((64, 64, 64), True, True), (None, False, True), ((128, 128, 128, 128), True, True),
                                                       ((32, 32), True, False)])
def test_hessian_mlp_surrogate(gpu, hidden, normalise, use_mfma):
    """MLP surrogate: stage tensors (y, J, d2y/dz dz) from the MFMA engine's second-order mode, then the same
    second-order forward-mode kernel.  hidden=None is the reference's own network (Linear-tanh-Linear)."""
    ac = make_aircraft("nn", hidden=hidden, normalise=normalise, use_mfma=use_mfma)
    orc = make_oracle(ac)
    X, U, lam = units(80, seed=41)
    Hm = ac.step_hess(dev(X, gpu), dev(U, gpu), 0.01, dev(lam, gpu)).cpu().numpy().astype(np.float64)
    want = oracle_step_hessian(orc, X, U, 0.01, lam)
    assert np.isfinite(Hm).all()
    assert rel_block(Hm, want) < 5e-4
    assert np.abs(Hm - Hm.transpose(1, 0, 2)).max() <= 2e-5 * np.abs(Hm).max()
    for z in (0, 1, 2, 16, 17, 18):
        assert not Hm[z].any() and not Hm[:, z].any()
    # the workspace can be sized ahead of time (hipGraph capture); a later, larger call re-uses / grows it
    from aircraft_amd import _lib
    assert _lib.load().ac_reserve_hess_workspace(ac._handle, 300) == 0
    assert _lib.load().ac_reserve_hess_workspace(ac._handle, -1) == -1
    X2, U2, lam2 = units(300, seed=42)
    H2 = ac.step_hess(dev(X2, gpu), dev(U2, gpu), 0.01, dev(lam2, gpu)).cpu().numpy().astype(np.float64)
    assert rel_block(H2[:, :, :40], oracle_step_hessian(orc, X2[:, :40], U2[:, :40], 0.01, lam2[:, :40])) < 5e-4


def test_hessian_per_unit_dt_and_numpy_vector(gpu):
    ac = make_aircraft("poly", normalise=True)
    orc = make_oracle(ac)
    X, U, lam = units(40, seed=33)
    dts = f32_exact(np.random.default_rng(2).uniform(0.005, 0.02, 40))
    Hm = ac.step_hess(dev(X, gpu), dev(U, gpu), dev(dts, gpu), dev(lam, gpu)).cpu().numpy().astype(np.float64)
    assert rel_block(Hm, oracle_step_hessian(orc, X, U, dts, lam)) < 2e-4
    one = ac.step_hess(X[:, 3], U[:, 3], float(dts[3]), lam[:, 3])  # numpy vectors in -> (21, 21) float64 out
    assert one.shape == (21, 21) and np.allclose(one, Hm[:, :, 3], rtol=0, atol=1e-6 * np.abs(Hm[:, :, 3]).max())


def test_hessian_quadrotor(gpu):
    from aircraft_amd import Quadrotor
    from oracle import Oracle
    from tests.test_gpu_quadrotor import pad7, quad_units

    q = Quadrotor()
    q.normalise = True
    q.com = np.array([0.02, -0.01, 0.03])
    orc = Oracle(q.airframe_dict(), "quad", None, substeps=1, normalise=True, epsilon=q.epsilon, gravity=q.gravity)
    X, U = quad_units(64, seed=9)
    lam = f32_exact(np.random.default_rng(3).normal(size=(13, 64)))
    Hm = q.step_hess(dev(X, gpu), dev(U, gpu), 0.02, dev(lam, gpu)).cpu().numpy().astype(np.float64)
    want = oracle_step_hessian(orc, X, pad7(U), 0.02, lam)
    assert rel_block(Hm, want) < 2e-4
    for z in (0, 1, 2, 17, 18, 19):  # position, and the three control rows this plugin does not have
        assert not Hm[z].any() and not Hm[:, z].any()
    assert np.abs(Hm[13:17, 13:17]).max() > 0  # thrust-thrust curvature comes from the normalisation and RK4 coupling


def test_shooting_hessian_in_place_equals_flat_call(gpu):
    import torch
    from aircraft_amd.control import MultipleShooting
    from aircraft_amd.synthetic import synthetic_problem

    ac = make_aircraft("poly", normalise=True)
    B, H = 24, 5
    X0, U = synthetic_problem(B, H, seed=23)
    ms = MultipleShooting(system=ac, dt=0.01, num_nodes=H, opts={"quaternion": "integration"})
    Ud = dev(U, gpu)
    Xt = ms.rollout(dev(X0, gpu), Ud)
    Lam = dev(np.random.default_rng(5).normal(size=(H, 13, B)), gpu)
    Hs = ms.hessian(Xt, Ud, Lam)
    assert Hs.shape == (H, 21, 21, B)
    flatX = Xt[:H].permute(1, 0, 2).reshape(13, H * B).contiguous()
    flatU = Ud.permute(1, 0, 2).reshape(7, H * B).contiguous()
    flatL = Lam.permute(1, 0, 2).reshape(13, H * B).contiguous()
    Hf = ac.step_hess(flatX, flatU, 0.01, flatL)
    assert torch.equal(Hs.permute(1, 2, 0, 3).reshape(21, 21, H * B), Hf)


def test_hessian_unsupported_cases_fail_loudly(gpu):
    from aircraft_amd import AircraftHipError

    X, U, lam = units(8, seed=1)
    ac = make_aircraft("poly")
    ac.physical_integration_substeps = 2
    with pytest.raises(AircraftHipError, match="UNSUPPORTED"):
        ac.step_hess(dev(X, gpu), dev(U, gpu), 0.01, dev(lam, gpu))


def test_wide_valu_flavour_is_refused(gpu):
    """Second-order blocks at width > 64 exist on the MFMA path only (include/aircraft_hip.h): the call must fail loudly,
    not fall back."""
    import torch
    from aircraft_amd import AircraftHipError
    from tests.helpers import make_aircraft
    ac = make_aircraft("nn", hidden=(128, 128), use_mfma=False)
    x = torch.zeros((13, 4), device=gpu); x[3] = 30.0; x[9] = 1.0
    with pytest.raises(AircraftHipError, match="UNSUPPORTED"):
        ac.step_hess(x, torch.zeros((7, 4), device=gpu), 0.01, torch.ones((13, 4), device=gpu))
