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
                                                       ((32, 32), True, False), ((128, 100, 128), True, False),
                                                       ((128, 128), True, False), ((128, 128), False, True)])
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


def test_hessian_workspace_must_be_reserved_through_the_abi(gpu):
    """The compute entry points never allocate: a raw ABI call without ac_reserve_hess_workspace returns AC_ERR_WORKSPACE
    (MLP stage tensors; sub-step composition buffers), with a message, and works after the reservation — also when the
    sub-step count changes afterwards (the Python layer re-reserves)."""
    import ctypes as C

    import torch

    from aircraft_amd import _lib

    lib = _lib.load()
    X, U, lam = units(8, seed=1)
    Xd, Ud, Ld = dev(X, gpu), dev(U, gpu), dev(lam, gpu)
    out = torch.empty((21, 21, 8), device=gpu)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    for ac in (make_aircraft("nn", hidden=(32, 32)), make_aircraft("poly", substeps=2)):
        ac._sync()
        st = ac._stream()
        assert lib.ac_step_hess_f32(ac._handle, p(Xd), p(Ud), C.c_float(0.01), None, p(Ld), 8, p(out), st) == -6
        assert b"ac_reserve_hess_workspace" in lib.ac_last_error()
        assert lib.ac_reserve_hess_workspace(ac._handle, 8) == 0
        assert lib.ac_step_hess_f32(ac._handle, p(Xd), p(Ud), C.c_float(0.01), None, p(Ld), 8, p(out), st) == 0
    ac = make_aircraft("poly")
    a = ac.step_hess(Xd, Ud, 0.01, Ld).clone()
    ac.physical_integration_substeps = 2  # attribute change after the first reservation: picked up, re-reserved
    b = ac.step_hess(Xd, Ud, 0.01, Ld)
    assert torch.isfinite(b).all() and not torch.equal(a, b)


def test_wide_valu_flavour_is_refused(gpu):
    """Second-order blocks at width > 64 with use_mfma = 0 exist for nets with one to three hidden 128 x 128 products (the
    reverse-sweep kernel's cross-lane flavour, include/aircraft_hip.h; test_hessian_mlp_surrogate covers two): a net with ONE
    hidden LAYER of that width has no product to sweep back through and no instance of the slab kernel, and the call must fail
    loudly, not fall back."""
    import torch
    from aircraft_amd import AircraftHipError
    from tests.helpers import make_aircraft
    ac = make_aircraft("nn", hidden=(128,), use_mfma=False)
    x = torch.zeros((13, 4), device=gpu); x[3] = 30.0; x[9] = 1.0
    with pytest.raises(AircraftHipError, match="UNSUPPORTED"):
        ac.step_hess(x, torch.zeros((7, 4), device=gpu), 0.01, torch.ones((13, 4), device=gpu))


def _single_layer_aircraft(act_last, normalise=True, seed=11):
    """A net that IS one Linear(5, 6) layer (optionally tanh on it): the L == 1 path of the engines (layer<1,1>, no
    first/hidden/last split, 21-slab second-order pass at WT = 2)."""
    from aircraft_amd import Aircraft, AircraftConfiguration, AircraftOpts, MlpData
    from tests.helpers import GLIDER

    base = MlpData.synthetic((16,), seed=seed)
    rng = np.random.default_rng(seed)
    W = (rng.uniform(-1, 1, (6, 5)) / np.sqrt(5)).astype(np.float32)
    b = rng.uniform(-0.3, 0.3, 6).astype(np.float32)
    md = MlpData([W], [b], [act_last], base.input_mean, base.input_std, base.output_mean, base.output_std)
    ac = Aircraft(AircraftOpts(coeff_model_type="nn", coeff_model_path=md, aircraft_config=AircraftConfiguration(dict(GLIDER)),
                               physical_integration_substeps=1))
    ac.normalise = normalise
    return ac


@pytest.mark.parametrize("act_last", [0, 1])
def test_single_layer_net_first_and_second_order(gpu, act_last):
    """L == 1 nets directly (identity and tanh last layer) through step, step_sens and step_hess: the path the folded
    all-linear net of test_activation_free_layers_are_folded reaches only through the host-side fold."""
    from tests.helpers import block_rel_err, parity_report, unit_max_rel

    ac = _single_layer_aircraft(act_last)
    orc = make_oracle(ac)
    X, U, lam = units(150, seed=51)  # ragged
    Xn, A, Bm, c = ac.step_sens(dev(X, gpu), dev(U, gpu), 0.01)
    Xr, Ar, Br, cr = orc.step_sens(X, U, 0.01)
    assert block_rel_err(Xn.cpu().numpy(), Xr) < 1e-5
    for key, g_, w_ in (("A", A, Ar), ("B", Bm, Br), ("c", c, cr)):
        e = unit_max_rel(g_.cpu().numpy(), w_)
        assert e.max() < 1e-4, (key, float(e.max()))
    Hm = ac.step_hess(dev(X, gpu), dev(U, gpu), 0.01, dev(lam, gpu)).cpu().numpy().astype(np.float64)
    want = oracle_step_hessian(orc, X, U, 0.01, lam)
    assert np.isfinite(Hm).all()
    parity_report(f"single_layer_hess[act={act_last}]", rel_block=rel_block(Hm, want))
    assert rel_block(Hm, want) < 5e-4
    # same handle, repeated: identical bits (no dependence on what earlier calls left in registers / LDS / workspace)
    H2 = ac.step_hess(dev(X, gpu), dev(U, gpu), 0.01, dev(lam, gpu)).cpu().numpy().astype(np.float64)
    assert np.array_equal(Hm, H2)


@pytest.mark.parametrize("hidden", [None, (64, 64, 64), (128, 128, 128, 128), "single", "folded"])
def test_hessian_reads_nothing_it_did_not_write(gpu, hidden):
    """Poison the handle's stage-tensor workspace and the output with NaNs before the call: if any element the
    second-order kernels read was not written by THIS call (an unwritten workspace row, a stale output entry), the NaN
    reaches the result.  Covers the L == 1 / 21-slab path (the reference's own net is Linear-tanh-Linear: 5-32-6 after
    the fold is L == 2; 'single' and 'folded' are L == 1), width 64 (one 21-slab pass) and width 128 (29 slab evaluations)."""
    import ctypes as C

    import torch

    from aircraft_amd import Aircraft, AircraftConfiguration, AircraftOpts, MlpData, _lib
    from tests.helpers import GLIDER

    if hidden == "single":
        ac = _single_layer_aircraft(1)
    elif hidden == "folded":
        base = MlpData.synthetic((48, 24, 40), seed=5)
        md = MlpData(base.weights, base.biases, [0, 0, 0, 0], base.input_mean, base.input_std, base.output_mean, base.output_std)
        ac = Aircraft(AircraftOpts(coeff_model_type="nn", coeff_model_path=md, aircraft_config=AircraftConfiguration(dict(GLIDER)),
                                   physical_integration_substeps=1))
        ac.normalise = True
    else:
        ac = make_aircraft("nn", hidden=hidden, normalise=True)
    n = 200
    X, U, lam = units(n, seed=61)
    Xd, Ud, Ld = dev(X, gpu), dev(U, gpu), dev(lam, gpu)
    clean = ac.step_hess(Xd, Ud, 0.01, Ld).clone()
    assert torch.isfinite(clean).all()
    lib = _lib.load()
    ptr, floats = C.c_void_p(), C.c_size_t()
    assert lib.ac_hess_workspace(ac._handle, C.byref(ptr), C.byref(floats)) == 0
    assert floats.value >= n * 504 and ptr.value
    # NaN the whole workspace through a raw device memset pattern (0xFF bytes = NaN) and the output buffer
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    torch.cuda.synchronize()
    assert hip.hipMemset(ptr, 0xFF, floats.value * 4) == 0
    assert hip.hipDeviceSynchronize() == 0
    out = torch.full((21, 21, n), float("nan"), device=gpu)
    got = ac.step_hess(Xd, Ud, 0.01, Ld, out=out)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all()
    assert torch.equal(got, clean)


@pytest.mark.parametrize("model,hidden,substeps,normalise", [("default", None, 3, True), ("poly", None, 10, False),
                                                             ("poly", None, 10, True), ("nn", (64, 64, 64), 4, True),
                                                             ("nn", None, 10, True), ("nn", (128, 128, 128), 3, True)])
def test_hessian_composed_over_substeps(gpu, model, hidden, substeps, normalise):
    """physical_integration_substeps > 1 (the reference's default is 10): the blocks of the sub-steps composed on the
    device — sum_s T_s' H_s T_s with the first-order chain and the pulled-back multipliers — against central differences
    of the oracle's exact Jacobians of the WHOLE sub-stepped update."""
    ac = make_aircraft(model, hidden=hidden, substeps=substeps, normalise=normalise)
    orc = make_oracle(ac)
    n = 70
    X, U, lam = units(n, seed=77)
    dt = 0.02
    Hm = ac.step_hess(dev(X, gpu), dev(U, gpu), dt, dev(lam, gpu)).cpu().numpy().astype(np.float64)
    want = oracle_step_hessian(orc, X, U, dt, lam)
    assert Hm.shape == (21, 21, n) and np.isfinite(Hm).all()
    from tests.helpers import parity_report
    parity_report(f"hess_substeps[{model}-{hidden}-{substeps}]", rel_block=rel_block(Hm, want))
    assert rel_block(Hm, want) < 1e-3
    assert np.abs(Hm - Hm.transpose(1, 0, 2)).max() <= 2e-5 * np.abs(Hm).max()
    for z in (0, 1, 2, 16, 17, 18):
        assert not Hm[z].any() and not Hm[:, z].any()
    # per-unit dt and the in-place shooting layout go through the same composition
    from aircraft_amd.control import MultipleShooting
    import torch
    dts = f32_exact(np.random.default_rng(5).uniform(0.01, 0.03, n))
    H2 = ac.step_hess(dev(X, gpu), dev(U, gpu), dev(dts, gpu), dev(lam, gpu)).cpu().numpy().astype(np.float64)
    assert rel_block(H2[:, :, :24], oracle_step_hessian(orc, X[:, :24], U[:, :24], dts[:24], lam[:, :24])) < 1e-3
    ms = MultipleShooting(system=ac, dt=dt, num_nodes=7, opts={"quaternion": "integration" if normalise else None})
    B = n // 7
    Xs = dev(X[:, : 7 * B].reshape(13, 7, B).transpose(1, 0, 2), gpu)
    Us = dev(U[:, : 7 * B].reshape(7, 7, B).transpose(1, 0, 2), gpu)
    Ls = dev(lam[:, : 7 * B].reshape(13, 7, B).transpose(1, 0, 2), gpu)
    Hs = ms.hessian(Xs, Us, Ls)
    flat = ac.step_hess(dev(X[:, : 7 * B], gpu), dev(U[:, : 7 * B], gpu), dt, dev(lam[:, : 7 * B], gpu))
    assert torch.allclose(Hs.permute(1, 2, 0, 3).reshape(21, 21, 7 * B), flat, rtol=0, atol=0)


@pytest.mark.parametrize("hidden,act_last,n,kernel", [
    ((128, 128, 128, 128), 0, 333, "rev"),   # three hidden products: the cfg3 net, ragged batch (5 full tasks + 13 units)
    ((128, 128, 128, 128), 1, 70, "rev"),    # tanh on the LAST layer as well: R_top carries act'(p), + act''(p) J J
    ((100, 128, 90), 0, 64, "rev"),          # two hidden products, widths below the padded 128
    ((128, 128, 128, 128), 0, 7, "rev"),     # fewer units than one wave
    ((128, 128), 0, 96, "rev"),              # one hidden product: its block and the transposed one take turns in the ring
    ((128, 128, 128, 128, 128), 0, 96, "slabs"),  # four hidden products: more blocks than the reverse plan holds
])
def test_width_128_stage_tensors_by_reverse_sweep(gpu, hidden, act_last, n, kernel):
    """Width 128 on the matrix cores: the stage tensors come from k_nn_stage_tensors_rev (forward tangents, reverse sweep through
    the transposed hidden blocks, per-unit contraction; ac_hess_rev.hpp) for nets with one to three hidden products, from the
    slab-per-derivative kernel otherwise; both against central differences of the oracle's exact float64 Jacobians."""
    import torch

    from aircraft_amd import Aircraft, AircraftConfiguration, AircraftOpts, MlpData
    from tests.helpers import GLIDER

    base = MlpData.synthetic(hidden, seed=17)
    acts = [1] * len(hidden) + [act_last]
    md = MlpData(base.weights, base.biases, acts, base.input_mean, base.input_std, base.output_mean, base.output_std)
    ac = Aircraft(AircraftOpts(coeff_model_type="nn", coeff_model_path=md, aircraft_config=AircraftConfiguration(dict(GLIDER)),
                               physical_integration_substeps=1))
    ac.normalise = True
    orc = make_oracle(ac)
    X, U, lam = units(n, seed=71)
    Xd, Ud, Ld = dev(X, gpu), dev(U, gpu), dev(lam, gpu)
    Hm = ac.step_hess(Xd, Ud, 0.01, Ld)
    torch.cuda.synchronize()
    Hh = Hm.cpu().numpy().astype(np.float64)
    m = min(n, 48)
    want = oracle_step_hessian(orc, X[:, :m], U[:, :m], 0.01, lam[:, :m])
    assert np.isfinite(Hh).all()
    assert rel_block(Hh[:, :, :m], want) < 5e-4
    assert np.abs(Hh - Hh.transpose(1, 0, 2)).max() <= 2e-5 * np.abs(Hh).max()
    # same inputs twice on the same handle: identical bits (scratch slots, ring position and workspace carry nothing over)
    H2 = ac.step_hess(Xd, Ud, 0.01, Ld)
    assert torch.equal(Hm, H2)
    # a unit's block does not depend on its position in the batch or on its neighbours (scratch slots are per wave)
    perm = torch.randperm(n, device=gpu)
    Hp = ac.step_hess(Xd[:, perm].contiguous(), Ud[:, perm].contiguous(), 0.01, Ld[:, perm].contiguous())
    assert torch.equal(Hp, Hm[:, :, perm])
    del kernel  # (documents which stage-tensor kernel the dispatcher picks: profiles/r03_hess_rev_stats.txt shows it by name)
