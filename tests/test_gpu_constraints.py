"""GPU parity of the rows the reference's NLP builds besides the explicit defect:
  * f with its Jacobians df/dx, df/du                     — ca.jacobian(state_derivative, .)
  * implicit defect rows and their blocks                  — control/base.py:282-284
  * quaternion norm / Baumgarte rows                       — control/base.py:285-304
  * flight-envelope rows and their state Jacobian          — control/aircraft.py:44-59
against the float64 oracle (exact forward-mode AD), every unit on its own."""
import numpy as np
import pytest

from tests.helpers import (block_rel_err, f32_exact, make_aircraft, make_oracle, near_trim_problem, parity_report, synthetic_problem,
                           synthetic_units, unit_max_rel, unit_rowblock_rel)

pytestmark = pytest.mark.gpu

MODELS = {"default": ("default", None, True), "linear": ("linear", None, True), "poly": ("poly", None, True),
          "real": ("nn", None, True), "cfg2_3x64": ("nn", (64, 64, 64), True), "cfg2_3x64_valu": ("nn", (64, 64, 64), False),
          "cfg3_4x128": ("nn", (128, 128, 128, 128), True)}


def dev(a, gpu):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(gpu)


def build(name, **kw):
    model, hidden, mf = MODELS[name]
    return make_aircraft(model, hidden=hidden, use_mfma=mf, **kw)


@pytest.mark.parametrize("name", list(MODELS))
def test_state_derivative_jacobians(gpu, name):
    ac = build(name, stall_scaling=True)
    n = 333  # ragged
    X, U = synthetic_units(n, seed=71, flaps=True)
    X, U = f32_exact(X), f32_exact(U)
    xd, Fx, Fu = ac.state_derivative_sens(dev(X, gpu), dev(U, gpu))
    xr, Fxr, Fur = make_oracle(ac).state_derivative_sens(X, U)
    assert block_rel_err(xd.cpu().numpy(), xr) < 2e-5
    # the value is the one ac_state_derivative_f32 returns
    # (to rounding: the dual-number path forms a / b as a * (1 / b))
    assert block_rel_err(xd.cpu().numpy(), ac.state_derivative(dev(X, gpu), dev(U, gpu)).cpu().numpy()) < 2e-5
    Fx_, Fu_ = Fx.cpu().numpy(), Fu.cpu().numpy()
    for key, g, w in (("Fx", Fx_, Fxr), ("Fu", Fu_, Fur)):
        e, eb = unit_max_rel(g, w), unit_rowblock_rel(g, w)
        parity_report(f"deriv_sens[{name}]", block=key, unit_rel_max=float(e.max()), rowblock_rel_max=float(eb.max()))
        assert e.max() < 2e-5, (key, float(e.max()))
        assert eb.max() < 2e-4, (key, float(eb.max()))
    # exact structure: f does not depend on p; p_dot = v; thrust has no effect
    assert not Fx_[:, :3].any() and np.array_equal(Fx_[:3, 3:6], np.broadcast_to(np.eye(3)[:, :, None], (3, 3, n)))
    assert not Fu_[:, 3:6].any()


def test_state_derivative_jacobians_numpy_vector_and_quadrotor(gpu):
    from aircraft_amd import Quadrotor

    ac = build("poly")
    x = np.array([0, 0, -200, 50, 0, 1, 0, 0.02, 0, 1, 0.01, 0, 0.0]); u = np.array([1.0, 3, 0, 0, 0, 0, 0.2])
    xd, Fx, Fu = ac.state_derivative_sens(x, u)
    assert isinstance(xd, np.ndarray) and xd.shape == (13,) and Fx.shape == (13, 13) and Fu.shape == (13, 7)
    xr, Fxr, Fur = make_oracle(ac).state_derivative_sens(x[:, None], u[:, None])
    assert np.abs(Fx - Fxr[..., 0]).max() < 2e-5 * np.abs(Fxr).max()
    q = Quadrotor()
    X, _ = synthetic_units(40, seed=3)
    T = np.random.default_rng(0).uniform(1, 4, (4, 40))
    xd, Fx, Fu = q.state_derivative_sens(dev(X, gpu), dev(T, gpu))
    xr, Fxr, Fur = make_oracle(q).state_derivative_sens(f32_exact(X), np.concatenate([f32_exact(T), np.zeros((3, 40))]))
    assert Fu.shape == (13, 4, 40)
    assert unit_max_rel(Fx.cpu().numpy(), Fxr).max() < 2e-5 and unit_max_rel(Fu.cpu().numpy(), Fur[:, :4]).max() < 2e-5


@pytest.mark.parametrize("name", ["poly", "cfg2_3x64"])
def test_implicit_defect_rows_and_blocks(gpu, name):
    """opts['integration'] = 'implicit': r_k = x_{k+1} - (x_k + dt f(x_{k+1}, u_k)) and its Jacobian blocks, on a batch of
    trajectories in the rollout-shaped buffers (the next nodes are read in place through a view)."""
    import torch
    from aircraft_amd.control import MultipleShooting

    ac = build(name)
    B, H, dt = 40, 9, 0.01
    X0, U = near_trim_problem(B, H, seed=5)  # trajectories that stay inside the envelope
    ms = MultipleShooting(system=ac, dt=dt, num_nodes=H, opts={"integration": "implicit", "quaternion": "constraint"})
    Ud = dev(U, gpu)
    X = ms.rollout(dev(X0, gpu), Ud)  # an explicit rollout: the implicit rows are small but not zero on it
    r = ms.defects(X, Ud)
    Xh = X.cpu().numpy().astype(np.float64)
    Uh = f32_exact(U)
    orc = make_oracle(ac)
    flatX = np.ascontiguousarray(Xh[1:].transpose(1, 0, 2).reshape(13, H * B))
    flatU = np.ascontiguousarray(Uh.transpose(1, 0, 2).reshape(7, H * B))
    fr, Fxr, Fur = orc.state_derivative_sens(flatX, flatU)
    want = (Xh[1:] - Xh[:-1]) - dt * fr.reshape(13, H, B).transpose(1, 0, 2)
    got = r.cpu().numpy()
    assert np.abs(got - want).max() <= 1e-5 * np.maximum(np.abs(Xh[1:]), 1.0).max()
    # explicit rows of the same trajectory are (near) zero, implicit ones differ from them at O(dt^2)
    assert float(ms.defects(X, Ud, integration="explicit").abs().max()) < 1e-4
    r2, Jn, Ju, jdt = ms.linearise_implicit(X, Ud)
    # defects() takes f from the forward kernel, linearise_implicit() from the value part of the derivative-sensitivity
    # kernel: the same fp32 arithmetic up to the order of a few roundings
    assert float((r2 - r).abs().max()) <= 2e-6 * max(float(X.abs().max()), 1.0)
    eye = np.eye(13)[None, :, :, None]
    Jn_w = eye - dt * Fxr.reshape(13, 13, H, B).transpose(2, 0, 1, 3)
    Ju_w = -dt * Fur.reshape(13, 7, H, B).transpose(2, 0, 1, 3)
    mv = lambda a: np.moveaxis(a, 0, -2).reshape(a.shape[1], a.shape[2], -1)  # (H, r, c, B) -> (r, c, H*B): unit last  # noqa: E731
    assert unit_max_rel(mv(Jn.cpu().numpy()), mv(Jn_w)).max() < 2e-5 and unit_max_rel(mv(Ju.cpu().numpy()), mv(Ju_w)).max() < 2e-5
    assert block_rel_err(-jdt.cpu().numpy(), fr.reshape(13, H, B).transpose(1, 0, 2)) < 2e-5
    with pytest.raises(NotImplementedError):
        ms.defects(X, Ud, integration="midpoint")


@pytest.mark.parametrize("mode", ["constraint", "baumgarte"])
def test_quaternion_rows(gpu, mode):
    """q.q - 1 and the Baumgarte-stabilised row 4 phi_dot + 4 phi of control/base.py:285-304 with d/dx, d/du, against the
    oracle's f, Fx, Fu in float64."""
    from aircraft_amd.control import MultipleShooting

    ac = build("poly")
    B, H = 33, 6
    X0, U = synthetic_problem(B, H + 1, seed=9)
    ms = MultipleShooting(system=ac, dt=0.01, num_nodes=H, opts={"quaternion": mode})
    assert ac.normalise is False
    Ud = dev(U, gpu)
    X = ms.system.rollout(dev(X0, gpu), Ud, 0.01)  # H + 2 nodes; un-normalised, so q.q - 1 is not exactly zero
    X = X * 1.0
    X[:, 6:10] *= 1.003  # make the rows visibly non-zero
    Xn, Un = X[1 : H + 1].contiguous(), Ud[1 : H + 1].contiguous()
    row, Jx, Ju = ms.quaternion_rows(Xn, Un)
    Xh = Xn.cpu().numpy().astype(np.float64); Uh = f32_exact(U)[1 : H + 1]
    q = Xh[:, 6:10]
    phi = (q * q).sum(axis=1) - 1.0
    if mode == "constraint":
        assert np.abs(row.cpu().numpy() - phi).max() < 1e-6
        Jw = np.zeros((H, 13, B)); Jw[:, 6:10] = 2 * q
        assert np.abs(Jx.cpu().numpy() - Jw).max() < 1e-6 and not Ju.cpu().numpy().any()
        return
    orc = make_oracle(ac)
    fr, Fxr, Fur = orc.state_derivative_sens(np.ascontiguousarray(Xh.transpose(1, 0, 2).reshape(13, H * B)),
                                             np.ascontiguousarray(Uh.transpose(1, 0, 2).reshape(7, H * B)))
    fr = fr.reshape(13, H, B).transpose(1, 0, 2); Fxr = Fxr.reshape(13, 13, H, B).transpose(2, 0, 1, 3)
    Fur = Fur.reshape(13, 7, H, B).transpose(2, 0, 1, 3)
    qd = fr[:, 6:10]
    want = 2 * 2.0 * (2 * (q * qd).sum(axis=1)) + 4.0 * phi
    Jw = 8.0 * np.einsum("hcb,hcjb->hjb", q, Fxr[:, 6:10])
    Jw[:, 6:10] += 8.0 * qd + 8.0 * q
    Juw = 8.0 * np.einsum("hcb,hcjb->hjb", q, Fur[:, 6:10])
    scale = max(np.abs(want).max(), 1.0)
    assert np.abs(row.cpu().numpy() - want).max() < 2e-5 * scale
    assert np.abs(Jx.cpu().numpy() - Jw).max() < 2e-5 * max(np.abs(Jw).max(), 1.0)
    assert np.abs(Ju.cpu().numpy() - Juw).max() < 2e-5 * max(np.abs(Juw).max(), 1.0)


@pytest.mark.parametrize("name", ["poly", "real"])
def test_envelope_rows(gpu, name):
    """|v_rel|^2, beta, alpha, z and their state Jacobian (control/aircraft.py:44-59) — flat and in-place shooting forms."""
    import torch
    from aircraft_amd.control import MultipleShooting

    ac = build(name)
    n = 500
    X, _ = synthetic_units(n, seed=13)
    X = f32_exact(X)
    rows, Jx = ac.envelope(dev(X, gpu))
    rr, Jr = make_oracle(ac).envelope(X)
    sc = np.maximum(np.abs(rr).max(axis=1, keepdims=True), 1e-3)
    assert (np.abs(rows.cpu().numpy() - rr) / sc).max() < 2e-6
    for r in range(4):
        assert unit_max_rel(Jx.cpu().numpy()[r], Jr[r]).max() < 1e-4, r  # measured 2.4e-5 (cancellation in d|v_rel|^2/dq)
    # same numbers as the getters the reference's rows are built from
    assert torch.allclose(rows[1], ac.beta(dev(X, gpu)), atol=1e-7) and torch.allclose(rows[2], ac.alpha(dev(X, gpu)), atol=1e-7)
    lo = torch.tensor([b[0] for b in ac.ENVELOPE_BOUNDS], device=gpu)[:, None]
    hi = torch.tensor([b[1] for b in ac.ENVELOPE_BOUNDS], device=gpu)[:, None]
    assert bool(((rows >= lo) & (rows <= hi)).all())  # SURVEY-spec synthetic states are inside the envelope
    rows_only, none = ac.envelope(dev(X, gpu), want_jacobian=False)
    assert none is None and torch.equal(rows_only, rows)
    # shooting form on a trajectory buffer, in place
    B, H = 20, 25
    ms = MultipleShooting(system=ac, dt=0.01, num_nodes=H)
    Xt = dev(X[:, : B * H].reshape(13, H, B).transpose(1, 0, 2), gpu)
    rt, Jt = ms.envelope(Xt)
    assert rt.shape == (H, 4, B) and Jt.shape == (H, 4, 13, B)
    assert torch.equal(rt.permute(1, 0, 2).reshape(4, H * B), rows[:, : B * H])
    assert torch.equal(Jt.permute(1, 2, 0, 3).reshape(4, 13, H * B), Jx[:, :, : B * H])
    # numpy vector in -> numpy out; quadrotor refused
    r1, J1 = ac.envelope(X[:, 0])
    assert isinstance(r1, np.ndarray) and r1.shape == (4,) and J1.shape == (4, 13)


def test_constraint_entry_points_status_codes(gpu):
    import ctypes as C

    import torch

    from aircraft_amd import AircraftHipError, Quadrotor, _lib

    lib = _lib.load()
    ac = build("default")
    ac._sync()
    h, st = ac._handle, ac._stream()
    f = lambda *s: torch.zeros(s, device=gpu)  # noqa: E731
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    X, U = f(13, 8), f(7, 8)
    xd, Fx, Fu, rows = f(13, 8), f(13, 13, 8), f(13, 7, 8), f(4, 8)
    assert lib.ac_state_derivative_sens_f32(h, p(X), p(U), 8, p(xd), p(Fx), p(Fu), st) == 0
    assert lib.ac_state_derivative_sens_f32(h, p(X), p(U), 8, p(xd), None, p(Fu), st) == -1
    assert lib.ac_state_derivative_sens_f32(h, None, None, 0, None, None, None, st) == 0
    assert lib.ac_shoot_derivative_sens_f32(h, p(X), p(U), -1, 2, p(xd), p(Fx), p(Fu), st) == -1
    assert lib.ac_envelope_f32(h, p(X), 8, p(rows), None, st) == 0
    assert lib.ac_envelope_f32(h, p(X), 8, None, None, st) == -1
    assert lib.ac_quat_rows_f32(h, 2, p(X), None, None, None, 8, 1, p(f(8)), p(f(13, 8)), p(f(7, 8)), st) == -1
    assert lib.ac_quat_rows_f32(h, 1, p(X), None, None, None, 8, 1, p(f(8)), p(f(13, 8)), p(f(7, 8)), st) == -1
    q = Quadrotor()
    q._sync()
    assert lib.ac_envelope_f32(q._handle, p(X), 8, p(rows), None, st) == -3
    assert b"fixed-wing" in lib.ac_last_error()


@pytest.mark.parametrize("name", ["poly", "real", "cfg2_3x64_valu"])
def test_explicit_defect_rows_with_per_node_steps(gpu, name):
    """opts['integration'] = 'explicit' with a step per node and instance (dt_k = 1 / progress_k^2, control/base.py:276):
    r_k = x_{k+1} - F(x_k, u_k, dt_k) written by the step kernel itself (ac_shoot_defect_f32), against the oracle's step on
    every (node, instance) pair; and the same rows through the implicit entry point with the same per-node steps."""
    import torch
    from aircraft_amd.control import MultipleShooting

    ac = build(name)
    B, H = 33, 7
    X0, U = near_trim_problem(B, H, seed=9)
    ms = MultipleShooting(system=ac, dt=0.01, num_nodes=H, opts={"quaternion": "constraint"})
    Ud = dev(U, gpu)
    X = ms.rollout(dev(X0, gpu), Ud)
    X = (X + 1e-3 * torch.randn_like(X)).contiguous()      # off the rollout: the rows are not zero
    dt = np.random.default_rng(4).uniform(4e-3, 1.2e-2, (H, B))
    dtd = dev(dt, gpu)
    r = ms.defects(X, Ud, dt=dtd, integration="explicit").cpu().numpy()
    Xh, Uh, dth = X.cpu().numpy().astype(np.float64), f32_exact(U), f32_exact(dt)
    orc = make_oracle(ac)
    flat = lambda a, rows: np.ascontiguousarray(a.transpose(1, 0, 2).reshape(rows, H * B))  # noqa: E731
    Fr = orc.state_update(flat(Xh[:-1], 13), flat(Uh, 7), dth.reshape(-1)).reshape(13, H, B).transpose(1, 0, 2)
    scale = np.maximum(np.abs(Xh[1:]), 1.0).max()
    assert np.abs(r - (Xh[1:] - Fr)).max() <= 1e-5 * scale
    assert np.abs(r).max() > 1e-4
    ri = ms.defects(X, Ud, dt=dtd, integration="implicit").cpu().numpy()
    fr = orc.state_derivative(flat(Xh[1:], 13), flat(Uh, 7)).reshape(13, H, B).transpose(1, 0, 2)
    assert np.abs(ri - ((Xh[1:] - Xh[:-1]) - dth[:, None, :] * fr)).max() <= 1e-5 * scale
    r2, Jn, Ju, jdt = ms.linearise_implicit(X, Ud, dt=dtd)
    assert float(np.abs(r2.cpu().numpy() - ri).max()) <= 2e-6 * scale
    # (f on states 1e-3 off a rollout, block-relative with the omega_dot floor of 0.1: the derivative's own parity is test_state_derivative's)
    assert block_rel_err(-jdt.cpu().numpy(), fr) < 1e-4
