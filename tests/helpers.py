"""Shared test helpers: build an `Aircraft` (product) and the matching float64 oracle (checker)."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

from aircraft_amd import Aircraft, AircraftConfiguration, AircraftOpts, MlpData  # noqa: E402
from aircraft_amd.synthetic import GLIDER, near_trim_problem, synthetic_problem, synthetic_units  # noqa: E402,F401


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def f32_exact(a):
    """Round to float32 and return as float64: the SAME numbers then go to the GPU (fp32) and to the oracle."""
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def model_path(model, hidden=None):
    if model == "poly":
        return os.path.join(GOLDEN, "poly_coef.npz")
    if model == "linear":
        return os.path.join(GOLDEN, "linearised.npz")
    if model == "nn":
        if hidden is None:
            return os.path.join(GOLDEN, "scaledmodel_weights_product.npz")
        return MlpData.synthetic(hidden, seed=42)
    return ""


def make_aircraft(model="default", *, hidden=None, substeps=1, normalise=False, stall_scaling=False,
                  use_mfma=True, airframe=None) -> Aircraft:
    cfg = AircraftConfiguration(dict(airframe or GLIDER))
    path = model_path(model, hidden)
    if model == "nn" and hidden is None:
        w = golden("scaledmodel_weights.npz")
        path = MlpData([w["W0"], w["W1"], w["W2"]], [w["b0"], w["b1"], w["b2"]], [0, 1, 0], w["input_mean"],
                       w["input_std"], w["output_mean"], w["output_std"])
    opts = AircraftOpts(coeff_model_type=model, coeff_model_path=path, aircraft_config=cfg,
                        physical_integration_substeps=substeps, stall_scaling=stall_scaling, use_mfma=use_mfma)
    ac = Aircraft(opts)
    ac.normalise = normalise
    return ac


def make_oracle(ac: Aircraft):
    """The float64 oracle for the same airframe / model / options as `ac` (oracle/oracle.py::for_aircraft)."""
    from oracle import for_aircraft

    return for_aircraft(ac)


BLOCKS = {"p": slice(0, 3), "v": slice(3, 6), "q": slice(6, 10), "w": slice(10, 13)}
FLOORS = {"p": 1.0, "v": 1.0, "q": 1.0, "w": 0.1}  # denominators never below these (m, m/s, -, rad/s)


def block_rel_err(x, ref, axis=-2):
    """max over blocks (p, v, q, omega) and over columns of  |x - ref|_inf / max(|ref|_inf, floor)  per block.
    x, ref: (..., 13, n).  This is the 'relative error on state' the 1e-5 parity bar is stated in."""
    x = np.asarray(x, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    worst = 0.0
    for name, sl in BLOCKS.items():
        d = np.abs(np.take(x, range(sl.start, sl.stop), axis=axis) - np.take(ref, range(sl.start, sl.stop), axis=axis))
        den = np.maximum(np.abs(np.take(ref, range(sl.start, sl.stop), axis=axis)).max(axis=axis), FLOORS[name])
        worst = max(worst, float((d.max(axis=axis) / den).max()))
    return worst


def rel_fro(a, ref):
    a = np.asarray(a, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    return float(np.linalg.norm(a - ref) / max(np.linalg.norm(ref), 1e-30))


def in_envelope(orc, X, U=None):
    """Boolean mask over the last axis: columns of X (..., 13, n) that satisfy the reference's flight envelope
    (control/aircraft.py:47-59): 20 <= |v_rel| <= 100 m/s, |beta| <= 10 deg, |alpha| <= 20 deg, z < 0.
    Outside it the surrogate models are extrapolating and trajectories tumble (omega of tens of rad/s), so any
    rounding difference is amplified exponentially — parity is asserted on in-envelope trajectories."""
    X = np.asarray(X, dtype=np.float64)
    lead = X.shape[:-2]
    n = X.shape[-1]
    Xf = np.moveaxis(X, -2, 0).reshape(13, -1)
    Uf = np.zeros((7, Xf.shape[1])) if U is None else np.moveaxis(np.asarray(U, dtype=np.float64), -2, 0).reshape(7, -1)
    a = orc.aero(Xf, Uf)
    vr = a[0:3]
    v2 = (vr * vr).sum(axis=0)
    ok = (v2 >= 20.0 ** 2) & (v2 <= 100.0 ** 2) & (np.abs(a[5]) <= np.deg2rad(10)) & (np.abs(a[4]) <= np.deg2rad(20))
    ok &= Xf[2] < 0
    ok &= np.isfinite(Xf).all(axis=0)
    return ok.reshape(lead + (n,))


def _block_worst(pert, ref):
    """per-instance (last axis) worst block-relative deviation over all leading axes"""
    worst = np.zeros(ref.shape[-1])
    for name, sl in BLOCKS.items():
        d = np.abs(pert[..., sl, :] - ref[..., sl, :]).max(axis=-2)
        den = np.maximum(np.abs(ref[..., sl, :]).max(axis=-2), FLOORS[name])
        r = d / den
        worst = np.maximum(worst, r.reshape(-1, r.shape[-1]).max(axis=0))
    return worst


def well_conditioned(orc, X0, U, dt, eps=1e-7, tol=1e-6, seed=0, rollout=True):
    """Instances whose REFERENCE result is itself reproducible: perturb x0 by a relative `eps` (about one fp32
    ulp) and keep the instances whose float64 oracle output moves by less than `tol` = 10 eps (block-relative, at
    every node).  On the others the dynamics (tumbling trajectories, RK4 outside its stability region for the stiff
    pitch-damping mode at high speed) amplify rounding-level differences past the 1e-5 bar in ANY arithmetic.
    Returns (mask, reference output)."""
    rng = np.random.default_rng(seed)
    f = (lambda x: orc.rollout(x, U, dt)) if rollout else (lambda x: orc.state_update(x, U, dt))
    with np.errstate(all="ignore"):
        ref = f(X0)
        pert = f(X0 * (1.0 + eps * rng.choice([-1.0, 1.0], X0.shape)))
        worst = _block_worst(pert, ref)
    fin = np.isfinite(ref).reshape(-1, ref.shape[-1]).all(axis=0)
    # ... and has not exploded (the anti-lift linear table reaches |omega| ~ 1e6 rad/s within 0.1 s)
    with np.errstate(all="ignore"):
        sane = (np.abs(ref[..., 3:6, :]).reshape(-1, ref.shape[-1]).max(axis=0) < 150.0) & \
               (np.abs(ref[..., 10:13, :]).reshape(-1, ref.shape[-1]).max(axis=0) < 20.0)
    return (worst < tol) & fin & sane, ref


def oracle_step_hessian(orc, X, U, dt, lam, h=1e-5):
    """(21, 21, n): Hessian of lam . F over z = (x, u, dt) by central differences of the oracle's EXACT float64
    Jacobians [A | B | c] (truncation ~h^2, round-off ~1e-16/h: about 1e-9 relative)."""
    n = X.shape[1]
    dtv = np.full(n, float(dt)) if np.ndim(dt) == 0 else np.asarray(dt, dtype=np.float64)

    def grad(Xq, Uq, dtq):
        _, A, Bm, c = orc.step_sens(Xq, Uq, dtq)
        J = np.concatenate([A, Bm, c[:, None, :]], axis=1)  # (13, 21, n)
        return np.einsum("in,izn->zn", lam, J)

    Hm = np.zeros((21, 21, n))
    for a in range(21):
        dX, dU, dd = np.zeros_like(X), np.zeros_like(U), np.zeros(n)
        if a < 13:
            dX[a] = h
        elif a < 20:
            dU[a - 13] = h
        else:
            dd[:] = h
        Hm[a] = (grad(X + dX, U + dU, dtv + dd) - grad(X - dX, U - dU, dtv - dd)) / (2 * h)
    return Hm
