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


def unit_max_rel(a, ref):
    """Per UNIT (last axis) relative error in the max norm:  max_entries |a - ref| / max_entries |ref|  -> (n,).
    One wrong unit in a batch shows up as one large entry (a batch-aggregated Frobenius norm would hide it)."""
    a = np.asarray(a, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    n = ref.shape[-1]
    d = np.abs(a - ref).reshape(-1, n).max(axis=0)
    return d / np.maximum(np.abs(ref).reshape(-1, n).max(axis=0), 1e-300)


def unit_rowblock_rel(a, ref, floor_frac=1e-3):
    """Per unit, the worst ROW block (the p, v, q, omega rows of a (13, m, n) Jacobian or a (13, n) vector):
    max|a - ref| over the block / max(max|ref| over the block, floor_frac * max|ref| of the whole unit)  -> (n,).
    Stricter than unit_max_rel where a row block is small next to the others (dp/du next to domega/du)."""
    a = np.asarray(a, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    n = ref.shape[-1]
    whole = np.abs(ref).reshape(-1, n).max(axis=0)
    worst = np.zeros(n)
    for sl in BLOCKS.values():
        d = np.abs(a[sl] - ref[sl]).reshape(-1, n).max(axis=0)
        den = np.maximum(np.abs(ref[sl]).reshape(-1, n).max(axis=0), floor_frac * whole)
        worst = np.maximum(worst, d / np.maximum(den, 1e-300))
    return worst


def parity_report(name, **kv):
    """Append one JSON line of measured parity figures (worst / p50 / p99 errors, checked fractions) to
    gpurun_out/parity_report.jsonl, so the numbers behind the assertions travel back from the GPU box."""
    import json

    path = os.environ.get("AIRCRAFT_PARITY_REPORT", os.path.join(ROOT, "gpurun_out", "parity_report.jsonl"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps({"test": name, **{k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in kv.items()}}) + "\n")
    except OSError:
        pass


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


def instance_err(out, ref):
    """Per instance (last axis): worst block-relative deviation of `out` from `ref` over all leading axes -> (B,)."""
    with np.errstate(all="ignore"):
        return _block_worst(np.asarray(out, dtype=np.float64), np.asarray(ref, dtype=np.float64))


def conditioning(orc, X0, U, dt, eps=1e-7, draws=3, seed=0, rollout=True):
    """How far the float64 REFERENCE itself moves when x0 is perturbed by a relative `eps` (about one fp32 ulp; random
    signs, worst of `draws` draws), per instance and block-relative at every node: (reference output, deviation (B,)).
    An fp32 evaluation commits roundings of that size at every operation, so no fp32 arithmetic can be expected to
    agree with the reference much better than this deviation on an instance whose dynamics amplify perturbations
    (tumbling trajectories; RK4 outside its stability region)."""
    rng = np.random.default_rng(seed)
    f = (lambda x: orc.rollout(x, U, dt)) if rollout else (lambda x: orc.state_update(x, U, dt))
    with np.errstate(all="ignore"):
        ref = f(X0)
        dev = np.zeros(ref.shape[-1])
        for _ in range(draws):
            pert = f(X0 * (1.0 + eps * rng.choice([-1.0, 1.0], X0.shape)))
            dev = np.maximum(dev, np.nan_to_num(_block_worst(pert, ref), nan=np.inf))
    return ref, dev


def check_against_conditioning(name, out, ref, dev, tol, factor=8.0, min_frac=None):
    """Assert on EVERY instance with a finite (and bounded, < 1e6) reference:  err <= max(tol, factor * dev)  — the result is within the
    stated tolerance, or within `factor` times what a one-ulp input perturbation does to the float64 reference itself —
    and report how many instances meet the plain tolerance (all-instance p50 / p99 / worst).  Nothing is masked."""
    err = instance_err(out, ref)
    # instances whose float64 reference is finite AND bounded: a trajectory that has blown up past 1e6 (m, m/s, rad/s)
    # overflows fp32 long before float64 and carries no information about parity
    with np.errstate(all="ignore"):
        r2 = np.asarray(ref, dtype=np.float64).reshape(-1, ref.shape[-1])
        fin = np.isfinite(r2).all(axis=0) & (np.abs(np.nan_to_num(r2, nan=np.inf)).max(axis=0) < 1e6)
    err_f = np.where(np.isfinite(err), err, np.inf)[fin]
    dev_f = dev[fin]
    bound = np.maximum(tol, factor * dev_f)
    frac_tol = float((err_f <= tol).mean()) if fin.any() else 0.0
    with np.errstate(all="ignore"):
        ratio = np.where(dev_f > 0, err_f / np.maximum(dev_f, 1e-300), 0.0)
    parity_report(name, instances=int(ref.shape[-1]), finite_reference=int(fin.sum()), frac_within_tol=frac_tol, tol=tol,
                  err_p50=float(np.median(err_f)), err_p99=float(np.quantile(err_f, 0.99)), err_max=float(err_f.max()),
                  frac_dev_below_tol=float((dev_f <= tol).mean()), worst_err_over_dev=float(np.max(ratio[err_f > tol])) if (err_f > tol).any() else 0.0,
                  violations=int((err_f > bound).sum()))
    # a non-finite GPU result where the float64 reference is finite only passes if the reference is hopelessly conditioned
    assert (err_f <= bound).all(), (name, "instances beyond max(tol, factor x conditioning):", int((err_f > bound).sum()),
                                    "worst err", float(err_f.max()))
    well = dev_f <= tol / factor  # instances any fp32 evaluation can be expected to get right
    if well.any():
        assert (err_f[well] <= tol).all(), (name, float(err_f[well].max()))
    if min_frac is not None:
        assert frac_tol >= min_frac, (name, "fraction within tol", frac_tol, "<", min_frac)
    return err, frac_tol


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
