"""The kernels' own arithmetic (aircraft_amd/csrc/ac_math.hpp + ac_dynamics.hpp) compiled for the host with g++
(tests/host_dyn/dyn_host.cpp, -DAC_HOST_CHECK) and checked against the float64 oracle: the fp32 forward-mode tangents of
the fused RK4 step and of f, lane group by lane group exactly as the sensitivity kernels split them, for every analytic
model and every directions-per-lane count the kernels instantiate.  No GPU: this is what lets a change to the device
math be validated in the CPU suite before it ever runs on the card.  (The GPU parity tests remain the parity tests.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.helpers import (block_rel_err, f32_exact, make_aircraft, make_oracle, synthetic_units, unit_max_rel,
                           unit_rowblock_rel)

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_dyn")
SO = os.path.join(HERE, "libdyn_host.so")
CSRC = os.path.join(os.path.dirname(HERE), "..", "aircraft_amd", "csrc")


def _lib():
    src = os.path.join(HERE, "dyn_host.cpp")
    deps = [src] + [os.path.join(CSRC, f) for f in ("ac_math.hpp", "ac_dynamics.hpp", "ac_adjoint.hpp")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        # -ffp-contract=off: the tolerance below then holds for the least favourable (unfused) rounding
        subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", SO, src], check=True)
    L = C.CDLL(SO)
    fp = C.POINTER(C.c_float)
    L.host_dyn_sens.restype = C.c_int
    L.host_dyn_sens.argtypes = [C.c_void_p, fp, fp, fp, C.c_int, C.c_int, fp, fp, C.c_float, C.c_long, fp, fp, fp, fp]
    return L


def _run(ac, what, N, X, U, dt):
    L = _lib()
    fp = C.POINTER(C.c_float)
    n = X.shape[1]
    p = ac._param_struct()
    d = ac.coefficient_model.oracle_data() or {}
    keep = [np.ascontiguousarray(d[k], dtype=np.float32) if k in d else None for k in ("W", "coef", "intercept")]
    ptr = [a.ctypes.data_as(fp) if a is not None else None for a in keep]
    Xf, Uf = np.ascontiguousarray(X, dtype=np.float32), np.ascontiguousarray(U, dtype=np.float32)
    Xn, A, B, c = (np.zeros(s, dtype=np.float32) for s in ((13, n), (13, 13, n), (13, 7, n), (13, n)))
    rc = L.host_dyn_sens(C.byref(p), ptr[0], ptr[1], ptr[2], what, N, Xf.ctypes.data_as(fp), Uf.ctypes.data_as(fp), dt, n,
                         Xn.ctypes.data_as(fp), A.ctypes.data_as(fp), B.ctypes.data_as(fp), c.ctypes.data_as(fp))
    assert rc == 0, rc
    return Xn, A, B, c


def _units(n, seed):
    X, U = synthetic_units(n, seed=seed, flaps=True)
    return f32_exact(X), f32_exact(U)


@pytest.mark.parametrize("N", [2, 4, 8])
@pytest.mark.parametrize("normalise", [True, False])
@pytest.mark.parametrize("model", ["default", "linear", "poly"])
def test_device_math_step_sens_on_host(model, normalise, N):
    ac = make_aircraft(model, normalise=normalise, stall_scaling=(model == "default" and not normalise))
    X, U = _units(300, seed=21)
    Xn, A, B, c = _run(ac, 0, N, X, U, 0.01)
    Xr, Ar, Br, cr = make_oracle(ac).step_sens(X, U, 0.01)
    assert block_rel_err(Xn, Xr) < 1e-5
    for got, want in ((A, Ar), (B, Br), (c, cr)):
        assert unit_max_rel(got, want).max() < 1e-5
        assert unit_rowblock_rel(got, want).max() < 5e-5
    assert np.array_equal(A[:, :3, :], np.broadcast_to(np.eye(13, dtype=np.float32)[:, :3, None], A[:, :3, :].shape))
    assert not B[:, 3:6, :].any()


@pytest.mark.parametrize("N", [2, 4])
@pytest.mark.parametrize("model", ["default", "linear", "poly"])
def test_device_math_derivative_sens_on_host(model, N):
    ac = make_aircraft(model, stall_scaling=(model == "linear"))
    X, U = _units(300, seed=22)
    Xd, Fx, Fu, _ = _run(ac, 1, N, X, U, 0.0)
    Xr, Fxr, Fur = make_oracle(ac).state_derivative_sens(X, U)
    assert unit_max_rel(Xd, Xr).max() < 1e-5
    assert unit_max_rel(Fx, Fxr).max() < 1e-5
    assert unit_max_rel(Fu, Fur).max() < 1e-5


def _run_adjoint(ac, N, X, U, dt, lam):
    L = _lib()
    fp = C.POINTER(C.c_float)
    L.host_dyn_adjoint.restype = C.c_int
    L.host_dyn_adjoint.argtypes = [C.c_void_p, fp, fp, fp, C.c_int, fp, fp, C.c_float, fp, C.c_long, fp, fp]
    n = X.shape[1]
    p = ac._param_struct()
    d = ac.coefficient_model.oracle_data() or {}
    keep = [np.ascontiguousarray(d[k], dtype=np.float32) if k in d else None for k in ("W", "coef", "intercept")]
    ptr = [a.ctypes.data_as(fp) if a is not None else None for a in keep]
    Xf, Uf, Lf = (np.ascontiguousarray(a, dtype=np.float32) for a in (X, U, lam))
    grad, Hm = np.zeros((21, n), dtype=np.float32), np.zeros((21, 21, n), dtype=np.float32)
    rc = L.host_dyn_adjoint(C.byref(p), ptr[0], ptr[1], ptr[2], N, Xf.ctypes.data_as(fp),
                            Uf.ctypes.data_as(fp), dt, Lf.ctypes.data_as(fp), n, grad.ctypes.data_as(fp), Hm.ctypes.data_as(fp))
    assert rc == 0, rc
    return grad, Hm


@pytest.mark.parametrize("model,normalise,stall,N", [("default", True, False, 2), ("default", False, True, 4), ("linear", True, True, 2),
                                                     ("default", True, True, 1), ("poly", True, False, 2), ("poly", False, True, 1)])
def test_reverse_sweep_gradient_and_hessian_on_host(model, normalise, stall, N):
    """ac_adjoint.hpp compiled for the host: the reverse sweep through the RK4 step in plain floats gives the gradient of
    lambda . F — checked against the oracle's EXACT Jacobians, [A | B | c]' lambda — and the same sweep in duals gives the
    second-order blocks, checked against central differences of those Jacobians (tests/helpers.py::oracle_step_hessian);
    symmetric, zero rows for the position and the dead controls."""
    from tests.helpers import oracle_step_hessian

    ac = make_aircraft(model, normalise=normalise, stall_scaling=stall)
    n = 40
    X, U = _units(n, seed=31)
    lam = f32_exact(np.random.default_rng(3).normal(size=(13, n)))
    grad, Hm = _run_adjoint(ac, N, X, U, 0.01, lam)
    orc = make_oracle(ac)
    _, A, B, c = orc.step_sens(X, U, 0.01)
    J = np.concatenate([A, B, c[:, None, :]], axis=1)          # (13, 21, n)
    want = np.einsum("in,izn->zn", lam, J)
    assert unit_max_rel(grad, want).max() < 2e-5
    Hr = oracle_step_hessian(orc, X, U, 0.01, lam)
    num = np.sqrt(((Hm - Hr) ** 2).sum(axis=(0, 1))); den = np.sqrt((Hr ** 2).sum(axis=(0, 1)))
    rel = num / np.maximum(den, 1e-30)
    if rel.max() >= 1e-3:   # (the checker's step across a |.| kink of the stall factors: see tests/test_gpu_fuzz.py)
        Hr7 = oracle_step_hessian(orc, X, U, 0.01, lam, h=1e-7)
        rel = np.minimum(rel, np.sqrt(((Hm - Hr7) ** 2).sum(axis=(0, 1))) / np.maximum(np.sqrt((Hr7 ** 2).sum(axis=(0, 1))), 1e-30))
    assert rel.max() < 1e-3, (int(rel.argmax()), float(rel.max()))
    assert np.abs(Hm - Hm.transpose(1, 0, 2)).max() <= 2e-4 * np.abs(Hm).max()
    assert not Hm[:3].any() and not Hm[16:19].any()


def test_cubic_fit_value_gradient_and_second_derivative_tables():
    """The host-derived gradient and second-derivative tables of the cubic fits (ac_set_poly uploads them; the sensitivity and
    second-order kernels stream them) evaluated by the device's own row-streaming code, against the polynomial itself in
    float64: sklearn's monomial order, derivatives by the product rule."""
    import itertools
    ac = make_aircraft("poly")
    d = ac.coefficient_model.oracle_data()
    coef = np.ascontiguousarray(d["coef"], dtype=np.float32)
    icpt = np.ascontiguousarray(d["intercept"], dtype=np.float32)
    L = _lib()
    fp = C.POINTER(C.c_float)
    L.host_poly_point.restype = C.c_int
    L.host_poly_point.argtypes = [fp, fp, C.c_int, fp, C.c_long, fp]
    rng = np.random.default_rng(5)
    n = 200
    F = np.ascontiguousarray(rng.uniform(-0.4, 0.4, (4, n)), dtype=np.float32)
    combos = [c for deg in (1, 2, 3) for c in itertools.combinations_with_replacement(range(4), deg)]
    assert len(combos) == 34
    F64 = F.astype(np.float64)
    pairs = [(v, q) for v in range(4) for q in range(v, 4)]
    for k in range(6):
        out = np.zeros((15, n), dtype=np.float32)
        assert L.host_poly_point(coef.ctypes.data_as(fp), icpt.ctypes.data_as(fp), k, F.ctypes.data_as(fp), n, out.ctypes.data_as(fp)) == 0
        val = np.full(n, float(icpt[k])); g = np.zeros((4, n)); h = np.zeros((4, 4, n))
        for t, cmb in enumerate(combos):
            c = float(coef[k, t])
            val += c * np.prod([F64[i] for i in cmb], axis=0)
            for pos, v in enumerate(cmb):  # d/df_v: drop one factor
                rest = cmb[:pos] + cmb[pos + 1:]
                g[v] += c * (np.prod([F64[i] for i in rest], axis=0) if rest else 1.0)
                for pos2, q in enumerate(rest):  # d2/df_v df_q: drop another
                    rest2 = rest[:pos2] + rest[pos2 + 1:]
                    h[v, q] += c * (np.prod([F64[i] for i in rest2], axis=0) if rest2 else 1.0)
        scale = max(1.0, np.abs(coef[k]).max())
        assert np.abs(out[0] - val).max() < 2e-6 * scale
        assert np.abs(out[1:5] - g).max() < 5e-6 * scale
        for e, (v, q) in enumerate(pairs):
            assert np.abs(out[5 + e] - h[v, q]).max() < 1e-5 * scale, (k, v, q)
            assert np.abs(h[v, q] - h[q, v]).max() < 1e-12 * scale
