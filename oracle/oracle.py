"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; aircraft_amd never does.  See oracle/aircraft_oracle.h for what the oracle is and
how it is pinned to the reference (simulation.h5 replay + ScaledModel golden vectors).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libaircraft_oracle.so")

MODEL_KINDS = {"default": 0, "linear": 1, "nn": 2, "poly": 3, "quad": 4}
MAX_LAYERS = 8


class _Params(C.Structure):
    _fields_ = [
        ("mass", C.c_double), ("S", C.c_double), ("b", C.c_double), ("c", C.c_double),
        ("Ixx", C.c_double), ("Iyy", C.c_double), ("Izz", C.c_double), ("Ixz", C.c_double),
        ("com", C.c_double * 3),
        ("rudder_moment_arm", C.c_double),
        ("epsilon", C.c_double),
        ("gravity", C.c_double * 3),
        ("substeps", C.c_int), ("normalise", C.c_int), ("stall_scaling", C.c_int), ("model_kind", C.c_int),
        ("linear_W", C.c_double * 36),
        ("poly_coef", C.c_double * (6 * 34)),
        ("poly_intercept", C.c_double * 6),
        ("mlp_n_layers", C.c_int),
        ("mlp_widths", C.c_int * (MAX_LAYERS + 1)),
        ("mlp_act", C.c_int * MAX_LAYERS),
        ("mlp_W", C.POINTER(C.c_double) * MAX_LAYERS),
        ("mlp_b", C.POINTER(C.c_double) * MAX_LAYERS),
        ("mlp_in_mean", C.c_double * 5), ("mlp_in_std", C.c_double * 5),
        ("mlp_out_mean", C.c_double * 6), ("mlp_out_std", C.c_double * 6),
    ]


def build(force: bool = False) -> str:
    """Compile oracle/libaircraft_oracle.so with the committed Makefile (g++)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
        os.path.join(_HERE, "aircraft_oracle.cpp")
    ):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        pp = C.POINTER(_Params)
        L.oracle_state_derivative_f64.argtypes = [pp, dp, dp, C.c_long, dp]
        L.oracle_step_f64.argtypes = [pp, dp, dp, dp, C.c_int, C.c_long, dp]
        L.oracle_rollout_f64.argtypes = [pp, dp, dp, C.c_double, C.c_long, C.c_long, dp]
        L.oracle_step_sens_f64.argtypes = [pp, dp, dp, dp, C.c_int, C.c_long, dp, dp, dp, dp]
        L.oracle_aero_f64.argtypes = [pp, dp, dp, C.c_long, dp]
        L.oracle_mlp_f64.argtypes = [pp, dp, C.c_long, dp, dp]
        L.oracle_state_derivative_sens_f64.argtypes = [pp, dp, dp, C.c_long, dp, dp, dp]
        L.oracle_state_derivative_sens_f64.restype = C.c_int
        L.oracle_envelope_f64.argtypes = [pp, dp, C.c_long, dp, dp]
        L.oracle_envelope_f64.restype = C.c_int
        L.oracle_num_threads.restype = C.c_int
        L.oracle_set_num_threads.argtypes = [C.c_int]
        for f in ("oracle_state_derivative_f64", "oracle_step_f64", "oracle_rollout_f64", "oracle_step_sens_f64",
                  "oracle_aero_f64", "oracle_mlp_f64"):
            getattr(L, f).restype = C.c_int
        _lib = L
    return _lib


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _c64(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


class Oracle:
    """Float64 oracle for one airframe + coefficient model.

    airframe: dict with mass, reference_area, span, chord, Ixx, Iyy, Izz, Ixz, com (3), rudder_moment_arm
    model:    'default' | 'linear' | 'nn' | 'poly'
    model_data: linear -> {'W': (6,6)} ; poly -> {'coef': (6,34), 'intercept': (6,)} ;
                nn -> {'weights': [W_l (out,in)], 'biases': [...], 'act': [0/1 per layer],
                       'input_mean','input_std','output_mean','output_std'}
    """

    def __init__(self, airframe, model="default", model_data=None, *, substeps=1, normalise=False,
                 stall_scaling=False, epsilon=1e-6, gravity=(0.0, 0.0, 9.81)):
        p = _Params()
        p.mass = airframe["mass"]; p.S = airframe["reference_area"]; p.b = airframe["span"]; p.c = airframe["chord"]
        p.Ixx = airframe["Ixx"]; p.Iyy = airframe["Iyy"]; p.Izz = airframe["Izz"]; p.Ixz = airframe["Ixz"]
        p.com[:] = list(airframe["com"])
        p.rudder_moment_arm = airframe.get("rudder_moment_arm", 0.5)
        p.epsilon = epsilon
        p.gravity[:] = list(gravity)
        p.substeps = int(substeps); p.normalise = int(bool(normalise)); p.stall_scaling = int(bool(stall_scaling))
        p.model_kind = MODEL_KINDS[model]
        self._keep = []
        md = model_data or {}
        if model == "linear":
            p.linear_W[:] = list(_c64(md["W"], (6, 6)).ravel())
        elif model == "poly":
            p.poly_coef[:] = list(_c64(md["coef"], (6, 34)).ravel())
            p.poly_intercept[:] = list(_c64(md["intercept"], (6,)))
        elif model == "nn":
            Ws = [_c64(w) for w in md["weights"]]
            bs = [_c64(b) for b in md["biases"]]
            assert len(Ws) <= MAX_LAYERS
            p.mlp_n_layers = len(Ws)
            p.mlp_widths[0] = Ws[0].shape[1]
            for l, (w, b) in enumerate(zip(Ws, bs)):
                assert w.shape[1] == p.mlp_widths[l] and b.shape == (w.shape[0],)
                p.mlp_widths[l + 1] = w.shape[0]
                p.mlp_act[l] = int(md["act"][l])
                p.mlp_W[l] = _dptr(w)
                p.mlp_b[l] = _dptr(b)
            self._keep += Ws + bs
            p.mlp_in_mean[:] = list(_c64(md["input_mean"], (5,))); p.mlp_in_std[:] = list(_c64(md["input_std"], (5,)))
            p.mlp_out_mean[:] = list(_c64(md["output_mean"], (6,))); p.mlp_out_std[:] = list(_c64(md["output_std"], (6,)))
        self.p = p
        self.model = model

    # -- batched entry points; X (13,n), U (7,n) float64 ------------------------------------
    def state_derivative(self, X, U):
        X = _c64(X); U = _c64(U); n = X.shape[1]
        out = np.empty((13, n))
        assert lib().oracle_state_derivative_f64(C.byref(self.p), _dptr(X), _dptr(U), n, _dptr(out)) == 0
        return out

    def _dt(self, dt, n):
        dt = np.asarray(dt, dtype=np.float64)
        if dt.ndim == 0:
            return np.array([float(dt)]), 1
        assert dt.shape == (n,)
        return np.ascontiguousarray(dt), 0

    def state_update(self, X, U, dt):
        X = _c64(X); U = _c64(U); n = X.shape[1]
        dta, sc = self._dt(dt, n)
        out = np.empty((13, n))
        assert lib().oracle_step_f64(C.byref(self.p), _dptr(X), _dptr(U), _dptr(dta), sc, n, _dptr(out)) == 0
        return out

    def rollout(self, X0, U, dt):
        """X0 (13,B); U (H,7,B) -> (H+1,13,B)"""
        X0 = _c64(X0); U = _c64(U); B = X0.shape[1]; H = U.shape[0]
        assert U.shape == (H, 7, B)
        out = np.empty((H + 1, 13, B))
        assert lib().oracle_rollout_f64(C.byref(self.p), _dptr(X0), _dptr(U), float(dt), B, H, _dptr(out)) == 0
        return out

    def step_sens(self, X, U, dt):
        X = _c64(X); U = _c64(U); n = X.shape[1]
        dta, sc = self._dt(dt, n)
        Xn = np.empty((13, n)); A = np.empty((13, 13, n)); Bm = np.empty((13, 7, n)); c = np.empty((13, n))
        assert lib().oracle_step_sens_f64(C.byref(self.p), _dptr(X), _dptr(U), _dptr(dta), sc, n, _dptr(Xn), _dptr(A),
                                          _dptr(Bm), _dptr(c)) == 0
        return Xn, A, Bm, c

    def state_derivative_sens(self, X, U):
        """(x_dot (13,n), Fx = df/dx (13,13,n), Fu = df/du (13,7,n)) by exact forward-mode AD"""
        X = _c64(X); U = _c64(U); n = X.shape[1]
        xd = np.empty((13, n)); Fx = np.empty((13, 13, n)); Fu = np.empty((13, 7, n))
        assert lib().oracle_state_derivative_sens_f64(C.byref(self.p), _dptr(X), _dptr(U), n, _dptr(xd), _dptr(Fx),
                                                      _dptr(Fu)) == 0
        return xd, Fx, Fu

    def envelope(self, X):
        """(rows (4,n) = |v_rel|^2, beta, alpha, z ; Jx (4,13,n)) — control/aircraft.py:44-59"""
        X = _c64(X); n = X.shape[1]
        rows = np.empty((4, n)); Jx = np.empty((4, 13, n))
        assert lib().oracle_envelope_f64(C.byref(self.p), _dptr(X), n, _dptr(rows), _dptr(Jx)) == 0
        return rows, Jx

    AERO_ROWS = {"v_frd_rel": slice(0, 3), "airspeed": 3, "alpha": 4, "beta": 5, "qbar": 6,
                 "coefficients": slice(7, 13), "forces_frd": slice(13, 16), "moments_frd": slice(16, 19),
                 "phi": 19, "theta": 20, "psi": 21}

    def aero(self, X, U):
        X = _c64(X); U = _c64(U); n = X.shape[1]
        out = np.empty((22, n))
        assert lib().oracle_aero_f64(C.byref(self.p), _dptr(X), _dptr(U), n, _dptr(out)) == 0
        return out

    def mlp(self, inputs, jac=True):
        inputs = _c64(inputs); n = inputs.shape[0]
        assert inputs.shape == (n, 5)
        y = np.empty((n, 6)); J = np.empty((n, 6, 5)) if jac else None
        assert lib().oracle_mlp_f64(C.byref(self.p), _dptr(inputs), n, _dptr(y), _dptr(J) if jac else None) == 0
        return (y, J) if jac else y


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(n):
    lib().oracle_set_num_threads(int(n))


def for_aircraft(ac) -> "Oracle":
    """The float64 oracle for the same airframe / coefficient model / options as the product object `ac`
    (an aircraft_amd.Aircraft or Quadrotor).  Used by tests/, smoke() and bench.py's cpu_baseline leg."""
    md = ac.coefficient_model.oracle_data() if hasattr(ac, "coefficient_model") else None
    if ac.model_kind == "nn":
        md = {k: ([np.asarray(a, dtype=np.float64) for a in v] if k in ("weights", "biases") else v)
              for k, v in md.items()}
    return Oracle(ac.airframe_dict(), ac.model_kind, md, substeps=ac.physical_integration_substeps,
                  normalise=ac.normalise, stall_scaling=ac.stall_scaling, epsilon=ac.epsilon, gravity=ac.gravity)
