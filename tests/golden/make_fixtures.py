#!/usr/bin/env python3
"""Regenerate the golden fixtures under tests/golden/ from the reference checkout.

Runs ONLY in the build container (needs /root/reference); the GPU box and the
test-suite use the committed .npz/.json outputs.  Everything written here is
*data* (inputs, expected outputs, fitted coefficients), never reference source.

Sources (all under /root/reference):
  data/trajectories/simulation.h5       stored rollout written by main/dynamics/dynamics.py:134-145
                                        (contiguous little-endian f64; offsets from SURVEY.md App. B)
  data/networks/fitted_models_casadi.pkl  sklearn cubic fits used by PolynomialModel
                                        (dynamics/coefficient_models.py:106-133)
  data/networks/linearised.csv          LinearModel matrix (coefficient_models.py:80-89)
  data/networks/model-dynamics.pth      ScaledModel checkpoint (utils.py:22-40)
  data/glider/problem_definition.json   airframe constants (lines 12-24)
  src/aircraft/surrogates/models.py     ScaledModel is IMPORTED (torch only) to produce
                                        forward values + autograd Jacobians = golden vectors
"""
import json
import os
import pickle
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def decode_simulation_h5():
    raw = open(f"{REF}/data/trajectories/simulation.h5", "rb").read()
    assert len(raw) == 11200
    state = np.frombuffer(raw, dtype="<f8", count=13 * 40, offset=2432).reshape(13, 40)
    control = np.frombuffer(raw, dtype="<f8", count=7 * 40, offset=8640).reshape(7, 40)
    times = np.frombuffer(raw, dtype="<f8", count=40, offset=10880)
    assert np.allclose(times, 0.1 * np.arange(40))
    assert np.allclose(control[1], 3.0)
    np.savez(f"{OUT}/simulation_h5.npz", state=state, control=control, times=times)
    print("simulation_h5.npz", state.shape, control.shape)
    # the data file itself (11 KB) is the fixture for aircraft_amd.trajectory_io's reader; when an HDF5 C library is
    # present, cross-check the raw-offset decode above against a real HDF5 read
    import shutil
    shutil.copyfile(f"{REF}/data/trajectories/simulation.h5", f"{OUT}/simulation.h5")
    os.chmod(f"{OUT}/simulation.h5", 0o644)
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from aircraft_amd import trajectory_io
    if trajectory_io.hdf5_available():
        t = trajectory_io.load_trajectory(f"{OUT}/simulation.h5", 0)
        assert np.array_equal(t.state, state) and np.array_equal(t.control, control) and np.array_equal(t.times, times)


class _Inert:
    """Stand-in for sklearn / casadi classes: stores state, executes nothing."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, st):
        self.__dict__["_state"] = st


# exactly the numpy globals an array / scalar pickle needs — never the `numpy` package as a whole
# (numpy.testing._private.utils.runstring and friends would execute code); same list as aircraft_amd/utils.py
_NUMPY_PICKLE_GLOBALS = {
    ("numpy._core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "_reconstruct"),
    ("numpy._core.multiarray", "scalar"), ("numpy.core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
}


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _NUMPY_PICKLE_GLOBALS:
            import importlib

            return getattr(importlib.import_module(module), name)
        if module.split(".")[0] in ("sklearn", "casadi"):
            return type(name, (_Inert,), {})
        raise pickle.UnpicklingError(f"blocked global {module}.{name}")


def decode_poly():
    with open(f"{REF}/data/networks/fitted_models_casadi.pkl", "rb") as f:
        d = _RestrictedUnpickler(f).load()
    keys = ["CX", "CY", "CZ", "Cl", "Cm", "Cn"]
    assert list(d["fitted_models"].keys()) == keys
    coef = np.stack([np.asarray(d["fitted_models"][k]["coef"], dtype=np.float64) for k in keys])
    intercept = np.array([float(d["fitted_models"][k]["intercept"]) for k in keys])
    assert coef.shape == (6, 34)
    st = d["fitted_models"]["CX"]["poly"]._state
    assert st["degree"] == 3 and st["include_bias"] is False and st["n_features_in_"] == 4
    np.savez(f"{OUT}/poly_coef.npz", coef=coef, intercept=intercept)
    print("poly_coef.npz", coef.shape, intercept)


def decode_linear():
    W = np.loadtxt(f"{REF}/data/networks/linearised.csv", delimiter=",", skiprows=1)
    assert W.shape == (6, 6)
    np.savez(f"{OUT}/linearised.npz", W=W)
    print("linearised.npz", W.shape)


def decode_params():
    pd = json.load(open(f"{REF}/data/glider/problem_definition.json"))
    out = {
        "aircraft": pd["aircraft"],
        # com override used by every driver (main/control/control.py:169-172, main/dynamics/dynamics.py:62,71)
        "com_override": [0.0131991, -1.78875e-08, 0.00313384],
        "rudder_moment_arm_default": 0.5,  # utils.py:215
        "epsilon": 1e-6,  # dynamics/base.py:11
        "gravity": [0.0, 0.0, 9.81],  # dynamics/base.py:13
    }
    json.dump(out, open(f"{OUT}/aircraft_params.json", "w"), indent=1)
    print("aircraft_params.json")


def scaledmodel_golden():
    import torch

    sys.dont_write_bytecode = True
    sys.path.insert(0, f"{REF}/src")
    from aircraft.surrogates.models import ScaledModel  # reference code, imported not copied

    ck = torch.load(f"{REF}/data/networks/model-dynamics.pth", map_location="cpu", weights_only=True)
    scaler = (ck["input_mean"], ck["input_std"], ck["output_mean"], ck["output_std"])
    model = ScaledModel(5, 6, scaler=scaler)
    model.load_state_dict(ck["model_state_dict"])
    model.eval()

    sd = ck["model_state_dict"]
    np.savez(
        f"{OUT}/scaledmodel_weights.npz",
        W0=sd["core_layers.0.weight"].numpy(), b0=sd["core_layers.0.bias"].numpy(),
        W1=sd["core_layers.1.weight"].numpy(), b1=sd["core_layers.1.bias"].numpy(),
        W2=sd["core_layers.3.weight"].numpy(), b2=sd["core_layers.3.bias"].numpy(),
        input_mean=ck["input_mean"].numpy(), input_std=ck["input_std"].numpy(),
        output_mean=ck["output_mean"].numpy(), output_std=ck["output_std"].numpy(),
    )

    rng = np.random.default_rng(42)  # config.py:5
    n = 64
    x = np.stack(
        [
            rng.uniform(300.0, 4000.0, n),  # qbar
            rng.uniform(-0.25, 0.25, n),  # alpha [rad]
            rng.uniform(-0.15, 0.15, n),  # beta [rad]
            rng.uniform(-5.0, 5.0, n),  # aileron [deg]
            rng.uniform(-5.0, 5.0, n),  # elevator [deg]
        ],
        axis=1,
    ).astype(np.float32)
    x[0] = [1531.25, 0.05, -0.02, 1.0, 3.0]  # SURVEY.md §8c known answer
    xt = torch.from_numpy(x)
    with torch.no_grad():
        y32 = model(xt).numpy()
        y64 = model.double()(xt.double()).numpy()
    model.float()
    jac32 = np.stack([torch.autograd.functional.jacobian(model, xt[i]).numpy() for i in range(n)])
    model.double()
    jac64 = np.stack([torch.autograd.functional.jacobian(model, xt[i].double()).numpy() for i in range(n)])
    np.savez(f"{OUT}/scaledmodel_golden.npz", x=x, y_f32=y32, y_f64=y64, jac_f32=jac32, jac_f64=jac64)
    print("scaledmodel_golden.npz", x.shape, y32.shape, jac32.shape)
    print("known answer:", y32[0])


def dubins_track():
    """The sampled 3-D Dubins path of the reference's own problem (data/glider/problem_definition.json: initial state,
    four waypoints, r_min) — the geometry `DubinsInitialiser` hands to MHTT as its track (control/initialisation.py:573-598).
    The Dubins construction and its sampling are the REFERENCE's code, imported (aircraft.dubins has no casadi dependency);
    the waypoint headings / pitches around it restate setup_waypoints_3d (:350-410) and the per-segment sampling rule restates
    generate_3d_dubins_path_native (:412-470), because that module imports casadi and cannot be imported here.
    What this pins: the INPUTS of the track functions (the path points).  The CasADi evaluation of the Hermite interpolant
    over them (:782-851) still cannot run here: oracle/track_oracle.py restates it."""
    import math

    sys.dont_write_bytecode = True
    sys.path.insert(0, f"{REF}/src")
    from aircraft.dubins.dubins3d import DubinsManeuver3D_constructor, compute_sampling  # reference code, imported

    pd = json.load(open(f"{REF}/data/glider/problem_definition.json"))
    wps = [list(map(float, w)) for w in pd["waypoints"]["waypoints"]]
    x0 = [float(v) for v in pd["waypoints"]["initial_state"]]
    r_min = float(pd["aircraft"]["r_min"])
    lim = [-math.pi / 2, math.pi / 2]  # DubinsInitialiser's default pitch limits (:580)

    def aim(a, b):  # heading and pitch of the straight line a -> b, pitch clipped to the limits
        dx, dy, dz = b[0] - a[0], b[1] - a[1], b[2] - a[2]
        return math.atan2(dy, dx), math.atan2(dz, math.hypot(dx, dy))

    # configuration (x, y, z, heading, pitch) per waypoint: the start aims at waypoint 1 from the initial position (the
    # reference replaces waypoint 0 by the initial position), inner waypoints aim at their successor, the last keeps the
    # direction of the one before it
    hd, pt = aim(x0[:3], wps[1])
    conf = [[x0[0], x0[1], x0[2], hd, pt]]
    for i in range(1, len(wps)):
        if i < len(wps) - 1:
            hd, pt = aim(wps[i], wps[i + 1])
            pt = min(max(pt, lim[0]), lim[1])
        else:
            hd, pt = conf[-1][3], conf[-1][4]
        conf.append([wps[i][0], wps[i][1], wps[i][2], hd, pt])
    pts, seg_len = [], []
    for qi, qf in zip(conf[:-1], conf[1:]):
        man = DubinsManeuver3D_constructor(qi, qf, r_min, lim)
        ns = max(50, int(man.length / 2.0))
        pts += [[float(p[0]), float(p[1]), float(p[2])] for p in compute_sampling(man, ns)]
        seg_len.append(float(man.length))
    pts = np.asarray(pts, dtype=np.float64)
    np.savez(f"{OUT}/dubins_track.npz", points=pts, configurations=np.asarray(conf), segment_lengths=np.asarray(seg_len),
             r_min=r_min, initial_state=np.asarray(x0), default_velocity=float(pd["waypoints"]["default_velocity"]))
    print("dubins_track.npz", pts.shape, "segment lengths", seg_len)


if __name__ == "__main__":
    decode_simulation_h5()
    decode_poly()
    decode_linear()
    decode_params()
    scaledmodel_golden()
    dubins_track()
