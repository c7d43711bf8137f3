"""Configuration objects and model-data loaders for the hot path.

Mirrors the parts of the reference's `aircraft.utils` that the dynamics plugin surface consumes:
`AircraftConfiguration` (reference: src/aircraft/utils.py:201-215), `TrajectoryConfiguration`
(utils.py:279-347, only the sections the path reads) and `load_model` (utils.py:22-40).
"""
from __future__ import annotations

import json
import pickle
from pathlib import Path
from typing import Union

import numpy as np


class AircraftConfiguration:
    """Airframe constants; defaults as in the reference (utils.py:201-215)."""

    def __init__(self, aircraft_dict: dict):
        self.mass = aircraft_dict.get("mass", 1.0)
        self.span = aircraft_dict.get("span", 1.0)
        self.length = aircraft_dict.get("length", 1.2)
        self.chord = aircraft_dict.get("chord", 1.0)
        self.reference_area = aircraft_dict.get("reference_area", 0.238)
        self.aero_centre_offset = aircraft_dict.get("aero_centre_offset", [0.133, 0, 0.003])
        self.Ixx = aircraft_dict.get("Ixx", 0.155)
        self.Iyy = aircraft_dict.get("Iyy", 0.114)
        self.Izz = aircraft_dict.get("Izz", 0.262)
        self.Ixz = aircraft_dict.get("Ixz", 0.01)
        self.r_min = aircraft_dict.get("r_min", 10.0)
        self.glide_ratio = aircraft_dict.get("glide_ratio", 10.0)
        self.rudder_moment_arm = aircraft_dict.get("rudder_moment_arm", 0.5)


class TrajectoryConfiguration:
    """JSON problem definition (data/glider/problem_definition.json).  Only `.aircraft` is on the hot
    path; the other sections are kept as plain dicts for callers that want them."""

    def __init__(self, trajectory_dict: Union[dict, str, Path]):
        if isinstance(trajectory_dict, (str, Path)):
            with open(trajectory_dict, "r") as f:
                trajectory_dict = json.load(f)
        assert isinstance(trajectory_dict, dict)
        self.trajectory_dict = trajectory_dict
        self._aircraft = AircraftConfiguration(trajectory_dict.get("aircraft", {}))
        self.waypoints = trajectory_dict.get("waypoints", {})
        self.state = trajectory_dict.get("state", {})
        self.control = trajectory_dict.get("control", {})

    @property
    def aircraft(self) -> AircraftConfiguration:
        return self._aircraft

    def __repr__(self):
        return str(self.trajectory_dict)


# ----------------------------------------------------------------------------------------------
# Coefficient-model data
# ----------------------------------------------------------------------------------------------
class MlpData:
    """A generic tanh/identity MLP with ScaledModel-style input/output scalers
    (reference: surrogates/models.py:101-155)."""

    def __init__(self, weights, biases, act, input_mean, input_std, output_mean, output_std):
        self.weights = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        self.biases = [np.ascontiguousarray(b, dtype=np.float32) for b in biases]
        self.act = [int(a) for a in act]
        self.input_mean = np.ascontiguousarray(input_mean, dtype=np.float32)
        self.input_std = np.ascontiguousarray(input_std, dtype=np.float32)
        self.output_mean = np.ascontiguousarray(output_mean, dtype=np.float32)
        self.output_std = np.ascontiguousarray(output_std, dtype=np.float32)
        assert len(self.weights) == len(self.biases) == len(self.act) >= 1
        assert self.weights[0].shape[1] == 5 and self.weights[-1].shape[0] == 6
        for w, b in zip(self.weights, self.biases):
            assert b.shape == (w.shape[0],)
        for w0, w1 in zip(self.weights[:-1], self.weights[1:]):
            assert w1.shape[1] == w0.shape[0]

    @property
    def widths(self):
        return [self.weights[0].shape[1]] + [w.shape[0] for w in self.weights]

    def flops_forward(self) -> int:
        """F_mlp = 2 * sum n_i n_{i+1} (SURVEY.md §8d contract figure)."""
        return 2 * sum(w.shape[0] * w.shape[1] for w in self.weights)

    def as_dict(self):
        return {"weights": self.weights, "biases": self.biases, "act": self.act, "input_mean": self.input_mean,
                "input_std": self.input_std, "output_mean": self.output_mean, "output_std": self.output_std}

    @staticmethod
    def synthetic(hidden, seed=42, scaler=None):
        """Random-init tanh MLP 5-hidden...-6 with default nn.Linear initialisation, wrapped with the reference
        checkpoint's input/output scalers so the outputs land in the coefficient ranges."""
        import torch

        g = torch.Generator().manual_seed(seed)
        widths = [5] + list(hidden) + [6]
        Ws, bs = [], []
        for i in range(len(widths) - 1):
            lin = torch.nn.Linear(widths[i], widths[i + 1])
            bound = 1.0 / np.sqrt(widths[i])
            with torch.no_grad():
                lin.weight.uniform_(-bound, bound, generator=g)
                lin.bias.uniform_(-bound, bound, generator=g)
            Ws.append(lin.weight.detach().numpy().copy())
            bs.append(lin.bias.detach().numpy().copy())
        act = [1] * (len(widths) - 2) + [0]
        if scaler is None:  # values of the reference checkpoint
            scaler = (
                [1745.4, 3.7321e-3, -1.17e-19, 0.0, 7.1278e-2],
                [954.02, 0.11559, 0.12078, 1.7552, 2.8405],
                [-0.11649, 2.9e-19, -0.18418, -1.4e-18, -0.017588, 1.5e-20],
                [0.0895, 0.0332, 0.6166, 0.0391, 0.2478, 0.0057],
            )
        return MlpData(Ws, bs, act, *scaler)


def load_model(filepath: Union[str, Path]) -> MlpData:
    """Load the surrogate: a reference `.pth` checkpoint (keys model_state_dict / input_mean / ...;
    reference utils.py:22-40, layer pattern Linear-Linear-Tanh-Linear, surrogates/models.py:114-123)
    or an `.npz` with W0,b0,W1,b1,W2,b2 + scalers."""
    filepath = str(filepath)
    if filepath.endswith(".npz"):
        z = np.load(filepath)
        n = len([k for k in z.files if k.startswith("W")])
        Ws = [z[f"W{i}"] for i in range(n)]
        bs = [z[f"b{i}"] for i in range(n)]
        act = list(z["act"]) if "act" in z.files else [0, 1, 0][:n]
        return MlpData(Ws, bs, act, z["input_mean"], z["input_std"], z["output_mean"], z["output_std"])
    import torch

    ck = torch.load(filepath, map_location="cpu", weights_only=True)
    sd = ck["model_state_dict"]
    # nn.Sequential indices: Linear modules keep their position; a Tanh sits between index i and the next Linear
    idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith("core_layers.")})
    Ws = [sd[f"core_layers.{i}.weight"].numpy() for i in idx]
    bs = [sd[f"core_layers.{i}.bias"].numpy() for i in idx]
    act = [1 if (j + 1 < len(idx) and idx[j + 1] - idx[j] > 1) else 0 for j in range(len(idx))]
    return MlpData(Ws, bs, act, ck["input_mean"].numpy(), ck["input_std"].numpy(), ck["output_mean"].numpy(),
                   ck["output_std"].numpy())


class _Inert:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, st):
        self.__dict__["_state"] = st


# The only numpy globals an array / scalar pickle needs (numpy >= 2 writes numpy._core, older files numpy.core).
_NUMPY_PICKLE_GLOBALS = {
    ("numpy._core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "_reconstruct"),
    ("numpy._core.multiarray", "scalar"), ("numpy.core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
}


class _RestrictedUnpickler(pickle.Unpickler):
    """The reference's fitted_models_casadi.pkl holds sklearn and casadi objects; only the numpy
    coefficient arrays are needed.  Exactly the six numpy reconstructors an array pickle uses are
    allowed — NOT the `numpy` package as a whole (numpy.testing._private.utils.runstring, numpy.load
    with allow_pickle and others would execute code) — sklearn / casadi classes unpickle to inert
    stubs, and every other global is refused."""

    def find_class(self, module, name):
        if (module, name) in _NUMPY_PICKLE_GLOBALS:
            import importlib

            return getattr(importlib.import_module(module), name)
        root = module.split(".")[0]
        if root in ("sklearn", "casadi"):
            return type(name, (_Inert,), {})
        raise pickle.UnpicklingError(f"blocked global {module}.{name}")


POLY_KEYS = ["CX", "CY", "CZ", "Cl", "Cm", "Cn"]


def load_poly(filepath: Union[str, Path]):
    """(coef (6,34), intercept (6,)) from the reference's pickle (coefficient_models.py:106-114) or an .npz."""
    filepath = str(filepath)
    if filepath.endswith(".npz"):
        z = np.load(filepath)
        return np.asarray(z["coef"], dtype=np.float64), np.asarray(z["intercept"], dtype=np.float64)
    with open(filepath, "rb") as f:
        d = _RestrictedUnpickler(f).load()
    fm = d["fitted_models"]
    coef = np.stack([np.asarray(fm[k]["coef"], dtype=np.float64) for k in POLY_KEYS])
    intercept = np.array([float(fm[k]["intercept"]) for k in POLY_KEYS])
    if coef.shape != (6, 34):
        raise ValueError(f"expected 6x34 cubic-fit coefficients, got {coef.shape}")
    return coef, intercept


def load_linear(filepath: Union[str, Path]):
    """6x6 matrix of the LinearModel: a CSV with a header row (coefficient_models.py:82) or an .npz."""
    filepath = str(filepath)
    if filepath.endswith(".npz"):
        return np.asarray(np.load(filepath)["W"], dtype=np.float64)
    W = np.loadtxt(filepath, delimiter=",", skiprows=1)
    if W.shape != (6, 6):
        raise ValueError(f"expected a 6x6 linear coefficient table, got {W.shape}")
    return W
