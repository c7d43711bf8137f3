"""Aerodynamic-coefficient model registry — same keys and factory signature as the reference's
`COEFF_MODEL_REGISTRY` (src/aircraft/dynamics/coefficient_models.py:24-37):

    COEFF_MODEL_REGISTRY[key](path, aircraft, **kwargs) -> CoefficientModel

Here a `CoefficientModel` is a bag of numbers that `install()`s itself into the HIP handle; the
arithmetic itself runs in the kernels (aircraft_amd/csrc/ac_dynamics.hpp, ac_mlp.hpp).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict

import numpy as np

from .. import _lib
from ..utils import MlpData, load_linear, load_model, load_poly

__all__ = ["CoefficientModel", "DefaultModel", "LinearModel", "NeuralModel", "PolynomialModel",
           "COEFF_MODEL_REGISTRY"]


class CoefficientModel:
    kind = "default"

    def install(self, handle) -> None:  # handle: ctypes void* of ac_handle
        raise NotImplementedError

    def oracle_data(self):
        """Model data in the form the test oracle takes (tests only use this)."""
        return None


class DefaultModel(CoefficientModel):
    """Analytic stability-derivative model (reference coefficient_models.py:41-78); constants live in the kernel."""
    kind = "default"

    def __init__(self, aircraft=None):
        self.aircraft = aircraft

    def install(self, handle) -> None:
        return None


class LinearModel(CoefficientModel):
    """W(6x6) . [qbar, alpha, beta, aileron, elevator, 1] (reference coefficient_models.py:80-89)."""
    kind = "linear"

    def __init__(self, coeff_path, aircraft=None):
        self.W = np.asarray(coeff_path, dtype=np.float64) if isinstance(coeff_path, np.ndarray) else load_linear(coeff_path)
        assert self.W.shape == (6, 6)
        self.aircraft = aircraft

    def install(self, handle) -> None:
        W = np.ascontiguousarray(self.W, dtype=np.float32)
        _lib.check(_lib.load().ac_set_linear(handle, W.ctypes.data_as(C.POINTER(C.c_float))), "ac_set_linear")

    def oracle_data(self):
        return {"W": self.W}


class PolynomialModel(CoefficientModel):
    """Six cubic fits in (alpha, beta, aileron, elevator) evaluated at the main / wing / elevator / rudder
    effective angles (reference coefficient_models.py:106-133)."""
    kind = "poly"

    def __init__(self, poly_path, aircraft=None):
        if isinstance(poly_path, dict):
            self.coef, self.intercept = np.asarray(poly_path["coef"], float), np.asarray(poly_path["intercept"], float)
        else:
            self.coef, self.intercept = load_poly(poly_path)
        assert self.coef.shape == (6, 34) and self.intercept.shape == (6,)
        self.aircraft = aircraft

    def install(self, handle) -> None:
        coef = np.ascontiguousarray(self.coef, dtype=np.float32)
        ic = np.ascontiguousarray(self.intercept, dtype=np.float32)
        fp = C.POINTER(C.c_float)
        _lib.check(_lib.load().ac_set_poly(handle, coef.ctypes.data_as(fp), ic.ctypes.data_as(fp)), "ac_set_poly")

    def oracle_data(self):
        return {"coef": self.coef, "intercept": self.intercept}


class NeuralModel(CoefficientModel):
    """MLP surrogate (reference coefficient_models.py:91-104 wrapping surrogates/models.py:101-155).
    `model_path` is a reference .pth checkpoint, an .npz, or an `MlpData`.  `use_mfma=False` selects the
    VALU cross-lane matmul ("MFMA off" validation baseline)."""
    kind = "nn"

    def __init__(self, model_path, aircraft=None, realtime: bool = False, use_mfma: bool = True):
        self.data = model_path if isinstance(model_path, MlpData) else load_model(model_path)
        self.aircraft = aircraft
        self.realtime = realtime  # accepted for signature parity; no first-order approximation here
        self.use_mfma = bool(use_mfma)

    def install(self, handle) -> None:
        d = self.data
        L = len(d.weights)
        fp = C.POINTER(C.c_float)
        widths = (C.c_int * (L + 1))(*d.widths)
        act = (C.c_int * L)(*d.act)
        Wp = (fp * L)(*[w.ctypes.data_as(fp) for w in d.weights])
        bp = (fp * L)(*[b.ctypes.data_as(fp) for b in d.biases])
        _lib.check(
            _lib.load().ac_set_mlp(handle, L, widths, act, Wp, bp, d.input_mean.ctypes.data_as(fp),
                                   d.input_std.ctypes.data_as(fp), d.output_mean.ctypes.data_as(fp),
                                   d.output_std.ctypes.data_as(fp), int(self.use_mfma)),
            "ac_set_mlp",
        )

    def oracle_data(self):
        return self.data.as_dict()


CoeffModelFactory = Callable[..., CoefficientModel]

COEFF_MODEL_REGISTRY: Dict[str, CoeffModelFactory] = {
    "linear": lambda path, aircraft, **kwargs: LinearModel(path, aircraft),
    "poly": lambda path, aircraft, **kwargs: PolynomialModel(path, aircraft),
    "nn": lambda path, aircraft, **kwargs: NeuralModel(path, aircraft, realtime=kwargs.get("realtime", False),
                                                      use_mfma=kwargs.get("use_mfma", True)),
    "default": lambda path, aircraft, **kwargs: DefaultModel(aircraft),
}
