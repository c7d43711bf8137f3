"""Fixed-wing plugin of the batched 6-DoF model — the reference's `Aircraft(opts)` / `AircraftOpts`
(src/aircraft/dynamics/aircraft.py:22-330) on MI355X.

Same constructor options and mutable attributes (`com`, `mass`, `normalise`,
`physical_integration_substeps`, `stall_scaling`); 7 controls
[aileron, elevator, rudder (deg), thrust(3), flaps] (aircraft.py:143-166)."""
from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path
from typing import Tuple, Union

import numpy as np

from .. import _lib
from ..utils import AircraftConfiguration
from .base import SixDOF, SixDOFOpts
from .coefficient_models import COEFF_MODEL_REGISTRY, CoefficientModel, DefaultModel

__all__ = ["Aircraft", "AircraftOpts"]


def _torch_mod():
    import torch

    return torch


@dataclass
class AircraftOpts(SixDOFOpts):
    """reference dynamics/aircraft.py:22-38"""
    coeff_model_type: str = "default"  # "linear", "poly", "nn", or "default"
    coeff_model_path: Union[Path, str, object] = ""  # .csv / .pkl / .pth / .npz, or in-memory model data
    realtime: bool = False
    aircraft_config: AircraftConfiguration = field(default_factory=lambda: AircraftConfiguration({}))
    stall_angle_alpha: Tuple[float, float] = (float(np.deg2rad(-10)), float(np.deg2rad(10)))
    stall_angle_beta: Tuple[float, float] = (float(np.deg2rad(-10)), float(np.deg2rad(10)))
    stall_scaling: bool = False
    use_mfma: bool = True  # build-side: False selects the VALU matmul for the "nn" model

    def __post_init__(self):
        self.mass = self.aircraft_config.mass
        # unknown keys silently fall back to "default", as in the reference (aircraft.py:37)
        factory = COEFF_MODEL_REGISTRY.get(self.coeff_model_type, COEFF_MODEL_REGISTRY["default"])
        self.coefficient_model = lambda aircraft: factory(self.coeff_model_path, aircraft, realtime=self.realtime,
                                                          use_mfma=self.use_mfma)


def inertia_about_com(cfg_Ixx, cfg_Iyy, cfg_Izz, cfg_Ixz, mass, com):
    """I = I0 + m K(com)   (reference dynamics/aircraft.py:137-141, 168-187), float64."""
    x, y, z = (float(c) for c in com)
    I0 = np.array([[cfg_Ixx, 0.0, cfg_Ixz], [0.0, cfg_Iyy, 0.0], [cfg_Ixz, 0.0, cfg_Izz]], dtype=np.float64)
    K = np.array([[y * y + z * z, -x * y, -x * z], [-y * x, x * x + z * z, -y * z], [-z * x, -z * y, x * x + y * y]],
                 dtype=np.float64)
    return I0 + float(mass) * K


class Aircraft(SixDOF):
    num_controls = _lib.NUM_CONTROLS

    def __init__(self, opts: AircraftOpts, **kwargs):
        super().__init__(opts=opts, **kwargs)
        self.opts = opts
        self.grav = self.gravity
        self.stall_scaling = opts.stall_scaling
        self.initialise_aircraft(opts.aircraft_config)
        self.coefficient_model: CoefficientModel = (
            opts.coefficient_model(self) if opts.coefficient_model else DefaultModel(self))

    def initialise_aircraft(self, config: AircraftConfiguration) -> None:
        """reference dynamics/aircraft.py:123-141"""
        self.S = config.reference_area
        self.b = config.span
        self.c = config.chord
        self.mass = config.mass
        self.com = np.asarray(config.aero_centre_offset, dtype=np.float64)
        self.rudder_moment_arm = config.rudder_moment_arm
        self.length = config.length
        self.Ixx, self.Iyy, self.Izz, self.Ixz = config.Ixx, config.Iyy, config.Izz, config.Ixz

    @property
    def inertia_tensor(self) -> np.ndarray:
        return inertia_about_com(self.Ixx, self.Iyy, self.Izz, self.Ixz, self.mass, self.com)

    @property
    def inverse_inertia_tensor(self) -> np.ndarray:
        return np.linalg.inv(self.inertia_tensor)

    @property
    def model_kind(self) -> str:
        return self.coefficient_model.kind

    def _param_struct(self) -> "_lib.AcParams":
        p = _lib.AcParams()
        p.mass, p.S, p.b, p.c = float(self.mass), float(self.S), float(self.b), float(self.c)
        I = self.inertia_tensor
        p.inertia[:] = [float(v) for v in I.ravel()]
        p.inertia_inv[:] = [float(v) for v in np.linalg.inv(I).ravel()]
        p.com[:] = [float(v) for v in np.asarray(self.com, dtype=np.float64).ravel()]
        p.rudder_moment_arm = float(self.rudder_moment_arm)
        p.epsilon = float(self.epsilon)
        p.gravity[:] = [float(g) for g in self.gravity]
        ns = self.physical_integration_substeps
        p.substeps = int(self.opts.physical_integration_substeps if ns is None else ns)
        p.normalise = int(bool(self.normalise))
        p.stall_scaling = int(bool(self.stall_scaling))
        p.model_kind = _lib.MODEL_KINDS[self.coefficient_model.kind]
        return p

    def _install_model(self) -> None:
        self.coefficient_model.install(self._handle)

    # ---- flight envelope (reference control/aircraft.py:44-59) ------------------------------------------------------
    ENVELOPE_BOUNDS = ((20.0 ** 2, 100.0 ** 2), (-float(np.deg2rad(10)), float(np.deg2rad(10))),
                       (-float(np.deg2rad(20)), float(np.deg2rad(20))), (-float("inf"), 0.0))

    def envelope(self, x, want_jacobian: bool = True):
        """Rows of AircraftControl.state_constraint for a column batch x (13, n): rows (4, n) = (|v_rel|^2, beta, alpha, z),
        bounded by ENVELOPE_BOUNDS, and their state Jacobian Jx (4, 13, n) (None if not wanted)."""
        torch = _torch_mod()
        lib = self._sync()
        X, npx, vec = self._in(x, self.num_states, "x")
        n = X.shape[1]
        rows = torch.empty((4, n), device=X.device, dtype=torch.float32)
        Jx = torch.empty((4, 13, n), device=X.device, dtype=torch.float32) if want_jacobian else None
        _lib.check(lib.ac_envelope_f32(self._handle, X.data_ptr(), n, rows.data_ptr(),
                                       Jx.data_ptr() if Jx is not None else None, self._stream()), "ac_envelope_f32")
        if npx:
            rows = rows.cpu().numpy().astype(np.float64)
            Jx = None if Jx is None else Jx.cpu().numpy().astype(np.float64)
        if vec:
            rows = rows[..., 0]
            Jx = None if Jx is None else Jx[..., 0]
        return rows, Jx

    # what the test oracle needs to rebuild the same airframe (tests only)
    def airframe_dict(self) -> dict:
        return {"mass": float(self.mass), "reference_area": float(self.S), "span": float(self.b), "chord": float(self.c),
                "Ixx": self.Ixx, "Iyy": self.Iyy, "Izz": self.Izz, "Ixz": self.Ixz,
                "com": [float(v) for v in np.asarray(self.com).ravel()],
                "rudder_moment_arm": float(self.rudder_moment_arm)}
