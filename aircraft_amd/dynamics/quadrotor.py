"""The reference's second `SixDOF` plugin: `Quadrotor` ("RPG Time Optimal Quadrotor Simplification for testing",
src/aircraft/dynamics/quadrotor.py:8-54) on the same kernels.

    mass 1, inertia = I3, com = 0, one RK4 sub-step, four controls = rotor thrusts T0..T3
    forces_frd  = (0, 0, sum T)                                                  quadrotor.py:43-45
    moments_frd = (T0-T1-T2+T3, -T0-T1+T2+T3, (T0-T1+T2-T3)/2) + com x F          quadrotor.py:48-54, base.py:253-278

The rigid body, RK4, quaternion handling, sensitivities and getters are the shared SixDOF path; only the
force/moment source differs (`AC_MODEL_QUAD`).  Device control buffers keep the common 7-row layout: the thrusts are
rows 0-3, rows 4-6 are ignored.  `state_update(x, u, dt)` takes u as (4, n).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .. import _lib
from .base import SixDOF, SixDOFOpts


class Quadrotor(SixDOF):
    num_controls = 4

    def __init__(self, *, opts: Optional[SixDOFOpts] = None, device=None) -> None:
        super().__init__(opts=opts or SixDOFOpts(physical_integration_substeps=1, mass=1.0), device=device)
        self.mass = 1.0
        self.com = np.zeros(3)
        self.physical_integration_substeps = 1
        self.model_kind = "quad"

    @property
    def inertia_tensor(self) -> np.ndarray:
        """Inertia tensor about the centre of mass (quadrotor.py:20-28)."""
        return np.eye(3)

    def _param_struct(self) -> "_lib.AcParams":
        p = _lib.AcParams()
        p.mass, p.S, p.b, p.c = float(self.mass), 1.0, 1.0, 1.0
        I = self.inertia_tensor
        p.inertia[:] = [float(v) for v in I.ravel()]
        p.inertia_inv[:] = [float(v) for v in np.linalg.inv(I).ravel()]
        p.com[:] = [float(v) for v in np.asarray(self.com, dtype=np.float64).ravel()]
        p.rudder_moment_arm = 0.0
        p.epsilon = float(self.epsilon)
        p.gravity[:] = [float(g) for g in self.gravity]
        p.substeps = int(self.physical_integration_substeps)
        p.normalise = int(bool(self.normalise))
        p.stall_scaling = 0
        p.model_kind = _lib.MODEL_KINDS["quad"]
        return p

    def _install_model(self) -> None:
        return None

    # what the test oracle needs to rebuild the same body (tests only)
    def airframe_dict(self) -> dict:
        I = self.inertia_tensor
        return {"mass": float(self.mass), "reference_area": 1.0, "span": 1.0, "chord": 1.0, "Ixx": float(I[0, 0]),
                "Iyy": float(I[1, 1]), "Izz": float(I[2, 2]), "Ixz": float(I[0, 2]),
                "com": [float(v) for v in np.asarray(self.com).ravel()], "rudder_moment_arm": 0.0}
