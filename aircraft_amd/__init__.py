"""aircraft_amd — MI355X-native implementation of AIrcraft's MPC-rollout hot path.

The package mirrors the reference's plugin surface for that path only
(`Aircraft(AircraftOpts)`, `COEFF_MODEL_REGISTRY`, `state_update`, `state_derivative`, rollout
`initialise()`); the arithmetic runs in hand-written HIP kernels behind the C ABI of
include/aircraft_hip.h.  PyTorch is used for device buffers, streams and torch.distributed only.
"""
from .utils import AircraftConfiguration, TrajectoryConfiguration, MlpData, load_model, load_poly, load_linear
from .dynamics.base import SixDOF, SixDOFOpts, BatchedFunction
from .dynamics.aircraft import Aircraft, AircraftOpts
from .dynamics.quadrotor import Quadrotor
from .dynamics.coefficient_models import (COEFF_MODEL_REGISTRY, CoefficientModel, DefaultModel, LinearModel,
                                          NeuralModel, PolynomialModel)
from ._lib import AircraftHipError
from .trajectory_io import TrajectoryData, load_trajectory, save_trajectory

__all__ = [
    "AircraftConfiguration", "TrajectoryConfiguration", "MlpData", "load_model", "load_poly", "load_linear",
    "SixDOF", "SixDOFOpts", "BatchedFunction", "Aircraft", "AircraftOpts", "Quadrotor", "COEFF_MODEL_REGISTRY",
    "CoefficientModel", "DefaultModel", "LinearModel", "NeuralModel", "PolynomialModel", "AircraftHipError",
    "TrajectoryData", "load_trajectory", "save_trajectory",
]
__version__ = "0.1.0"
