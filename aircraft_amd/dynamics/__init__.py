from .base import SixDOF, SixDOFOpts, BatchedFunction
from .aircraft import Aircraft, AircraftOpts
from .coefficient_models import COEFF_MODEL_REGISTRY
