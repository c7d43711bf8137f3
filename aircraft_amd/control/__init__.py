from .base import MultipleShooting
from .ilqr import ILQR, QuadraticCost
from .moving_horizon import RecedingHorizon

__all__ = ["MultipleShooting", "ILQR", "QuadraticCost", "RecedingHorizon"]
