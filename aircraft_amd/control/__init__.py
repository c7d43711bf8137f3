from .base import MultipleShooting
from .ilqr import ILQR, QuadraticCost

__all__ = ["MultipleShooting", "ILQR", "QuadraticCost"]
