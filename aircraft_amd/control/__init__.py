from .base import MultipleShooting
from .ilqr import ILQR, GoalAcquisition, QuadraticCost
from .moving_horizon import MHTT, MHTTWeights, RecedingHorizon
from .track import Track

__all__ = ["MultipleShooting", "ILQR", "QuadraticCost", "RecedingHorizon", "MHTT", "MHTTWeights", "Track"]
