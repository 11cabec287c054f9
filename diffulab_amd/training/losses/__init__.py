from .common import LossFunction
from .repa import RepaLoss

__all__ = ["LossFunction", "RepaLoss"]
