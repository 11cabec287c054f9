from .euler import Euler
from .euler_maruyama import EulerMaruyama

__all__ = ["Euler", "EulerMaruyama"]
