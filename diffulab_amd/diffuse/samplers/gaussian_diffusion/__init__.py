from .ddim import DDIM
from .ddpm import DDPM

__all__ = ["DDIM", "DDPM"]
