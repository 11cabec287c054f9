from .diffusion import Diffusion
from .flow import Flow
from .gaussian_diffusion import GaussianDiffusion

__all__ = ["Diffusion", "Flow", "GaussianDiffusion"]
