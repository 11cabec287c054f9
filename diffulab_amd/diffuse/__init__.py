from .diffuser import Diffuser
from .modelizations import Diffusion, Flow, GaussianDiffusion
from .utils import SamplingOutput

__all__ = ["Diffuser", "Diffusion", "Flow", "GaussianDiffusion", "SamplingOutput"]
