from .common import Denoiser, ModelInput, ModelOutput
from .mmdit import MMDiT

__all__ = ["Denoiser", "MMDiT", "ModelInput", "ModelOutput"]
