from .common import Denoiser, ModelInput, ModelOutput
from .mmdit import MMDiT
from .sprint import SprintDiT
from .unet import UNetModel

__all__ = ["Denoiser", "MMDiT", "ModelInput", "ModelOutput", "SprintDiT", "UNetModel"]
