from .common import Denoiser, ModelInput, ModelOutput
from .ddt import DDT
from .mmdit import MMDiT
from .sprint import SprintDiT
from .unet import UNetModel

__all__ = ["DDT", "Denoiser", "MMDiT", "ModelInput", "ModelOutput", "SprintDiT", "UNetModel"]
