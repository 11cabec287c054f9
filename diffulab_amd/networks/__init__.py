from .denoisers import Denoiser, MMDiT, ModelInput, ModelOutput, SprintDiT

__all__ = ["Denoiser", "MMDiT", "ModelInput", "ModelOutput", "SprintDiT"]
