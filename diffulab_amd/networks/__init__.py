from .denoisers import Denoiser, MMDiT, ModelInput, ModelOutput

__all__ = ["Denoiser", "MMDiT", "ModelInput", "ModelOutput"]
