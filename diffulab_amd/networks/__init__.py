from .denoisers import DDT, Denoiser, MMDiT, ModelInput, ModelOutput, SprintDiT, UNetModel
from .embedders import ContextEmbedder, PrecomputedEmbedder
from .repa import PerceiverResampler

__all__ = ["DDT", "Denoiser", "MMDiT", "ModelInput", "ModelOutput", "SprintDiT", "UNetModel", "ContextEmbedder", "PrecomputedEmbedder",
           "PerceiverResampler"]
