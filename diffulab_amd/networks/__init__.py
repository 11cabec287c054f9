from .denoisers import DDT, Denoiser, MMDiT, ModelInput, ModelOutput, SprintDiT
from .embedders import ContextEmbedder, PrecomputedEmbedder

__all__ = ["DDT", "Denoiser", "MMDiT", "ModelInput", "ModelOutput", "SprintDiT", "ContextEmbedder", "PrecomputedEmbedder"]
