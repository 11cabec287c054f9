from .denoisers import Denoiser, MMDiT, ModelInput, ModelOutput, SprintDiT
from .embedders import ContextEmbedder, PrecomputedEmbedder

__all__ = ["Denoiser", "MMDiT", "ModelInput", "ModelOutput", "SprintDiT", "ContextEmbedder", "PrecomputedEmbedder"]
