from .common import ContextEmbedder, ContextEmbedderOutput
from .precomputed import PrecomputedEmbedder

__all__ = ["ContextEmbedder", "ContextEmbedderOutput", "PrecomputedEmbedder"]
