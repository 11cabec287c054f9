"""Oracle: REPA alignment loss in plain torch.  TEST INFRASTRUCTURE ONLY.

Reference: training/losses/repa.py:96-102 (3-layer SiLU projection MLP on the hooked block output) and :190-198
(``coeff * (1 - cosine_similarity(proj(src), dst, dim=-1).mean())``).  Parameters keyed like the reference ``state_dict``
(``proj.{0,2,4}.{weight,bias}``).
"""

from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor

from .dit import silu


def param_shapes(denoiser_dimension: int, hidden_dim: int, embedding_dim: int) -> dict[str, tuple[int, ...]]:
    return {"proj.0.weight": (hidden_dim, denoiser_dimension), "proj.0.bias": (hidden_dim,),
            "proj.2.weight": (hidden_dim, hidden_dim), "proj.2.bias": (hidden_dim,),
            "proj.4.weight": (embedding_dim, hidden_dim), "proj.4.bias": (embedding_dim,)}


def repa_loss(P: dict[str, Tensor], src_features: Tensor, dst_features: Tensor, coeff: float = 1.0) -> Tensor:
    h = silu(src_features @ P["proj.0.weight"].t() + P["proj.0.bias"])
    h = silu(h @ P["proj.2.weight"].t() + P["proj.2.bias"])
    proj = h @ P["proj.4.weight"].t() + P["proj.4.bias"]
    cos = F.cosine_similarity(proj, dst_features, dim=-1)
    return coeff * (1 - cos.mean())
