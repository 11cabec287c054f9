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


# ------------------------------------------------------------------------------------------------ Perceiver resampler
def resampler_param_shapes(dim: int, depth: int, head_dim: int, num_heads: int, ff_mult: int, num_latents: int) -> dict[str, tuple[int, ...]]:
    """state_dict layout of networks/repa/perceiver_resampler.py:PerceiverResampler"""
    inner, s = head_dim * num_heads, {"latents": (num_latents, dim)}
    for i in range(depth):
        a, f = f"layers.{i}.0.", f"layers.{i}.1."
        for n in ("norm_x", "norm_latents"):
            s[a + n + ".weight"], s[a + n + ".bias"] = (dim,), (dim,)
        s[a + "to_q.weight"], s[a + "to_kv.weight"], s[a + "to_out.weight"] = (inner, dim), (2 * inner, dim), (dim, inner)
        s[f + "0.weight"], s[f + "0.bias"] = (dim,), (dim,)
        s[f + "1.weight"], s[f + "3.weight"] = (int(dim * ff_mult), dim), (dim, int(dim * ff_mult))
    s["norm.weight"], s["norm.bias"] = (dim,), (dim,)
    return s


def perceiver_resampler(P: dict[str, Tensor], x: Tensor, depth: int, head_dim: int, num_heads: int, rope_base: float = 10_000.0) -> Tensor:
    """perceiver_resampler.py:172-252: learned latents refined by `depth` x (cross/self attention over [x ; latents] + GELU MLP),
    rotary embedding (2-D grid of sqrt(n) x sqrt(n) positions, axes head_dim/2 each) on the keys that come from x only."""
    from .dit import apply_rope, layer_norm, rope_tables

    B, n, _ = x.shape
    g = int(n**0.5)
    cos, sin = rope_tables(g, g, [head_dim // 2, head_dim // 2], rope_base)
    lat = P["latents"][None].expand(B, -1, -1)
    m = lat.shape[1]
    H = num_heads
    for i in range(depth):
        a, f = f"layers.{i}.0.", f"layers.{i}.1."
        xn = layer_norm(x, P[a + "norm_x.weight"], P[a + "norm_x.bias"], 1e-5)
        ln = layer_norm(lat, P[a + "norm_latents.weight"], P[a + "norm_latents.bias"], 1e-5)
        q = (ln @ P[a + "to_q.weight"].t()).reshape(B, m, H, head_dim)
        kx, vx = (xn @ P[a + "to_kv.weight"].t()).chunk(2, dim=-1)
        kl, vl = (ln @ P[a + "to_kv.weight"].t()).chunk(2, dim=-1)
        kx = apply_rope(kx.reshape(B, n, H, head_dim), cos, sin)
        k = torch.cat((kx, kl.reshape(B, m, H, head_dim)), dim=1).transpose(1, 2)
        v = torch.cat((vx.reshape(B, n, H, head_dim), vl.reshape(B, m, H, head_dim)), dim=1).transpose(1, 2)
        sim = (q.transpose(1, 2) * head_dim**-0.5) @ k.transpose(-1, -2)
        out = (torch.softmax(sim, dim=-1) @ v).transpose(1, 2).reshape(B, m, H * head_dim)
        lat = out @ P[a + "to_out.weight"].t() + lat
        h = layer_norm(lat, P[f + "0.weight"], P[f + "0.bias"], 1e-5) @ P[f + "1.weight"].t()
        lat = F.gelu(h) @ P[f + "3.weight"].t() + lat
    return layer_norm(lat, P["norm.weight"], P["norm.bias"], 1e-5)


def repa_loss_resampled(P: dict[str, Tensor], R: dict[str, Tensor], src_features: Tensor, dst_features: Tensor, coeff: float,
                        depth: int, head_dim: int, num_heads: int) -> Tensor:
    """repa.py:190-198 with use_resampler=True: proj MLP -> resampler -> cosine loss"""
    h = silu(src_features @ P["proj.0.weight"].t() + P["proj.0.bias"])
    h = silu(h @ P["proj.2.weight"].t() + P["proj.2.bias"])
    proj = h @ P["proj.4.weight"].t() + P["proj.4.bias"]
    lat = perceiver_resampler(R, proj, depth, head_dim, num_heads)
    return coeff * (1 - F.cosine_similarity(lat, dst_features, dim=-1).mean())
