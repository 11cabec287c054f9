"""CPU restatement of SprintDiT with simple_dit=True (reference networks/denoisers/sprint.py) -- TEST INFRASTRUCTURE ONLY.

Encoder DiT blocks on every token -> drop a fraction of the tokens (training) -> deep DiT blocks on the kept tokens with their
RoPE rows gathered -> restore into a mask-token canvas (optionally dropping the whole deep path per sample) ->
fuse = Linear(2D -> D) on [restored ; encoder output] -> decoder DiT blocks -> modulated last layer.
Pinned by tests/golden/sprint.npz (outputs of the reference module with its random draws recorded).
"""

from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import Tensor

from . import dit as odit


@dataclass
class SprintConfig(odit.DiTConfig):
    encoder_depth: int = 2
    deep_layers_depth: int = 8
    decoder_depth: int = 2
    drop_rate: float = 0.75


def stacks(cfg: SprintConfig) -> list[tuple[str, int]]:
    return [("layers", cfg.encoder_depth), ("deep_layers", cfg.deep_layers_depth), ("decoder_layers", cfg.decoder_depth)]


def param_shapes(cfg: SprintConfig) -> dict[str, tuple[int, ...]]:
    """state_dict layout of SprintDiT(simple_dit=True) (sprint.py:96-256)"""
    D = cfg.inner_dim
    one = odit.param_shapes(odit.DiTConfig(**{**{k: getattr(cfg, k) for k in odit.DiTConfig.__dataclass_fields__}, "depth": 1}))
    shapes: dict[str, tuple[int, ...]] = {"mask_token": (1, 1, D)}
    shapes.update({k: v for k, v in one.items() if not k.startswith("layers.")})
    shapes["fuse.weight"] = (D, 2 * D)
    for name, depth in stacks(cfg):
        for i in range(depth):
            shapes.update({f"{name}.{i}." + k[len("layers.0."):]: v for k, v in one.items() if k.startswith("layers.0.")})
    return shapes


def n_kept(S: int, drop_rate: float) -> int:
    """sprint.py:342"""
    return max(1, int(S * (1.0 - float(drop_rate))))


def kept_indices(scores: Tensor, k: int) -> Tensor:
    """sprint.py:343-346: the k highest-scoring tokens of every sample, in ascending position order"""
    idx = torch.topk(scores, k=k, dim=1, largest=True, sorted=False).indices.to(torch.long)
    return torch.sort(idx, dim=1).values


def sprint_forward(P: dict[str, Tensor], x: Tensor, t: Tensor, y_eff: Tensor | None, cfg: SprintConfig,
                   kept: Tensor | None = None, path_drop: Tensor | None = None, skip_deep: bool = False,
                   taps: dict[str, Tensor] | None = None) -> Tensor:
    """sprint.py:505-573 (_forward_dit).  kept: int64 [B, k] kept token positions (None: eval mode, every token is kept);
    path_drop: bool [B] samples whose restored canvas is the mask token (restore_tokens' path_drop_p draw);
    skip_deep: the p >= 1 branch (deep layers skipped, canvas = mask token)."""
    tok, gh, gw = odit.patchify(P, x, cfg)
    B, S, D = tok.shape
    emb = odit.cond_embedding(P, t, y_eff, cfg)
    cos, sin = odit.rope_tables(gh, gw, cfg.rope_axes_dim, cfg.rope_base)
    for i in range(cfg.encoder_depth):
        tok = odit.dit_block(P, f"layers.{i}.", tok, emb, cos, sin, cfg)
        if taps is not None:
            taps[f"layer{i}"] = tok
    mask = P["mask_token"].to(tok.dtype)
    if skip_deep:
        restored = mask.expand(B, S, D)
    else:
        if kept is None:
            kept = torch.arange(S)[None].expand(B, S)
        xd = torch.gather(tok, 1, kept[..., None].expand(-1, -1, D))
        cd, sd = cos[kept], sin[kept]  # [B, k, P]
        for i in range(cfg.deep_layers_depth):
            xd = odit.dit_block(P, f"deep_layers.{i}.", xd, emb, cd, sd, cfg)
            if taps is not None:
                taps[f"deep{i}"] = xd
        restored = mask.expand(B, S, D).clone().scatter(1, kept[..., None].expand(-1, -1, D), xd)
        if path_drop is not None:
            restored = torch.where(path_drop[:, None, None], mask.expand(B, S, D), restored)
    xf = torch.cat((restored, tok), dim=-1) @ P["fuse.weight"].t()
    for i in range(cfg.decoder_depth):
        xf = odit.dit_block(P, f"decoder_layers.{i}.", xf, emb, cos, sin, cfg)
        if taps is not None:
            taps[f"decoder{i}"] = xf
    return odit.unpatchify(odit.last_layer(P, xf, emb, cfg), gh, gw, cfg)
