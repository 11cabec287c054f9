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


# ------------------------------------------------------------------------------------------------ joint text-image form
from . import mmdit as ommdit  # noqa: E402


@dataclass
class SprintJointConfig(ommdit.JointConfig):
    encoder_depth: int = 2
    deep_layers_depth: int = 8
    n_single_stream_blocks: int = 0
    decoder_depth: int = 2
    drop_rate: float = 0.75


def joint_param_shapes(cfg: SprintJointConfig) -> dict[str, tuple[int, ...]]:
    """state_dict layout of SprintDiT(simple_dit=False) with a one-output context embedder (sprint.py:96-256)"""
    D, E = cfg.inner_dim, cfg.embedding_dim
    F = cfg.mlp_ratio * D
    base = ommdit.param_shapes(ommdit.JointConfig(**{**{k: getattr(cfg, k) for k in ommdit.JointConfig.__dataclass_fields__}, "depth": 1,
                                                     "n_single_stream_blocks": 0}))
    s: dict[str, tuple[int, ...]] = {"mask_token": (1, 1, D)}
    s.update({k: v for k, v in base.items() if not k.startswith("layers.")})
    s["fuse.weight"], s["fuse_context.weight"] = (D, 2 * D), (D, 2 * D)
    joint = {k[len("layers.0."):]: v for k, v in base.items() if k.startswith("layers.0.")}
    single = {"mlp.0.weight": (2 * F, D), "mlp.2.weight": (D, F), "attention.qkv.weight": (3 * D, D),
              "attention.qk_norm.query_norm.scale": (D,), "attention.qk_norm.key_norm.scale": (D,),
              "attention.proj_out.weight": (D, D), "modulation.1.weight": (3 * D, E), "modulation.1.bias": (3 * D,),
              "norm.weight": (D,), "norm.bias": (D,)}
    nj = cfg.deep_layers_depth - cfg.n_single_stream_blocks
    for name, depth in (("layers", cfg.encoder_depth), ("deep_layers", cfg.deep_layers_depth), ("decoder_layers", cfg.decoder_depth)):
        for i in range(depth):
            blk = single if (name == "deep_layers" and i >= nj) else joint
            s.update({f"{name}.{i}." + k: v for k, v in blk.items()})
    return s


def single_stream_block(P, pre: str, x: Tensor, c: Tensor, emb: Tensor, cos: Tensor, sin: Tensor, keep: Tensor | None,
                        cfg: SprintJointConfig) -> tuple[Tensor, Tensor]:
    """mmdit.py:497-532 (MMDiTSingleStreamBlock._forward): attention and MLP in parallel on the modulated [context ; input] tokens"""
    B, Lc, D = c.shape
    H, dh = cfg.num_heads, cfg.head_dim
    lat = torch.cat((c, x), dim=1)
    T = lat.shape[1]
    mod = odit.silu(emb) @ P[pre + "modulation.1.weight"].t() + P[pre + "modulation.1.bias"]
    a, b, g = mod[:, None, :].chunk(3, dim=-1)
    m = odit.layer_norm(lat, P[pre + "norm.weight"], P[pre + "norm.bias"], 1e-5) * (1 + a) + b
    q, k, v = (m @ P[pre + "attention.qkv.weight"].t()).split(D, dim=-1)
    q = odit.rms_norm(q, P[pre + "attention.qk_norm.query_norm.scale"]).to(v.dtype)
    k = odit.rms_norm(k, P[pre + "attention.qk_norm.key_norm.scale"]).to(v.dtype)
    q = odit.apply_rope(q.reshape(B, T, H, dh), cos, sin).transpose(1, 2)
    k = odit.apply_rope(k.reshape(B, T, H, dh), cos, sin).transpose(1, 2)
    v = v.reshape(B, T, H, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * dh**-0.5
    if keep is not None:
        full = torch.cat((keep.bool(), torch.ones(B, T - Lc, dtype=torch.bool)), dim=1)
        s = s.masked_fill(~full[:, None, None, :], float("-inf"))
    att = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, T, D) @ P[pre + "attention.proj_out.weight"].t()
    u1, u3 = (m @ P[pre + "mlp.0.weight"].t()).chunk(2, dim=-1)
    lat = lat + (att + (odit.silu(u1) * u3) @ P[pre + "mlp.2.weight"].t()) * g
    return lat[:, Lc:], lat[:, :Lc]


def sprint_mmdit_forward(P: dict[str, Tensor], x: Tensor, t: Tensor, ctx: Tensor, keep: Tensor | None, cfg: SprintJointConfig,
                         kept: Tensor | None = None, path_drop: Tensor | None = None, skip_deep: bool = False) -> Tensor:
    """sprint.py:389-503 (_forward_mmdit) with the context already through the embedder; routing arguments as in sprint_forward"""
    tok, gh, gw = odit.patchify(P, x, cfg)
    B, S, D = tok.shape
    emb = odit.cond_embedding(P, t, None, cfg)
    c = ctx @ P["context_embed.weight"].t()
    Lc = c.shape[1]
    cos, sin = ommdit.rope_tables_joint(Lc, gh, gw, cfg.rope_axes_dim, cfg.rope_base)
    for i in range(cfg.encoder_depth):
        tok, c = ommdit.joint_block(P, f"layers.{i}.", tok, c, emb, cos, sin, keep, cfg)
    enc_c = c
    mask = P["mask_token"].to(tok.dtype)
    if skip_deep:
        restored = mask.expand(B, S, D)
    else:
        if kept is None:
            kept = torch.arange(S)[None].expand(B, S)
        xd = torch.gather(tok, 1, kept[..., None].expand(-1, -1, D))
        cd = torch.cat((cos[None, :Lc].expand(B, -1, -1), cos[Lc:][kept]), dim=1)  # [B, Lc + k, P]
        sd = torch.cat((sin[None, :Lc].expand(B, -1, -1), sin[Lc:][kept]), dim=1)
        nj = cfg.deep_layers_depth - cfg.n_single_stream_blocks
        for i in range(cfg.deep_layers_depth):
            blk = ommdit.joint_block if i < nj else single_stream_block
            xd, c = blk(P, f"deep_layers.{i}.", xd, c, emb, cd, sd, keep, cfg)
        restored = mask.expand(B, S, D).clone().scatter(1, kept[..., None].expand(-1, -1, D), xd)
        if path_drop is not None:
            restored = torch.where(path_drop[:, None, None], mask.expand(B, S, D), restored)
    xf = torch.cat((restored, tok), dim=-1) @ P["fuse.weight"].t()
    cf = torch.cat((c, enc_c), dim=-1) @ P["fuse_context.weight"].t()
    for i in range(cfg.decoder_depth):
        xf, cf = ommdit.joint_block(P, f"decoder_layers.{i}.", xf, cf, emb, cos, sin, keep, cfg)
    return odit.unpatchify(odit.last_layer(P, xf, emb, cfg), gh, gw, cfg)
