"""Oracle: guided-diffusion style UNet (``UNetModel``) forward in plain torch.  TEST INFRASTRUCTURE ONLY.

Functional restatement over a ``dict[str, Tensor]`` keyed like the reference ``state_dict``
(``input_blocks.{i}.{j}...``, ``middle_block.{j}...``, ``output_blocks.{i}.{j}...``, ``out.{0,2}``, ``time_embed.{0,2}``,
``label_embed.embedding``), self-attention / class-conditional variants: FiLM or additive conditioning
(``use_scale_shift_norm``), ResBlock or plain ``Upsample`` / ``Downsample`` resampling (``resblock_updown``), with or without
the resampling conv (``conv_resample``).  ``configs/model/unet.yaml`` builds the FiLM + ResBlock-resampling one.

Reference sites (``/root/reference/src/diffulab``):
  networks/utils/nn.py:11-25       GroupNorm32 (fp32 statistics, 32 groups, eps 1e-5)
  networks/utils/nn.py:28-88       Upsample (nearest x2 [+ 3x3 conv]) / Downsample (3x3 stride-2 conv, or avg-pool 2)
  networks/denoisers/unet.py:215-237   ResBlock._forward (FiLM scale/shift, up/down, zero-init out conv)
  networks/denoisers/unet.py:296-322   AttentionBlock._forward (GN -> 1x1 q / kv -> SDPA -> 1x1 out -> + x)
  networks/denoisers/unet.py:593-745   block wiring ; :832-853 forward
"""

from __future__ import annotations

from dataclasses import dataclass, field

import torch
import torch.nn.functional as F
from torch import Tensor

from .dit import silu, timestep_embedding


@dataclass
class UNetConfig:
    image_size: tuple[int, int] = (32, 32)
    in_channels: int = 1
    model_channels: int = 128
    out_channels: int = 1
    num_res_blocks: int = 2
    attention_resolutions: tuple[int, ...] = (4, 8, 16)   # DOWNSAMPLE FACTORS (unet.py:611), not resolutions
    channel_mult: tuple[int, ...] = (1, 2, 4, 8)
    num_heads: int = 2
    use_scale_shift_norm: bool = True
    resblock_updown: bool = True
    conv_resample: bool = True   # only read when resblock_updown is False (unet.py:646,735)
    n_classes: int | None = 10
    classifier_free: bool = False


@dataclass
class Block:
    kind: str                 # "conv" | "res" | "attn" | "down" | "up" (the last two: nn.py Downsample / Upsample)
    prefix: str
    cin: int = 0
    cout: int = 0
    up: bool = False
    down: bool = False


@dataclass
class Plan:
    input_blocks: list[list[Block]] = field(default_factory=list)
    middle: list[Block] = field(default_factory=list)
    output_blocks: list[list[Block]] = field(default_factory=list)
    final_ch: int = 0


def build_plan(cfg: UNetConfig) -> Plan:
    """unet.py:593-745, no context embedder."""
    mc = cfg.model_channels
    plan = Plan()
    ch = cfg.channel_mult[0] * mc
    plan.input_blocks.append([Block("conv", "input_blocks.0.0.", cfg.in_channels, ch)])
    chans = [ch]
    ds = 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            i = len(plan.input_blocks)
            layers = [Block("res", f"input_blocks.{i}.0.", ch, mult * mc)]
            ch = mult * mc
            if ds in cfg.attention_resolutions:
                layers.append(Block("attn", f"input_blocks.{i}.1.", ch, ch))
            plan.input_blocks.append(layers)
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            i = len(plan.input_blocks)
            plan.input_blocks.append([Block("res", f"input_blocks.{i}.0.", ch, ch, down=True) if cfg.resblock_updown
                                      else Block("down", f"input_blocks.{i}.0.", ch, ch)])
            chans.append(ch)
            ds *= 2
    plan.middle = [Block("res", "middle_block.0.", ch, ch), Block("attn", "middle_block.1.", ch, ch),
                   Block("res", "middle_block.2.", ch, ch)]
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for k in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            i = len(plan.output_blocks)
            layers = [Block("res", f"output_blocks.{i}.0.", ch + ich, mc * mult)]
            ch = mc * mult
            if ds in cfg.attention_resolutions:
                layers.append(Block("attn", f"output_blocks.{i}.{len(layers)}.", ch, ch))
            if level and k == cfg.num_res_blocks:
                layers.append(Block("res", f"output_blocks.{i}.{len(layers)}.", ch, ch, up=True) if cfg.resblock_updown
                              else Block("up", f"output_blocks.{i}.{len(layers)}.", ch, ch))
                ds //= 2
            plan.output_blocks.append(layers)
    plan.final_ch = ch
    return plan


def param_shapes(cfg: UNetConfig) -> dict[str, tuple[int, ...]]:
    plan = build_plan(cfg)
    te = 4 * cfg.model_channels
    s: dict[str, tuple[int, ...]] = {
        "time_embed.0.weight": (te, cfg.model_channels), "time_embed.0.bias": (te,),
        "time_embed.2.weight": (te, te), "time_embed.2.bias": (te,),
    }
    if cfg.n_classes is not None:
        s["label_embed.embedding.weight"] = (cfg.n_classes + (1 if cfg.classifier_free else 0), te)

    def add(b: Block) -> None:
        p = b.prefix
        if b.kind == "conv":
            s[p + "weight"], s[p + "bias"] = (b.cout, b.cin, 3, 3), (b.cout,)
        elif b.kind == "res":
            s[p + "in_layers.0.weight"], s[p + "in_layers.0.bias"] = (b.cin,), (b.cin,)
            s[p + "in_layers.2.weight"], s[p + "in_layers.2.bias"] = (b.cout, b.cin, 3, 3), (b.cout,)
            eo = 2 * b.cout if cfg.use_scale_shift_norm else b.cout
            s[p + "emb_layers.1.weight"], s[p + "emb_layers.1.bias"] = (eo, te), (eo,)
            s[p + "out_layers.0.weight"], s[p + "out_layers.0.bias"] = (b.cout,), (b.cout,)
            s[p + "out_layers.3.weight"], s[p + "out_layers.3.bias"] = (b.cout, b.cout, 3, 3), (b.cout,)
            if b.cin != b.cout:
                s[p + "skip_connection.weight"], s[p + "skip_connection.bias"] = (b.cout, b.cin, 1, 1), (b.cout,)
        elif b.kind in ("down", "up"):
            if cfg.conv_resample:  # Downsample.op / Upsample.conv (nn.py:47,79)
                n = "op." if b.kind == "down" else "conv."
                s[p + n + "weight"], s[p + n + "bias"] = (b.cout, b.cin, 3, 3), (b.cout,)
        else:
            c = b.cin
            for n in ("norm_x", "norm_context"):
                s[p + n + ".weight"], s[p + n + ".bias"] = (c,), (c,)
            s[p + "to_q.weight"], s[p + "to_q.bias"] = (c, c, 1), (c,)
            s[p + "to_kv.weight"], s[p + "to_kv.bias"] = (2 * c, c, 1), (2 * c,)
            s[p + "to_out.0.weight"], s[p + "to_out.0.bias"] = (c, c, 1), (c,)

    for blk in plan.input_blocks + [plan.middle] + plan.output_blocks:
        for b in blk:
            add(b)
    s["out.0.weight"], s["out.0.bias"] = (plan.final_ch,), (plan.final_ch,)
    s["out.2.weight"], s["out.2.bias"] = (cfg.out_channels, cfg.channel_mult[0] * cfg.model_channels, 3, 3), (cfg.out_channels,)
    return s


def group_norm32(x: Tensor, w: Tensor, b: Tensor, groups: int = 32, eps: float = 1e-5) -> Tensor:
    """nn.py:11-13: statistics in fp32 over (C/groups, *spatial) per sample."""
    B, C = x.shape[:2]
    xf = x.float().reshape(B, groups, -1)
    mu = xf.mean(dim=2, keepdim=True)
    var = ((xf - mu) ** 2).mean(dim=2, keepdim=True)
    y = ((xf - mu) * torch.rsqrt(var + eps)).reshape(x.shape)
    shape = (1, C) + (1,) * (x.dim() - 2)
    return (y * w.reshape(shape) + b.reshape(shape)).to(x.dtype)


def res_block(P: dict[str, Tensor], b: Block, x: Tensor, emb: Tensor, cfg: UNetConfig,
              taps: dict[str, Tensor] | None = None) -> Tensor:
    """unet.py:215-237."""
    p = b.prefix
    h = silu(group_norm32(x, P[p + "in_layers.0.weight"], P[p + "in_layers.0.bias"]))
    if b.up:
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif b.down:
        h = F.avg_pool2d(h, 2)
        x = F.avg_pool2d(x, 2)
    h = F.conv2d(h, P[p + "in_layers.2.weight"], P[p + "in_layers.2.bias"], padding=1)
    eo = (silu(emb) @ P[p + "emb_layers.1.weight"].t() + P[p + "emb_layers.1.bias"])[:, :, None, None]
    if cfg.use_scale_shift_norm:
        scale, shift = eo.chunk(2, dim=1)
        h = group_norm32(h, P[p + "out_layers.0.weight"], P[p + "out_layers.0.bias"]) * (1 + scale) + shift
        h = silu(h)
    else:
        h = silu(group_norm32(h + eo, P[p + "out_layers.0.weight"], P[p + "out_layers.0.bias"]))
    h = F.conv2d(h, P[p + "out_layers.3.weight"], P[p + "out_layers.3.bias"], padding=1)  # dropout p = 0
    if b.cin != b.cout:
        x = F.conv2d(x, P[p + "skip_connection.weight"], P[p + "skip_connection.bias"])
    if taps is not None:
        taps["h"] = h
    return x + h


def attention_block(P: dict[str, Tensor], b: Block, x: Tensor, cfg: UNetConfig) -> Tensor:
    """unet.py:296-322 (context = x, no mask)."""
    p = b.prefix
    B, C, Hh, Ww = x.shape
    xs = x.reshape(B, C, -1)
    q = F.conv1d(group_norm32(xs, P[p + "norm_x.weight"], P[p + "norm_x.bias"]), P[p + "to_q.weight"], P[p + "to_q.bias"])
    kv = F.conv1d(group_norm32(xs, P[p + "norm_context.weight"], P[p + "norm_context.bias"]), P[p + "to_kv.weight"],
                  P[p + "to_kv.bias"])
    k, v = kv.chunk(2, dim=1)
    H = cfg.num_heads
    d = C // H

    def heads(t: Tensor) -> Tensor:  # b (h d) n -> b h n d
        return t.reshape(B, H, d, -1).transpose(2, 3)

    qh, kh, vh = heads(q), heads(k), heads(v)
    att = torch.softmax(qh @ kh.transpose(-1, -2) * d**-0.5, dim=-1) @ vh
    out = att.transpose(2, 3).reshape(B, C, -1)
    out = F.conv1d(out, P[p + "to_out.0.weight"], P[p + "to_out.0.bias"])
    return (xs + out).reshape(B, C, Hh, Ww)


def run_blocks(P, blocks: list[Block], h: Tensor, emb: Tensor, cfg: UNetConfig) -> Tensor:
    for b in blocks:
        if b.kind == "conv":
            h = F.conv2d(h, P[b.prefix + "weight"], P[b.prefix + "bias"], padding=1)
        elif b.kind == "res":
            h = res_block(P, b, h, emb, cfg)
        elif b.kind == "down":  # nn.py:84-88
            h = (F.conv2d(h, P[b.prefix + "op.weight"], P[b.prefix + "op.bias"], stride=2, padding=1) if cfg.conv_resample
                 else F.avg_pool2d(h, 2))
        elif b.kind == "up":  # nn.py:49-56
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            if cfg.conv_resample:
                h = F.conv2d(h, P[b.prefix + "conv.weight"], P[b.prefix + "conv.bias"], padding=1)
        else:
            h = attention_block(P, b, h, cfg)
    return h


def unet_forward(P: dict[str, Tensor], x: Tensor, t: Tensor, y_eff: Tensor | None, cfg: UNetConfig) -> Tensor:
    """unet.py:832-853."""
    plan = build_plan(cfg)
    te = timestep_embedding(t, cfg.model_channels)
    emb = silu(te @ P["time_embed.0.weight"].t() + P["time_embed.0.bias"]) @ P["time_embed.2.weight"].t() + P["time_embed.2.bias"]
    if y_eff is not None:
        emb = emb + P["label_embed.embedding.weight"][y_eff.long()]
    hs = []
    h = x
    for blk in plan.input_blocks:
        h = run_blocks(P, blk, h, emb, cfg)
        hs.append(h)
    h = run_blocks(P, plan.middle, h, emb, cfg)
    for blk in plan.output_blocks:
        h = torch.cat([h, hs.pop()], dim=1)
        h = run_blocks(P, blk, h, emb, cfg)
    h = silu(group_norm32(h, P["out.0.weight"], P["out.0.bias"]))
    return F.conv2d(h, P["out.2.weight"], P["out.2.bias"], padding=1)
