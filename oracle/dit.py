"""Oracle: DiT (``MMDiT(simple_dit=True)``) forward in plain torch.  TEST INFRASTRUCTURE ONLY.

Functional restatement: the model is a ``dict[str, Tensor]`` keyed exactly like
the reference ``state_dict`` plus a small config.  The
backward pass used by the tests is torch autograd over these functions on CPU.

Reference sites (``/root/reference/src/diffulab``):
  networks/utils/nn.py:91-114     timestep_embedding
  networks/utils/nn.py:149-164    LabelEmbed (+ drop_labels)
  networks/utils/nn.py:262-307    get_cos_sin_ndim_grid (fp64 tables)
  networks/utils/nn.py:333-400    RotaryPositionalEmbeddingNDim (interleaved pairs)
  networks/utils/nn.py:427-431    RMSNorm ; :473-475 QKNorm
  networks/utils/nn.py:484-486    PackedSwiGLU ; :530-540 Modulation / modulate
  networks/denoisers/mmdit.py:75-104   DiTAttention.forward
  networks/denoisers/mmdit.py:288-309  DiTBlock._forward
  networks/denoisers/mmdit.py:542-549  ModulatedLastLayer.forward
  networks/denoisers/mmdit.py:757-787  patchify / unpatchify
  networks/denoisers/mmdit.py:853-928  simple_dit_forward / forward
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
from torch import Tensor


@dataclass
class DiTConfig:
    input_channels: int = 4
    output_channels: int = 4
    inner_dim: int = 384
    embedding_dim: int = 384
    num_heads: int = 6
    mlp_ratio: int = 4
    patch_size: int = 2
    depth: int = 12
    rope_base: float = 10_000.0
    frequency_embedding: int = 256
    n_classes: int | None = 1000
    classifier_free: bool = True
    rope_axes_dim: list[int] = field(default_factory=list)

    def __post_init__(self) -> None:
        if not self.rope_axes_dim:
            hd = self.inner_dim // self.num_heads
            self.rope_axes_dim = [hd // 2, hd // 2]  # mmdit.py:673-677 (partial_rotary_factor=1)

    @property
    def head_dim(self) -> int:
        return self.inner_dim // self.num_heads


# ----------------------------------------------------------------------------- primitives


def timestep_embedding(t: Tensor, dim: int, max_period: float = 10000.0) -> Tensor:
    """nn.py:106-114: [cos(t f_i) | sin(t f_i)], f_i = exp(-ln(P) i / half), fp32."""
    half = dim // 2
    idx = torch.arange(half, dtype=torch.float32)
    freqs = torch.exp(idx * (-math.log(max_period)) / half)
    ang = t.to(torch.float32).reshape(-1, 1) * freqs.reshape(1, -1)
    out = torch.cat((ang.cos(), ang.sin()), dim=1)
    if dim % 2:
        out = torch.cat((out, out.new_zeros(out.shape[0], 1)), dim=1)
    return out


def rope_tables(grid_h: int, grid_w: int, axes_dim: list[int], base: float) -> tuple[Tensor, Tensor]:
    """nn.py:276-307 on the meshgrid(indexing="ij") ids built at mmdit.py:871-886.

    Returns cos, sin of shape [grid_h*grid_w, sum(axes_dim)//2] (fp64 angles -> fp32).
    Axis 0 is the row index, axis 1 the column index; the batch repeat of the
    reference is dropped (every batch element gets the same table).
    """
    rows = torch.arange(grid_h, dtype=torch.float64).repeat_interleave(grid_w)
    cols = torch.arange(grid_w, dtype=torch.float64).repeat(grid_h)
    cos_parts, sin_parts = [], []
    for pos, d in zip((rows, cols), axes_dim):
        expo = torch.arange(0, d, 2, dtype=torch.float64) / d
        inv = 1.0 / (torch.tensor(float(base), dtype=torch.float64) ** expo)
        ang = pos[:, None] * inv[None, :]
        cos_parts.append(ang.cos().to(torch.float32))
        sin_parts.append(ang.sin().to(torch.float32))
    return torch.cat(cos_parts, dim=1), torch.cat(sin_parts, dim=1)


def apply_rope(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """nn.py:345-353: rotate interleaved pairs (x[2j], x[2j+1]) of the first 2*P channels.

    x: [B, N, H, dh]; cos/sin: [N, P], or [B, N, P] per-sample tables (SPRINT's gathered rows, sprint.py:348-353).
    """
    rot = 2 * cos.shape[-1]
    xr, xp = x[..., :rot], x[..., rot:]
    a = xr[..., 0::2]
    b = xr[..., 1::2]
    c = (cos[None] if cos.dim() == 2 else cos)[:, :, None, :].to(x.dtype)
    s = (sin[None] if sin.dim() == 2 else sin)[:, :, None, :].to(x.dtype)
    ra = a * c - b * s
    rb = a * s + b * c
    out = torch.stack((ra, rb), dim=-1).flatten(-2)
    return torch.cat((out, xp), dim=-1)


def rms_norm(x: Tensor, scale: Tensor, eps: float = 1e-6) -> Tensor:
    """nn.py:427-431: fp32 stats over the last dim, cast back, then * scale."""
    xf = x.to(torch.float32)
    r = torch.rsqrt((xf * xf).mean(dim=-1, keepdim=True) + eps)
    return (xf * r).to(x.dtype) * scale


def layer_norm(x: Tensor, w: Tensor | None, b: Tensor | None, eps: float) -> Tensor:
    x = x.float()  # no-op on the fp32 path; under bf16_autocast() the statistics and the output are f32, like torch's layer_norm
    #                on the GPU autocast list
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    y = (x - mu) * torch.rsqrt(var + eps)
    if w is not None:
        y = y * w + b
    return y


def silu(x: Tensor) -> Tensor:
    return x * torch.sigmoid(x)


def attention(q: Tensor, k: Tensor, v: Tensor, scale: float) -> Tensor:
    """softmax(q k^T * scale) v, no mask (mmdit.py:92-98).  q,k,v: [B, H, N, dh]."""
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, v)


def bf16_autocast():
    """the oracle's bf16 leg (test yardstick only): the SAME functions under ``torch.autocast("cpu", bfloat16)`` -- every ``@`` /
    matmul runs on bf16 operands and returns bf16, the residual stream is bf16, LayerNorm / RMSNorm statistics and softmax stay
    f32 -- i.e. the dtype flow of the reference DiT under accelerate's bf16 mixed precision.  Comparing it with
    the fp32 oracle gives the error a bf16 pipeline makes on its own; the HIP path is held to a multiple of that."""
    return torch.autocast("cpu", dtype=torch.bfloat16)


# ----------------------------------------------------------------------------- blocks


def dit_attention(P: dict[str, Tensor], pre: str, x: Tensor, cos: Tensor, sin: Tensor, cfg: DiTConfig) -> Tensor:
    """mmdit.py:75-104."""
    B, N, D = x.shape
    H, dh = cfg.num_heads, cfg.head_dim
    qkv = x @ P[pre + "qkv.weight"].t()
    q, k, v = qkv.split(D, dim=-1)
    q = rms_norm(q, P[pre + "qk_norm.query_norm.scale"]).to(v.dtype)
    k = rms_norm(k, P[pre + "qk_norm.key_norm.scale"]).to(v.dtype)
    q = apply_rope(q.reshape(B, N, H, dh), cos, sin).transpose(1, 2)
    k = apply_rope(k.reshape(B, N, H, dh), cos, sin).transpose(1, 2)
    v = v.reshape(B, N, H, dh).transpose(1, 2)
    o = attention(q, k, v, dh**-0.5)
    o = o.transpose(1, 2).reshape(B, N, D)
    return o @ P[pre + "proj_out.weight"].t()


def dit_block(
    P: dict[str, Tensor], pre: str, x: Tensor, emb: Tensor, cos: Tensor, sin: Tensor, cfg: DiTConfig,
    taps: dict[str, Tensor] | None = None,
) -> Tensor:
    """mmdit.py:288-309 (adaLN-zero block).  ``emb`` is [B, E] or [B, S, E] (nn.py:530-534)."""
    D = cfg.inner_dim
    mod = silu(emb) @ P[pre + "modulation.lin.weight"].t() + P[pre + "modulation.lin.bias"]
    if mod.dim() == 2:
        mod = mod[:, None, :]
    a, b, g, d, e, z = mod.split(D, dim=-1)
    h1 = layer_norm(x, P[pre + "norm_1.weight"], P[pre + "norm_1.bias"], 1e-5) * (1 + a) + b
    att = dit_attention(P, pre + "attention.", h1, cos, sin, cfg)
    x1 = x + att * g
    h2 = layer_norm(x1, P[pre + "norm_2.weight"], P[pre + "norm_2.bias"], 1e-5) * (1 + d) + e
    u = h2 @ P[pre + "mlp_input.0.weight"].t()
    u1, u3 = u.chunk(2, dim=-1)
    hid = silu(u1) * u3
    t2 = hid @ P[pre + "mlp_input.2.weight"].t()
    out = x1 + t2 * z
    if taps is not None:
        taps.update(ln_mod_1=h1, attn_proj=att, x_mid=x1, ln_mod_2=h2, mlp_hidden=hid, mlp_out=t2)
    return out


def patchify(P: dict[str, Tensor], x: Tensor, cfg: DiTConfig) -> tuple[Tensor, int, int]:
    """mmdit.py:757-765: stride-p conv without bias, tokens ordered (h w)."""
    B, C, Hh, Ww = x.shape
    p = cfg.patch_size
    gh, gw = Hh // p, Ww // p
    patches = x.reshape(B, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * p * p)
    w = P["conv_proj.weight"].reshape(cfg.inner_dim, C * p * p)
    return patches @ w.t(), gh, gw


def unpatchify(tok: Tensor, gh: int, gw: int, cfg: DiTConfig) -> Tensor:
    """mmdit.py:778-787: 'b (h w) (p1 p2 c) -> b c (h p1) (w p2)' (c fastest)."""
    B = tok.shape[0]
    p, C = cfg.patch_size, cfg.output_channels
    t = tok.reshape(B, gh, gw, p, p, C).permute(0, 5, 1, 3, 2, 4)
    return t.reshape(B, C, gh * p, gw * p)


def cond_embedding(P: dict[str, Tensor], t: Tensor, y_eff: Tensor | None, cfg: DiTConfig) -> Tensor:
    """mmdit.py:866-868: time MLP of the sinusoidal embedding (+ label row).

    ``y_eff`` are the labels AFTER the classifier-free drop (nn.py:149), i.e. dropped
    entries already replaced by ``n_classes``.
    """
    te = timestep_embedding(t, cfg.frequency_embedding)
    h = silu(te @ P["time_embed.0.weight"].t() + P["time_embed.0.bias"])
    emb = h @ P["time_embed.2.weight"].t() + P["time_embed.2.bias"]
    if y_eff is not None:
        emb = emb + P["label_embed.embedding.weight"][y_eff.long()]
    return emb


def drop_labels(y: Tensor, p: float, n_classes: int, u: Tensor | None = None) -> Tensor:
    """nn.py:136-164: where(rand < p, n_classes, y).  ``u`` lets a test inject the uniforms."""
    if p <= 0:
        return y
    if u is None:
        u = torch.rand(y.shape, device=y.device)
    return torch.where(u < p, torch.full_like(y, n_classes), y)


def last_layer(P: dict[str, Tensor], x: Tensor, emb: Tensor, cfg: DiTConfig) -> Tensor:
    """mmdit.py:542-549."""
    D = cfg.inner_dim
    mod = silu(emb) @ P["last_layer.adaLN_modulation.1.weight"].t() + P["last_layer.adaLN_modulation.1.bias"]
    if mod.dim() == 2:
        mod = mod[:, None, :]
    a, b = mod.split(D, dim=-1)
    h = layer_norm(x, None, None, 1e-6) * (1 + a) + b
    return h @ P["last_layer.linear.weight"].t() + P["last_layer.linear.bias"]


def dit_forward(
    P: dict[str, Tensor], x: Tensor, t: Tensor, y_eff: Tensor | None, cfg: DiTConfig,
    taps: dict[str, Tensor] | None = None,
) -> Tensor:
    """mmdit.py:903-928 with simple_dit=True.  Returns the prediction [B, C_out, H, W]."""
    tok, gh, gw = patchify(P, x, cfg)
    emb = cond_embedding(P, t, y_eff, cfg)
    cos, sin = rope_tables(gh, gw, cfg.rope_axes_dim, cfg.rope_base)
    for i in range(cfg.depth):
        tok = dit_block(P, f"layers.{i}.", tok, emb, cos, sin, cfg)
        if taps is not None:
            taps[f"layer{i}"] = tok
    out = last_layer(P, tok, emb, cfg)
    return unpatchify(out, gh, gw, cfg)


# ----------------------------------------------------------------------------- parameters


def param_shapes(cfg: DiTConfig) -> dict[str, tuple[int, ...]]:
    """state_dict layout of MMDiT(simple_dit=True)."""
    D, E, p = cfg.inner_dim, cfg.embedding_dim, cfg.patch_size
    shapes: dict[str, tuple[int, ...]] = {}
    if cfg.n_classes is not None:
        shapes["label_embed.embedding.weight"] = (cfg.n_classes + (1 if cfg.classifier_free else 0), E)
    shapes["last_layer.linear.weight"] = (p * p * cfg.output_channels, D)
    shapes["last_layer.linear.bias"] = (p * p * cfg.output_channels,)
    shapes["last_layer.adaLN_modulation.1.weight"] = (2 * D, E)
    shapes["last_layer.adaLN_modulation.1.bias"] = (2 * D,)
    shapes["time_embed.0.weight"] = (E, cfg.frequency_embedding)
    shapes["time_embed.0.bias"] = (E,)
    shapes["time_embed.2.weight"] = (E, E)
    shapes["time_embed.2.bias"] = (E,)
    shapes["conv_proj.weight"] = (D, cfg.input_channels, p, p)
    for i in range(cfg.depth):
        pre = f"layers.{i}."
        shapes[pre + "modulation.lin.weight"] = (6 * D, E)
        shapes[pre + "modulation.lin.bias"] = (6 * D,)
        shapes[pre + "norm_1.weight"] = (D,)
        shapes[pre + "norm_1.bias"] = (D,)
        shapes[pre + "attention.qkv.weight"] = (3 * D, D)
        shapes[pre + "attention.qk_norm.query_norm.scale"] = (D,)
        shapes[pre + "attention.qk_norm.key_norm.scale"] = (D,)
        shapes[pre + "attention.proj_out.weight"] = (D, D)
        shapes[pre + "norm_2.weight"] = (D,)
        shapes[pre + "norm_2.bias"] = (D,)
        shapes[pre + "mlp_input.0.weight"] = (2 * cfg.mlp_ratio * D, D)
        shapes[pre + "mlp_input.2.weight"] = (D, cfg.mlp_ratio * D)
    return shapes
