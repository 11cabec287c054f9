"""``diffulab.networks.utils.nn`` -- the reference's primitive layers (networks/utils/nn.py:11-540) as STANDALONE building blocks for a
user who extends a denoiser.  The shipped denoisers do not use this module: their engines run these primitives FUSED into larger
launches (QK-RMSNorm + RoPE inside the attention forward, LayerNorm-modulate as a GEMM epilogue, the SwiGLU in the MLP-up GEMM's
epilogue, ... -- DESIGN.md section 4).  Same class / function names, constructor arguments, ``state_dict`` keys and arithmetic as
the reference; on device tensors

  * ``timestep_embedding``                 -> ``dl_f32_timestep_embedding`` (f32 out, as the reference returns)
  * ``PackedSwiGLU``                       -> ``dl_swiglu_{fwd,bwd}`` (bf16 rows) / ``dl_f32_swiglu_{fwd,bwd}`` (f32 rows), autograd-aware
  * ``qk_norm_rope`` (extra, the fused form of ``QKNorm`` + ``RotaryPositionalEmbeddingNDim`` on packed qkv rows)
                                           -> ``dl_qk_norm_rope_{fwd,bwd}``, autograd-aware
  * ``GroupNorm32``                        -> ``dl_gn_stats`` + ``dl_gn_apply_fwd`` on NHWC rows in inference; torch under autograd

and plain torch device ops for what has no kernel of its own because it only exists fused (``RMSNorm``, ``QKNorm``,
``RotaryPositionalEmbeddingNDim`` applied separately, ``Modulation`` / ``modulate``, ``LabelEmbed``, ``Upsample`` / ``Downsample``):
those are thin, elementwise or tiny, and off the hot path by construction.  The random draw of ``LabelEmbed.drop_labels`` is the
reference's (``torch.rand`` on the labels' device, nn.py:148), so seeded runs drop the same labels.
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from ... import ops

__all__ = ["GroupNorm32", "normalization", "Upsample", "Downsample", "timestep_embedding", "LabelEmbed", "get_cos_sin_ndim_grid",
           "RotaryPositionalEmbeddingNDim", "RMSNorm", "QKNorm", "PackedSwiGLU", "ModulationOut", "Modulation", "modulate", "qk_norm_rope"]


class GroupNorm32(nn.GroupNorm):
    """nn.py:11-13: statistics in f32 whatever the input type.  Inference on bf16 device tensors of >= 8-aligned channels runs the
    UNet engine's GroupNorm kernels (NHWC rows); everything else (autograd, f32, CPU) is the reference's torch expression."""

    def forward(self, input: Tensor) -> Tensor:
        x = input
        if (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and not torch.is_grad_enabled() and x.shape[1] % 8 == 0
                and self.affine and x.shape[1] % self.num_groups == 0):
            B, C, H, W = x.shape
            rows = x.permute(0, 2, 3, 1).contiguous().view(B * H * W, C)
            stats = torch.empty(B, self.num_groups, 2, device=x.device, dtype=torch.float32)
            out = torch.empty_like(rows)
            ops.gn_stats(rows, stats, B, H * W, C, self.num_groups, self.eps)
            ops.gn_apply_fwd(rows, stats, self.weight.float(), self.bias.float(), None, None, False, out, B, H * W, C, self.num_groups)
            return out.view(B, H, W, C).permute(0, 3, 1, 2)
        return super().forward(x.float()).type(x.dtype)


def normalization(channels: int) -> GroupNorm32:
    return GroupNorm32(32, channels)


class Upsample(nn.Module):
    """nn.py:28-56 (nearest x2, optional 3x3 conv)"""

    def __init__(self, channels: int, use_conv: bool, out_channels: int | None = None) -> None:
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        if use_conv:
            self.conv = nn.Conv2d(self.channels, self.out_channels, 3, padding=1)

    def forward(self, x: Tensor) -> Tensor:
        assert x.shape[1] == self.channels
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        return self.conv(x) if self.use_conv else x


class Downsample(nn.Module):
    """nn.py:59-88 (3x3 stride-2 conv, or 2x2 average pooling)"""

    def __init__(self, channels: int, use_conv: bool, out_channels: int | None = None) -> None:
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        if use_conv:
            self.op = nn.Conv2d(self.channels, self.out_channels, 3, stride=2, padding=1)
        else:
            assert self.channels == self.out_channels
            self.op = nn.AvgPool2d(kernel_size=2, stride=2)

    def forward(self, x: Tensor) -> Tensor:
        assert x.shape[1] == self.channels
        return self.op(x)


def timestep_embedding(timesteps: Tensor, dim: int, max_period: int = 10000) -> Tensor:
    """nn.py:91-114: [cos(t f_i) | sin(t f_i)] with f_i = exp(-ln(max_period) i / half), f32 [B, dim] (a zero column when dim is
    odd).  Device tensors: one launch of the f32 embedding kernel (the engines' conditioning path)."""
    half = dim // 2
    if timesteps.is_cuda and dim % 2 == 0 and half > 0:
        out = torch.empty(timesteps.shape[0], dim, device=timesteps.device, dtype=torch.float32)
        ops.f32_timestep_embedding(timesteps.detach().float().contiguous(), out, float(max_period))
        return out
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half).to(device=timesteps.device)
    args = timesteps[:, None].float() * freqs[None]
    embedding = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        embedding = torch.cat([embedding, torch.zeros_like(embedding[:, :1])], dim=-1)
    return embedding


class LabelEmbed(nn.Module):
    """nn.py:117-164: class-embedding table with one extra row for the dropped label"""

    def __init__(self, num_classes: int, embed_dim: int, classifier_free_guidance: bool = False) -> None:
        super().__init__()
        self.num_classes = num_classes
        self.embed_dim = embed_dim
        self.classifier_free_guidance = classifier_free_guidance
        self.embedding = nn.Embedding(num_classes + 1 if classifier_free_guidance else num_classes, embed_dim)

    def drop_labels(self, labels: Tensor, p: float) -> Tensor:
        return torch.where(torch.rand(labels.size(), device=labels.device) < p, self.num_classes, labels)

    def forward(self, labels: Tensor, p: float = 0) -> Tensor:
        if p > 0:
            assert self.classifier_free_guidance, "Label dropout is only supported with classifier-free guidance."
            labels = self.drop_labels(labels, p)
        return self.embedding(labels).squeeze(1)


def get_cos_sin_ndim_grid(pos_id: Tensor, base: float, axes_dim: list[int]) -> tuple[Tensor, Tensor]:
    """nn.py:262-307: per axis i, angles = pos[..., i] (fp64) x base^(-2j / d_i); cos / sin rounded to f32, axes concatenated"""
    assert len(axes_dim) == pos_id.shape[-1], "axes_dim length must match pos_id n_axes"
    cos_chunks, sin_chunks = [], []
    for axis_idx, axis_dim in enumerate(axes_dim):
        pos_i = pos_id[..., axis_idx].to(dtype=torch.float64)
        freqs = 1.0 / (base ** (torch.arange(0, axis_dim, 2, dtype=torch.float64, device=pos_i.device) / axis_dim))
        angles = torch.einsum("...s,d->...sd", pos_i, freqs)
        cos_chunks.append(angles.cos().float())
        sin_chunks.append(angles.sin().float())
    return torch.cat(cos_chunks, dim=-1), torch.cat(sin_chunks, dim=-1)


class RotaryPositionalEmbeddingNDim(nn.Module):
    """nn.py:310-400: interleaved-pair rotation of the first sum(axes_dim) channels of every head of q and k"""

    def __init__(self, axes_dim: list[int]) -> None:
        super().__init__()
        assert len(axes_dim) > 0, "axes_dim must be non-empty"
        for d in axes_dim:
            assert d % 2 == 0, f"Each axis_dim must be even, got {d}"
        self.axes_dim = axes_dim
        self.n_axes = len(axes_dim)
        self.dim = int(sum(axes_dim))

    @staticmethod
    def _apply_rotary(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
        cos, sin = cos[:, None, :, :], sin[:, None, :, :]
        xe, xo = x[..., 0::2], x[..., 1::2]
        return torch.stack([xe * cos - xo * sin, xe * sin + xo * cos], dim=-1).flatten(-2)

    def forward(self, q: Tensor, k: Tensor, v: Tensor, cos_sin: tuple[Tensor, Tensor]) -> tuple[Tensor, Tensor, Tensor]:
        cos, sin = (t.to(device=q.device, dtype=q.dtype) for t in cos_sin)
        q, k = q.transpose(1, 2), k.transpose(1, 2)
        q = torch.cat([self._apply_rotary(q[..., : self.dim], cos, sin), q[..., self.dim :]], dim=-1)
        k = torch.cat([self._apply_rotary(k[..., : self.dim], cos, sin), k[..., self.dim :]], dim=-1)
        return q.transpose(1, 2), k.transpose(1, 2), v


class RMSNorm(nn.Module):
    """nn.py:403-431: x * rsqrt(mean(x^2) + 1e-6) in f32, cast back, times the learnable scale"""

    def __init__(self, dim: int):
        super().__init__()
        self.scale = nn.Parameter(torch.ones(dim))

    def forward(self, x: Tensor) -> Tensor:
        x_dtype = x.dtype
        x = x.float()
        rrms = torch.rsqrt(torch.mean(x**2, dim=-1, keepdim=True) + 1e-6)
        return (x * rrms).to(dtype=x_dtype) * self.scale


class QKNorm(nn.Module):
    """nn.py:434-475"""

    def __init__(self, dim: int):
        super().__init__()
        self.query_norm = RMSNorm(dim)
        self.key_norm = RMSNorm(dim)

    def forward(self, q: Tensor, k: Tensor, v: Tensor) -> tuple[Tensor, Tensor]:
        return self.query_norm(q).to(v), self.key_norm(k).to(v)


class _QkNormRope(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, scale_q, scale_k, cos, sin, B, N, H):
        D = qkv.shape[1] // 3
        dh = D // H
        rot = 2 * cos.shape[-1]
        q, k, v = (torch.empty(B, H, N, dh, device=qkv.device, dtype=torch.bfloat16) for _ in range(3))
        rrms = torch.empty(B * N, 2, device=qkv.device, dtype=torch.float32)
        ops.qk_norm_rope_fwd(qkv, scale_q, scale_k, cos, sin, q, k, v, rrms, B, N, H, dh, rot)
        ctx.save_for_backward(qkv, scale_q, scale_k, cos, sin, rrms)
        ctx.dims = (B, N, H, dh, rot)
        return q, k, v

    @staticmethod
    def backward(ctx, dq, dk, dv):
        qkv, scale_q, scale_k, cos, sin, rrms = ctx.saved_tensors
        B, N, H, dh, rot = ctx.dims
        dqkv = torch.empty_like(qkv)
        dscale = torch.zeros(2, H * dh, device=qkv.device, dtype=torch.float32)
        ops.qk_norm_rope_bwd(dq.contiguous(), dk.contiguous(), dv.contiguous(), qkv, scale_q, scale_k, cos, sin, rrms, dqkv, dscale, B, N, H, dh, rot)
        return dqkv, dscale[0], dscale[1], None, None, None, None, None


def qk_norm_rope(qkv: Tensor, qk_norm: QKNorm, cos_sin: tuple[Tensor, Tensor], batch: int, tokens: int, heads: int) -> tuple[Tensor, Tensor, Tensor]:
    """EXTRA (not in the reference module): ``QKNorm`` (over the full D-wide row) followed by ``RotaryPositionalEmbeddingNDim`` and the
    head split as ONE launch on the packed rows ``qkv`` bf16 [batch * tokens, 3 D] -- the arithmetic of mmdit.py:81-91 -- returning
    q, k, v as bf16 [batch, heads, tokens, head_dim]; cos / sin are the f32 [tokens, rot / 2] tables of ``get_cos_sin_ndim_grid`` for
    one sample (every sample shares them).  Differentiable in qkv and the two scales."""
    cos, sin = (t.reshape(-1, t.shape[-1]).to(device=qkv.device, dtype=torch.float32).contiguous() for t in cos_sin)
    assert qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.shape[0] == batch * tokens and cos.shape[0] == tokens
    return _QkNormRope.apply(qkv.contiguous(), qk_norm.query_norm.scale.float(), qk_norm.key_norm.scale.float(), cos, sin, batch, tokens, heads)


class _SwiGLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u):
        rows = u.reshape(-1, u.shape[-1]).contiguous()
        h = torch.empty(rows.shape[0], rows.shape[1] // 2, device=u.device, dtype=u.dtype)
        (ops.swiglu_fwd if u.dtype == torch.bfloat16 else ops.f32_swiglu_fwd)(rows, h)
        ctx.save_for_backward(rows)
        ctx.shape = u.shape
        return h.view(*u.shape[:-1], rows.shape[1] // 2)

    @staticmethod
    def backward(ctx, dh):
        (rows,) = ctx.saved_tensors
        du = torch.empty_like(rows)
        dh = dh.reshape(rows.shape[0], -1).contiguous()
        (ops.swiglu_bwd if rows.dtype == torch.bfloat16 else ops.f32_swiglu_bwd)(dh, rows, du)
        return du.view(ctx.shape)


class PackedSwiGLU(nn.Module):
    """nn.py:478-486: silu(x1) * x3 for x = [x1 | x3].  bf16 / f32 device rows of a 16-element-aligned width run the engines'
    standalone SwiGLU kernels (with their backward); anything else is the reference's torch expression."""

    def forward(self, x: Tensor) -> Tensor:
        if x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and x.shape[-1] % 16 == 0:
            return _SwiGLU.apply(x)
        x1, x3 = torch.chunk(x, 2, dim=-1)
        return F.silu(x1) * x3


@dataclass
class ModulationOut:
    alpha: Tensor
    beta: Tensor
    gamma: Tensor
    delta: Tensor
    epsilon: Tensor
    zeta: Tensor


class Modulation(nn.Module):
    """nn.py:499-536: lin(silu(vec)) cut into six chunks ([B, 1, D] each for a [B, E] input)"""

    def __init__(self, embedding_dim: int, input_dim: int):
        super().__init__()
        self.lin = nn.Linear(embedding_dim, 6 * input_dim, bias=True)

    def forward(self, vec: Tensor) -> ModulationOut:
        out = self.lin(F.silu(vec))
        if out.dim() == 2:
            out = out[:, None, :]
        return ModulationOut(*out.chunk(6, dim=-1))


def modulate(x: Tensor, scale: Tensor, shift: Tensor) -> Tensor:
    return x * (1 + scale) + shift
