"""CPU restatement of MMDiT with simple_dit=False: joint text-image blocks (reference networks/denoisers/mmdit.py:107-210,
312-439, 789-851) behind a precomputed-embedding context (embedders/precomputed.py) -- TEST INFRASTRUCTURE ONLY.
Pinned by tests/golden/mmdit_joint.npz (outputs of the reference module with its random draws recorded).
"""

from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import Tensor

from . import dit as odit


@dataclass
class JointConfig(odit.DiTConfig):
    context_dim: int = 96
    n_classes: int | None = None
    n_single_stream_blocks: int = 0  # the last blocks of the stack are MMDiTSingleStreamBlocks (mmdit.py:715-728)

    def __post_init__(self) -> None:
        if not self.rope_axes_dim:
            hd = self.inner_dim // self.num_heads
            self.rope_axes_dim = [hd // 3] * 3  # mmdit.py:659-664 (text position, row, col)


STREAMS = ("input", "context")


def param_shapes(cfg: JointConfig) -> dict[str, tuple[int, ...]]:
    """state_dict layout of MMDiT(simple_dit=False, context_embedder with one output, n_single_stream_blocks=0)"""
    D, E, p = cfg.inner_dim, cfg.embedding_dim, cfg.patch_size
    s: dict[str, tuple[int, ...]] = {"context_embed.weight": (D, cfg.context_dim)}
    s["last_layer.linear.weight"], s["last_layer.linear.bias"] = (p * p * cfg.output_channels, D), (p * p * cfg.output_channels,)
    s["last_layer.adaLN_modulation.1.weight"], s["last_layer.adaLN_modulation.1.bias"] = (2 * D, E), (2 * D,)
    s["time_embed.0.weight"], s["time_embed.0.bias"] = (E, cfg.frequency_embedding), (E,)
    s["time_embed.2.weight"], s["time_embed.2.bias"] = (E, E), (E,)
    s["conv_proj.weight"] = (D, cfg.input_channels, p, p)
    F = cfg.mlp_ratio * D
    for i in range(cfg.depth - cfg.n_single_stream_blocks, cfg.depth):
        pre = f"layers.{i}."
        s.update({pre + "mlp.0.weight": (2 * F, D), pre + "mlp.2.weight": (D, F), pre + "attention.qkv.weight": (3 * D, D),
                  pre + "attention.qk_norm.query_norm.scale": (D,), pre + "attention.qk_norm.key_norm.scale": (D,),
                  pre + "attention.proj_out.weight": (D, D), pre + "modulation.1.weight": (3 * D, E),
                  pre + "modulation.1.bias": (3 * D,), pre + "norm.weight": (D,), pre + "norm.bias": (D,)})
    for i in range(cfg.depth - cfg.n_single_stream_blocks):
        pre = f"layers.{i}."
        for st in STREAMS:
            s[pre + f"modulation_{st}.lin.weight"], s[pre + f"modulation_{st}.lin.bias"] = (6 * D, E), (6 * D,)
            for n in (1, 2):
                s[pre + f"{st}_norm_{n}.weight"], s[pre + f"{st}_norm_{n}.bias"] = (D,), (D,)
            s[pre + f"attention.qkv_{st}.weight"] = (3 * D, D)
            s[pre + f"attention.qk_norm_{st}.query_norm.scale"] = (D,)
            s[pre + f"attention.qk_norm_{st}.key_norm.scale"] = (D,)
            s[pre + f"attention.{st}_proj_out.weight"] = (D, D)
            s[pre + f"mlp_{st}.0.weight"], s[pre + f"mlp_{st}.2.weight"] = (2 * cfg.mlp_ratio * D, D), (D, cfg.mlp_ratio * D)
    return s


def rope_tables_joint(n_ctx: int, gh: int, gw: int, axes_dim: list[int], base: float) -> tuple[Tensor, Tensor]:
    """mmdit.py:813-835 + nn.py:262-307: text tokens sit at (1..n_ctx, 0, 0), image tokens at (0, h, w); rows ordered
    [text ; image], angles in fp64 then cast to fp32"""
    pos = torch.zeros(n_ctx + gh * gw, 3, dtype=torch.float64)
    pos[:n_ctx, 0] = torch.arange(1, n_ctx + 1, dtype=torch.float64)
    pos[n_ctx:, 1] = torch.arange(gh, dtype=torch.float64).repeat_interleave(gw)
    pos[n_ctx:, 2] = torch.arange(gw, dtype=torch.float64).repeat(gh)
    cs, sn = [], []
    for a, d in enumerate(axes_dim):
        freqs = 1.0 / (base ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
        ang = pos[:, a, None] * freqs[None, :]
        cs.append(ang.cos().float())
        sn.append(ang.sin().float())
    return torch.cat(cs, 1), torch.cat(sn, 1)


def _mods(P, pre: str, st: str, emb: Tensor, D: int):
    m = odit.silu(emb) @ P[pre + f"modulation_{st}.lin.weight"].t() + P[pre + f"modulation_{st}.lin.bias"]
    return m[:, None, :].split(D, dim=-1)


def joint_attention(P, pre: str, x: Tensor, c: Tensor, cos: Tensor, sin: Tensor, keep: Tensor | None, cfg: JointConfig):
    """mmdit.py:172-210: per-stream qkv + QK-norm, tokens concatenated [context ; input], RoPE, key-padding mask, SDPA, per-stream
    output projections"""
    B, N, D = x.shape
    Lc, H, dh = c.shape[1], cfg.num_heads, cfg.head_dim
    qs, ks, vs = [], [], []
    for st, h in (("context", c), ("input", x)):
        q, k, v = (h @ P[pre + f"qkv_{st}.weight"].t()).split(D, dim=-1)
        qs.append(odit.rms_norm(q, P[pre + f"qk_norm_{st}.query_norm.scale"]).to(v.dtype))
        ks.append(odit.rms_norm(k, P[pre + f"qk_norm_{st}.key_norm.scale"]).to(v.dtype))
        vs.append(v)
    T = Lc + N
    q = odit.apply_rope(torch.cat(qs, 1).reshape(B, T, H, dh), cos, sin).transpose(1, 2)
    k = odit.apply_rope(torch.cat(ks, 1).reshape(B, T, H, dh), cos, sin).transpose(1, 2)
    v = torch.cat(vs, 1).reshape(B, T, H, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * dh**-0.5
    if keep is not None:
        full = torch.cat((keep.bool(), torch.ones(B, N, dtype=torch.bool)), dim=1)
        s = s.masked_fill(~full[:, None, None, :], float("-inf"))
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, T, D)
    return o[:, Lc:] @ P[pre + "input_proj_out.weight"].t(), o[:, :Lc] @ P[pre + "context_proj_out.weight"].t()


def joint_block(P, pre: str, x: Tensor, c: Tensor, emb: Tensor, cos: Tensor, sin: Tensor, keep: Tensor | None, cfg: JointConfig):
    """mmdit.py:404-439 (MMDiTBlock._forward)"""
    D = cfg.inner_dim
    mi, mc = _mods(P, pre, "input", emb, D), _mods(P, pre, "context", emb, D)
    hx = odit.layer_norm(x, P[pre + "input_norm_1.weight"], P[pre + "input_norm_1.bias"], 1e-5) * (1 + mi[0]) + mi[1]
    hc = odit.layer_norm(c, P[pre + "context_norm_1.weight"], P[pre + "context_norm_1.bias"], 1e-5) * (1 + mc[0]) + mc[1]
    ax, ac = joint_attention(P, pre + "attention.", hx, hc, cos, sin, keep, cfg)
    x, c = x + ax * mi[2], c + ac * mc[2]
    outs = []
    for st, h, m in (("input", x, mi), ("context", c, mc)):
        z = odit.layer_norm(h, P[pre + f"{st}_norm_2.weight"], P[pre + f"{st}_norm_2.bias"], 1e-5) * (1 + m[3]) + m[4]
        u1, u3 = (z @ P[pre + f"mlp_{st}.0.weight"].t()).chunk(2, dim=-1)
        outs.append(h + ((odit.silu(u1) * u3) @ P[pre + f"mlp_{st}.2.weight"].t()) * m[5])
    return outs[0], outs[1]


def drop_context(emb: Tensor, keep: Tensor, null_emb: Tensor, null_keep: Tensor, drop: Tensor) -> tuple[Tensor, Tensor]:
    """embedders/precomputed.py:23-39: samples with drop[b] get the null embedding and its mask"""
    B = emb.shape[0]
    return (torch.where(drop[:, None, None], null_emb[None].expand(B, -1, -1), emb),
            torch.where(drop[:, None], null_keep[None].expand(B, -1), keep))


def mmdit_forward(P: dict[str, Tensor], x: Tensor, t: Tensor, ctx: Tensor, keep: Tensor | None, cfg: JointConfig,
                  taps: dict[str, Tensor] | None = None) -> Tensor:
    """mmdit.py:789-851,903-928 with the context already passed through the embedder (ctx [B, Lc, context_dim], keep bool
    [B, Lc] = attn_mask)"""
    tok, gh, gw = odit.patchify(P, x, cfg)
    emb = odit.cond_embedding(P, t, None, cfg)
    c = ctx @ P["context_embed.weight"].t()
    cos, sin = rope_tables_joint(c.shape[1], gh, gw, cfg.rope_axes_dim, cfg.rope_base)
    nj = cfg.depth - cfg.n_single_stream_blocks
    for i in range(cfg.depth):
        if i < nj:
            tok, c = joint_block(P, f"layers.{i}.", tok, c, emb, cos, sin, keep, cfg)
        else:
            from .sprint import single_stream_block  # (oracle.sprint imports this module)

            tok, c = single_stream_block(P, f"layers.{i}.", tok, c, emb, cos, sin, keep, cfg)
        if taps is not None:
            taps[f"layer{i}"], taps[f"context{i}"] = tok, c
    return odit.unpatchify(odit.last_layer(P, tok, emb, cfg), gh, gw, cfg)
