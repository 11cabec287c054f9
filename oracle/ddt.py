"""CPU restatement of DDT with simple_ddt=True (reference networks/denoisers/ddt.py; the shipped configs/model/ddt.yaml) -- TEST
INFRASTRUCTURE ONLY.

Encoder: DiT blocks on conv_proj_encoder(x) conditioned on time (+ label) embedding.  Decoder: DiT blocks on conv_proj_decoder(x)
whose adaLN conditioning is PER TOKEN: z = silu(encoder_output + time_embedding[:, None, :]) (ddt.py:423-424), so every block's
Modulation (and the last layer's) maps [B, S, D] -> [B, S, 6D] ([B, S, 2D]).
Pinned by tests/golden/ddt.npz.
"""

from __future__ import annotations

from dataclasses import dataclass

from torch import Tensor

from . import dit as odit


@dataclass
class DDTConfig(odit.DiTConfig):
    encoder_depth: int = 8
    decoder_depth: int = 4

    def __post_init__(self) -> None:
        self.embedding_dim = self.inner_dim  # ddt.py:139-176: every embedding has the token width
        super().__post_init__()


def param_shapes(cfg: DDTConfig) -> dict[str, tuple[int, ...]]:
    D = cfg.inner_dim
    one = odit.param_shapes(odit.DiTConfig(**{**{k: getattr(cfg, k) for k in odit.DiTConfig.__dataclass_fields__}, "depth": 1}))
    s = {k: v for k, v in one.items() if not k.startswith("layers.") and k != "conv_proj.weight"}
    s["conv_proj_encoder.weight"] = s["conv_proj_decoder.weight"] = one["conv_proj.weight"]
    for name, depth in (("layers", cfg.encoder_depth), ("decoder_layers", cfg.decoder_depth)):
        for i in range(depth):
            s.update({f"{name}.{i}." + k[len("layers.0."):]: v for k, v in one.items() if k.startswith("layers.0.")})
    assert s["time_embed.2.weight"] == (D, D)
    return s


def ddt_forward(P: dict[str, Tensor], x: Tensor, t: Tensor, y_eff: Tensor | None, cfg: DDTConfig,
                taps: dict[str, Tensor] | None = None) -> Tensor:
    """ddt.py:466-512 with simple_ddt=True"""
    enc_in, gh, gw = odit.patchify({"conv_proj.weight": P["conv_proj_encoder.weight"]}, x, cfg)
    dec, _, _ = odit.patchify({"conv_proj.weight": P["conv_proj_decoder.weight"]}, x, cfg)
    emb = odit.cond_embedding(P, t, y_eff, cfg)
    cos, sin = odit.rope_tables(gh, gw, cfg.rope_axes_dim, cfg.rope_base)
    h = enc_in
    for i in range(cfg.encoder_depth):
        h = odit.dit_block(P, f"layers.{i}.", h, emb, cos, sin, cfg)
        if taps is not None:
            taps[f"layer{i}"] = h
    temb = odit.cond_embedding(P, t, None, cfg)  # ddt.py:423: the time embedding alone (no label row)
    z = odit.silu(h + temb[:, None, :])
    for i in range(cfg.decoder_depth):
        dec = odit.dit_block(P, f"decoder_layers.{i}.", dec, z, cos, sin, cfg)
    return odit.unpatchify(odit.last_layer(P, dec, z, cfg), gh, gw, cfg)


# ------------------------------------------------------------------------------------------------ joint-encoder form
from . import mmdit as ommdit  # noqa: E402


@dataclass
class DDTJointConfig(ommdit.JointConfig):
    encoder_depth: int = 8
    decoder_depth: int = 4

    def __post_init__(self) -> None:
        self.embedding_dim = self.inner_dim
        super().__post_init__()


def joint_param_shapes(cfg: DDTJointConfig) -> dict[str, tuple[int, ...]]:
    """state_dict layout of DDT(simple_ddt=False) with a one-output context embedder and n_single_stream_blocks = 0"""
    kw = {k: getattr(cfg, k) for k in ommdit.JointConfig.__dataclass_fields__}
    enc = ommdit.param_shapes(ommdit.JointConfig(**{**kw, "depth": cfg.encoder_depth, "n_single_stream_blocks": 0}))
    dit = odit.param_shapes(odit.DiTConfig(**{**{k: getattr(cfg, k) for k in odit.DiTConfig.__dataclass_fields__}, "depth": 1,
                                               "n_classes": None}))
    s = {k: v for k, v in enc.items() if k != "conv_proj.weight"}
    s["conv_proj_encoder.weight"] = s["conv_proj_decoder.weight"] = enc["conv_proj.weight"]
    for i in range(cfg.decoder_depth):
        s.update({f"decoder_layers.{i}." + k[len("layers.0."):]: v for k, v in dit.items() if k.startswith("layers.0.")})
    return s


def ddt_joint_forward(P: dict[str, Tensor], x: Tensor, t: Tensor, ctx: Tensor, keep: Tensor | None, cfg: DDTJointConfig) -> Tensor:
    """ddt.py:274-344 (encode_mmddt) + :404-464 (decode) with the context already through the embedder"""
    enc_in, gh, gw = odit.patchify({"conv_proj.weight": P["conv_proj_encoder.weight"]}, x, cfg)
    dec, _, _ = odit.patchify({"conv_proj.weight": P["conv_proj_decoder.weight"]}, x, cfg)
    emb = odit.cond_embedding(P, t, None, cfg)
    c = ctx @ P["context_embed.weight"].t()
    Lc = c.shape[1]
    cos, sin = ommdit.rope_tables_joint(Lc, gh, gw, cfg.rope_axes_dim, cfg.rope_base)
    h = enc_in
    for i in range(cfg.encoder_depth):
        h, c = ommdit.joint_block(P, f"layers.{i}.", h, c, emb, cos, sin, keep, cfg)
    z = odit.silu(h + emb[:, None, :])
    ci, si = cos[Lc:], sin[Lc:]  # decoder position ids (0, h, w) = the image rows of the joint table
    for i in range(cfg.decoder_depth):
        dec = odit.dit_block(P, f"decoder_layers.{i}.", dec, z, ci, si, cfg)
    return odit.unpatchify(odit.last_layer(P, dec, z, cfg), gh, gw, cfg)
