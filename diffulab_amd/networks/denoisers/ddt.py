"""``DDT`` denoiser, MI355X build -- drop-in for ``diffulab.networks.denoisers.ddt.DDT`` with ``simple_ddt=True`` (the shipped
``configs/model/ddt.yaml``): same constructor kwargs (ddt.py:66-86), ``forward`` kwargs (ddt.py:466-475), ``state_dict`` keys
(``conv_proj_encoder`` / ``conv_proj_decoder``, ``layers`` / ``decoder_layers``) and initialisation (ddt.py:222-230).  The module
owns the parameters; the arithmetic is ``diffulab_amd.ddt_engine.DDTEngine``.
"""

from __future__ import annotations

import logging
from typing import Any

import torch
import torch.nn as nn
from torch import Tensor

from ...ddt_engine import DDTDims, DDTEngine
from .common import FlatArenaDenoiser, ModelOutput
from .mmdit import DiTBlock, MMDiT, _LabelEmbed, _LastLayer


class DDT(FlatArenaDenoiser):
    def __init__(
        self,
        simple_ddt: bool = False,
        input_channels: int = 3,
        output_channels: int | None = None,
        inner_dim: int = 768,
        num_heads: int = 12,
        mlp_ratio: int = 4,
        patch_size: int = 16,
        encoder_depth: int = 8,
        n_single_stream_blocks: int = 0,
        decoder_depth: int = 4,
        rope_base: int = 10_000,
        partial_rotary_factor: float = 1,
        rope_axes_dim: list[int] | None = None,
        frequency_embedding: int = 256,
        n_classes: int | None = None,
        classifier_free: bool = False,
        context_embedder: Any | None = None,
        use_checkpoint: bool = False,
    ) -> None:
        super().__init__()
        assert not (n_classes is not None and context_embedder is not None), "n_classes and context_embedder cannot both be specified"
        assert n_single_stream_blocks < encoder_depth, "n_single_stream_blocks must be less than encoder_depth"
        if not simple_ddt or context_embedder is not None:
            raise NotImplementedError("diffulab_amd.DDT: only simple_ddt=True (DiT encoder, class labels) has a HIP path; the joint "
                                      "text-image encoder (ddt.py:274-344) is not built")
        if n_single_stream_blocks > 0:
            logging.warning("n_single_stream_blocks is ignored when simple_ddt=True. All blocks are single-stream DiT blocks.")
        if encoder_depth < 1 or decoder_depth < 1:
            raise NotImplementedError("diffulab_amd.DDT: encoder and decoder need at least one block each")
        self.simple_ddt = True
        self.patch_size = patch_size
        self.input_channels = input_channels
        self.output_channels = output_channels or input_channels
        self.context_embedder = None
        self.frequency_embedding = frequency_embedding
        self.rope_base = rope_base
        self.n_classes = n_classes
        self.classifier_free = classifier_free
        self.use_checkpoint = use_checkpoint
        heads_dim = inner_dim // num_heads
        if rope_axes_dim is None:
            rope_axes_dim = [int((partial_rotary_factor * heads_dim) // 2)] * 2
        self.rope_axes_dim = list(rope_axes_dim)
        self.dims = DDTDims(input_channels=input_channels, output_channels=self.output_channels, inner_dim=inner_dim,
                            embedding_dim=inner_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, patch_size=patch_size,
                            rope_base=float(rope_base), frequency_embedding=frequency_embedding, n_classes=n_classes,
                            classifier_free=classifier_free, rope_axes_dim=self.rope_axes_dim, encoder_depth=encoder_depth,
                            decoder_depth=decoder_depth)
        self.dims.validate()
        self.label_embed = _LabelEmbed(n_classes, inner_dim, classifier_free) if n_classes is not None else None
        self.last_layer = _LastLayer(inner_dim, inner_dim, patch_size, self.output_channels)
        self.time_embed = nn.Sequential(nn.Linear(frequency_embedding, inner_dim), nn.SiLU(), nn.Linear(inner_dim, inner_dim))
        self.conv_proj_encoder = nn.Conv2d(input_channels, inner_dim, kernel_size=patch_size, stride=patch_size, bias=False)
        self.conv_proj_decoder = nn.Conv2d(input_channels, inner_dim, kernel_size=patch_size, stride=patch_size, bias=False)
        self.layers = nn.ModuleList([DiTBlock(inner_dim, inner_dim, mlp_ratio) for _ in range(encoder_depth)])
        self.decoder_layers = nn.ModuleList([DiTBlock(inner_dim, inner_dim, mlp_ratio) for _ in range(decoder_depth)])
        self.apply(MMDiT._init_weights)

    def _make_engine(self, device: torch.device) -> DDTEngine:
        return DDTEngine(self.dims, device)

    def forward(
        self,
        x: Tensor,
        timesteps: Tensor,
        initial_context: Any | None = None,
        p: float = 0.0,
        y: Tensor | None = None,
        x_context: Tensor | None = None,
        intermediate_features: bool = False,
    ) -> ModelOutput:
        assert not (initial_context is not None and y is not None), "initial_context and y cannot both be specified"
        if initial_context is not None:
            raise NotImplementedError("simple_ddt has no context stream")
        if intermediate_features:
            raise NotImplementedError("diffulab_amd.DDT: intermediate_features (use forward hooks on .layers[i])")
        if p > 0:
            assert self.classifier_free, (
                "probability of dropping for classifier free guidance is only available if model is set up to be classifier free")
            assert self.n_classes, (
                "probability of dropping for classifier free guidance is only available if a number of classes is set")
        if x_context is not None:
            x = torch.cat([x, x_context], dim=1)
        eng = self.engine
        dev = eng.dev
        x = x.to(device=dev, dtype=torch.float32).contiguous()
        t = timesteps.to(device=dev, dtype=torch.float32).contiguous()
        y_eff = None
        if self.label_embed is not None:
            assert y is not None, "class-conditional DDT needs labels `y`"
            y_eff = y.to(device=dev, dtype=torch.int64)
            if p > 0:  # LabelEmbed.drop_labels nn.py:149
                y_eff = torch.where(torch.rand(y_eff.size(), device=dev) < p, self.n_classes, y_eff)
            y_eff = y_eff.contiguous()
        taps = tuple(i for i, layer in enumerate(self.layers) if layer._forward_hooks)
        if not taps:
            return {"x": self._run(x, t, y_eff)}
        pred, *feats = self._run(x, t, y_eff, taps)
        for i, f in zip(taps, feats):
            for hook in list(self.layers[i]._forward_hooks.values()):
                hook(self.layers[i], (), f)
        return {"x": pred}
