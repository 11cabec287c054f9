"""``DDT`` denoiser, MI355X build -- drop-in for ``diffulab.networks.denoisers.ddt.DDT``: ``simple_ddt=True`` (the shipped
``configs/model/ddt.yaml``) and the joint text-image encoder form (``simple_ddt=False`` behind a one-output context embedder,
``n_single_stream_blocks == 0``; ``configs/train_imagenet_repa_txt_to_img.yaml``).  Same constructor kwargs (ddt.py:66-86), ``forward`` kwargs (ddt.py:466-475), ``state_dict`` keys
(``conv_proj_encoder`` / ``conv_proj_decoder``, ``layers`` / ``decoder_layers``) and initialisation (ddt.py:222-230).  The module
owns the parameters; the arithmetic is ``diffulab_amd.ddt_engine.DDTEngine`` / ``DDTJointEngine``.
"""

from __future__ import annotations

import logging
from typing import Any

import torch
import torch.nn as nn
from torch import Tensor

from ...ddt_engine import DDTDims, DDTEngine, DDTJointDims, DDTJointEngine
from .common import FlatArenaDenoiser, ModelOutput
from ...diffuse.utils import to_device
from .mmdit import DiTBlock, MMDiT, MMDiTBlock, _LabelEmbed, _LastLayer


class DDT(FlatArenaDenoiser):
    cfg_pair_capable = True  # `p` only reaches the label drop: guided sampler steps batch their two forwards (forward_cfg_pair)

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
        if simple_ddt and context_embedder is not None:
            raise NotImplementedError("diffulab_amd.DDT: simple_ddt=True takes class labels, not a context embedder")
        if n_single_stream_blocks > 0:
            if not simple_ddt:
                raise NotImplementedError("diffulab_amd.DDT: MMDiTSingleStreamBlock encoder layers are not built (n_single_stream_blocks=0)")
            logging.warning("n_single_stream_blocks is ignored when simple_ddt=True. All blocks are single-stream DiT blocks.")
        if encoder_depth < 1 or decoder_depth < 1:
            raise NotImplementedError("diffulab_amd.DDT: encoder and decoder need at least one block each")
        self.simple_ddt = simple_ddt
        self.patch_size = patch_size
        self.input_channels = input_channels
        self.output_channels = output_channels or input_channels
        self.context_embedder = context_embedder
        self.frequency_embedding = frequency_embedding
        self.rope_base = rope_base
        self.n_classes = n_classes
        self.classifier_free = classifier_free
        self.use_checkpoint = use_checkpoint
        heads_dim = inner_dim // num_heads
        if not simple_ddt:
            self._init_joint(inner_dim, num_heads, mlp_ratio, encoder_depth, decoder_depth, rope_axes_dim, partial_rotary_factor, heads_dim)
            return
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

    def _init_joint(self, inner_dim: int, num_heads: int, mlp_ratio: int, encoder_depth: int, decoder_depth: int,
                    rope_axes_dim: list[int] | None, partial_rotary_factor: float, heads_dim: int) -> None:
        """ddt.py:105-131,150-220 with a one-output context embedder"""
        ce = self.context_embedder
        assert ce is not None, "for ddt with text context embedder must be provided"
        assert isinstance(ce.output_size, tuple) and all(isinstance(i, int) for i in ce.output_size), (
            "context_embedder.output_size must be a tuple of integers")
        if ce.n_output != 1:
            raise NotImplementedError("diffulab_amd.DDT: context embedders with a pooled embedding are not built")
        if any(True for _ in ce.parameters()):
            raise NotImplementedError("diffulab_amd.DDT: the context embedder must be parameter-free (precomputed embeddings)")
        self.pooled_embedding = False
        self.mlp_pooled_context = None
        self.context_embed = nn.Linear(ce.output_size[0], inner_dim, bias=False)
        if rope_axes_dim is None:
            rope_axes_dim = [int((partial_rotary_factor * heads_dim) // 3)] * 3
        self.rope_axes_dim = list(rope_axes_dim)
        self.dims = DDTJointDims(input_channels=self.input_channels, output_channels=self.output_channels, inner_dim=inner_dim,
                                 embedding_dim=inner_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, patch_size=self.patch_size,
                                 rope_base=float(self.rope_base), frequency_embedding=self.frequency_embedding, n_classes=None,
                                 classifier_free=self.classifier_free, rope_axes_dim=self.rope_axes_dim,
                                 context_dim=ce.output_size[0], encoder_depth=encoder_depth, decoder_depth=decoder_depth)
        self.dims.validate()
        self.label_embed = None
        self.last_layer = _LastLayer(inner_dim, inner_dim, self.patch_size, self.output_channels)
        self.time_embed = nn.Sequential(nn.Linear(self.frequency_embedding, inner_dim), nn.SiLU(), nn.Linear(inner_dim, inner_dim))
        mk = lambda: nn.Conv2d(self.input_channels, inner_dim, kernel_size=self.patch_size, stride=self.patch_size, bias=False)  # noqa: E731
        self.conv_proj_encoder, self.conv_proj_decoder = mk(), mk()
        self.layers = nn.ModuleList([MMDiTBlock(inner_dim, inner_dim, mlp_ratio) for _ in range(encoder_depth)])
        self.decoder_layers = nn.ModuleList([DiTBlock(inner_dim, inner_dim, mlp_ratio) for _ in range(decoder_depth)])
        self.apply(MMDiT._init_weights)

    @property
    def precisions(self) -> tuple[str, ...]:  # the fp32-class regime exists for the class-conditional form (ddt_engine_f32.py)
        return ("bf16", "fp32") if self.simple_ddt else ("bf16",)

    def _make_engine(self, device: torch.device):
        if self.precision == "fp32":
            from ...ddt_engine_f32 import DDTEngineF32

            return DDTEngineF32(self.dims, device)
        return DDTEngine(self.dims, device) if self.simple_ddt else DDTJointEngine(self.dims, device)

    def _graph_inputs(self, eng) -> tuple:  # (the context tensors are per-call inputs of the joint launch sequence)
        return () if self.simple_ddt else tuple(eng.context)

    def _graph_set_inputs(self, eng, tensors: tuple) -> None:
        if not self.simple_ddt:
            eng.context = tuple(tensors)

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
        if self.simple_ddt and initial_context is not None:
            raise NotImplementedError("simple_ddt has no context stream")
        if intermediate_features:
            raise NotImplementedError("diffulab_amd.DDT: intermediate_features (use forward hooks on .layers[i])")
        if p > 0:
            assert self.classifier_free, (
                "probability of dropping for classifier free guidance is only available if model is set up to be classifier free")
            if self.simple_ddt:
                assert self.n_classes, (
                    "probability of dropping for classifier free guidance is only available if a number of classes is set")
        if x_context is not None:
            x = torch.cat([x, x_context], dim=1)
        eng = self.engine
        dev = eng.dev
        x = x.to(device=dev, dtype=torch.float32).contiguous()
        t = to_device(timesteps, dev, torch.float32)
        y_eff = None
        if not self.simple_ddt:
            assert self.context_embedder is not None, "for MMDiT context embedder must be provided"
            out = self.context_embedder(initial_context, p)
            keep = out.get("attn_mask", None)
            eng.context = (out["embeddings"].to(device=dev), keep.to(device=dev) if keep is not None else None)
        elif self.label_embed is not None:
            assert y is not None, "class-conditional DDT needs labels `y`"
            y_eff = self._effective_labels(y.to(device=dev, dtype=torch.int64), p).contiguous()  # (drop_labels nn.py:149)
        taps = tuple(i for i, layer in enumerate(self.layers) if layer._forward_hooks)
        if not taps:
            return {"x": self._run(x, t, y_eff)}
        pred, *feats = self._run(x, t, y_eff, taps)
        for i, f in zip(taps, feats):
            for hook in list(self.layers[i]._forward_hooks.values()):
                hook(self.layers[i], (), f)
        return {"x": pred}
