"""``SprintDiT`` denoiser, MI355X build -- drop-in for ``diffulab.networks.denoisers.sprint.SprintDiT`` with ``simple_dit=True``
(the shipped ``configs/model/sprint.yaml``): same constructor kwargs (sprint.py:68-91), ``forward`` kwargs (sprint.py:575-584),
``state_dict`` keys (``mask_token``, ``fuse``, ``layers`` / ``deep_layers`` / ``decoder_layers``) and initialisation.

The module owns the parameters (views of one flat arena) and decides the token routing of a step with torch device ops, exactly
where the reference draws its random numbers (label drop nn.py:149, token scores sprint.py:343, path drop sprint.py:384):
``_draw_label_drop`` / ``_draw_scores`` / ``_draw_path_drop`` are separate methods so tests can inject recorded draws.  Everything
else is the hand-written HIP path of ``diffulab_amd.sprint_engine.SprintEngine``.
"""

from __future__ import annotations

import logging
from typing import Any

import torch
import torch.nn as nn
from torch import Tensor

from ...sprint_engine import Route, SprintDims, SprintEngine
from .common import FlatArenaDenoiser, ModelOutput
from .mmdit import DiTBlock, MMDiT, _LabelEmbed, _LastLayer


class SprintDiT(FlatArenaDenoiser):
    def __init__(
        self,
        simple_dit: bool = False,
        input_channels: int = 3,
        output_channels: int | None = None,
        inner_dim: int = 768,
        embedding_dim: int = 768,
        num_heads: int = 12,
        mlp_ratio: int = 4,
        patch_size: int = 16,
        encoder_depth: int = 2,
        deep_layers_depth: int = 8,
        n_single_stream_blocks: int = 0,
        decoder_depth: int = 2,
        rope_base: int = 10_000,
        partial_rotary_factor: float = 1,
        rope_axes_dim: list[int] | None = None,
        frequency_embedding: int = 256,
        n_classes: int | None = None,
        classifier_free: bool = False,
        context_embedder: Any | None = None,
        use_checkpoint: bool = False,
        drop_rate: float = 0.75,
    ) -> None:
        super().__init__()
        if not simple_dit or context_embedder is not None:
            raise NotImplementedError("diffulab_amd.SprintDiT: only simple_dit=True has a HIP path so far (the joint text-image "
                                      "blocks, sprint.py:389-502, are the next scope row)")
        if n_single_stream_blocks > 0:
            raise NotImplementedError("diffulab_amd.SprintDiT: n_single_stream_blocks > 0 (with simple_dit=True the reference "
                                      "replaces the deep stack by MMDiTSingleStreamBlocks, sprint.py:147-151) is not built")
        if encoder_depth < 1 or deep_layers_depth < 1 or decoder_depth < 1:
            raise NotImplementedError("diffulab_amd.SprintDiT: every stage needs at least one block")
        self.simple_dit = True
        self.patch_size = patch_size
        self.input_channels = input_channels
        self.output_channels = output_channels or input_channels
        self.context_embedder = None
        self.frequency_embedding = frequency_embedding
        self.rope_base = rope_base
        self.n_classes = n_classes
        self.classifier_free = classifier_free
        self.drop_rate = drop_rate
        self.use_checkpoint = use_checkpoint
        heads_dim = inner_dim // num_heads
        if rope_axes_dim is None:
            rope_axes_dim = [int((partial_rotary_factor * heads_dim) // 2)] * 2
        self.rope_axes_dim = list(rope_axes_dim)
        self.dims = SprintDims(input_channels=input_channels, output_channels=self.output_channels, inner_dim=inner_dim,
                               embedding_dim=embedding_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, patch_size=patch_size,
                               rope_base=float(rope_base), frequency_embedding=frequency_embedding, n_classes=n_classes,
                               classifier_free=classifier_free, rope_axes_dim=self.rope_axes_dim, encoder_depth=encoder_depth,
                               deep_layers_depth=deep_layers_depth, decoder_depth=decoder_depth, drop_rate=drop_rate)
        self.dims.validate()

        self.mask_token = nn.Parameter(torch.zeros(1, 1, inner_dim))
        self.label_embed = _LabelEmbed(n_classes, embedding_dim, classifier_free) if n_classes is not None else None
        self.time_embed = nn.Sequential(nn.Linear(frequency_embedding, embedding_dim), nn.SiLU(),
                                        nn.Linear(embedding_dim, embedding_dim))
        self.conv_proj = nn.Conv2d(input_channels, inner_dim, kernel_size=patch_size, stride=patch_size, bias=False)
        self.fuse = nn.Linear(inner_dim * 2, inner_dim, bias=False)
        self.last_layer = _LastLayer(embedding_dim, inner_dim, patch_size, self.output_channels)
        mk = lambda n: nn.ModuleList([DiTBlock(inner_dim, embedding_dim, mlp_ratio) for _ in range(n)])  # noqa: E731
        self.layers = mk(encoder_depth)  # (name compatibility for RePA: hooks attach to the encoder blocks)
        self.deep_layers = mk(deep_layers_depth)
        self.decoder_layers = mk(decoder_depth)
        self.apply(MMDiT._init_weights)
        self._eval_routes: dict[tuple, Route] = {}

    def _make_engine(self, device: torch.device) -> SprintEngine:
        return SprintEngine(self.dims, device)

    # ------------------------------------------------------------------ random decisions of a step (device RNG, as the reference)
    def _draw_label_drop(self, y: Tensor, p: float) -> Tensor:
        return torch.where(torch.rand(y.size(), device=y.device) < p, self.n_classes, y)  # nn.py:149

    def _draw_scores(self, B: int, S: int, device: torch.device) -> Tensor:
        return torch.rand((B, S), device=device, dtype=torch.float32)  # sprint.py:343

    def _draw_path_drop(self, B: int, p: float, device: torch.device) -> Tensor:
        return torch.rand(B, device=device) < p  # sprint.py:384

    def _route(self, B: int, S: int, p: float, device: torch.device) -> Route:
        """drop_tokens / restore_tokens bookkeeping (sprint.py:317-387) as index tensors for the routing kernels"""
        train = self.training
        k = self.dims.n_kept(S) if train else S
        static = not train and not (0 < p < 1)
        key = (B, S, p >= 1, str(device))
        if static and key in self._eval_routes:
            return self._eval_routes[key]
        with torch.inference_mode(False), torch.no_grad():
            if p >= 1:  # deep layers skipped: the canvas is the mask token everywhere (sprint.py:556-557)
                r = Route(None, torch.full((B, S), -1, device=device, dtype=torch.int32), None, k, skip_deep=True)
            else:
                if train:
                    scores = self._draw_scores(B, S, device)
                    idx = torch.topk(scores, k=k, dim=1, largest=True, sorted=False).indices
                    idx = torch.sort(idx, dim=1).values
                else:
                    idx = torch.arange(S, device=device).expand(B, S)
                inv = torch.full((B, S), -1, device=device, dtype=torch.int32)
                inv.scatter_(1, idx, torch.arange(k, device=device, dtype=torch.int32).expand(B, k))
                keep = None
                if p > 0:
                    drop = self._draw_path_drop(B, p, device)
                    keep = (~drop).to(torch.int32).contiguous()
                    inv[drop] = -1
                r = Route(idx.to(torch.int32).contiguous(), inv.contiguous(), keep, k)
        if static:
            self._eval_routes[key] = r
        return r

    def _infer(self, eng, x: Tensor, t: Tensor, y_eff: Tensor | None) -> Tensor:  # eager (the routing is an input of the sequence)
        return eng.forward(x, t, y_eff, train=False).clone()

    # ------------------------------------------------------------------ forward (sprint.py:575-624)
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
            raise NotImplementedError("simple_dit has no context stream")
        if intermediate_features:
            raise NotImplementedError("diffulab_amd.SprintDiT: intermediate_features (use forward hooks on .layers[i])")
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
            assert y is not None, "class-conditional DiT needs labels `y`"
            y_eff = y.to(device=dev, dtype=torch.int64)
            if p > 0:
                y_eff = self._draw_label_drop(y_eff, p)
            y_eff = y_eff.contiguous()
        B, _, H, W = x.shape
        S = (H // self.patch_size) * (W // self.patch_size)
        eng.route = self._route(B, S, float(p), dev)
        taps = tuple(i for i, layer in enumerate(self.layers) if layer._forward_hooks)
        if not taps:
            return {"x": self._run(x, t, y_eff)}
        pred, *feats = self._run(x, t, y_eff, taps)
        for i, f in zip(taps, feats):
            for hook in list(self.layers[i]._forward_hooks.values()):
                hook(self.layers[i], (), f)
        return {"x": pred}
