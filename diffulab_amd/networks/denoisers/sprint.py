"""``SprintDiT`` denoiser, MI355X build -- drop-in for ``diffulab.networks.denoisers.sprint.SprintDiT``: the class-conditional
``simple_dit=True`` form (the shipped ``configs/model/sprint.yaml``) and the joint text-image form (``simple_dit=False`` behind a
one-output context embedder such as ``PrecomputedEmbedder``: MMDiTBlock encoder / decoder, MMDiTBlock + MMDiTSingleStreamBlock deep
stage, ``fuse_context``; ``configs/train_imagenet_repa_txt_to_img_sprint.yaml``).  Same constructor kwargs (sprint.py:68-91), ``forward`` kwargs (sprint.py:575-584),
``state_dict`` keys (``mask_token``, ``fuse``, ``layers`` / ``deep_layers`` / ``decoder_layers``) and initialisation.

The module owns the parameters (views of one flat arena) and decides the token routing of a step with torch device ops, exactly
where the reference draws its random numbers (label drop nn.py:149, token scores sprint.py:343, path drop sprint.py:384):
``_draw_label_drop`` / ``_draw_scores`` / ``_draw_path_drop`` are separate methods so tests can inject recorded draws.  Everything
else is the hand-written HIP path of ``diffulab_amd.sprint_engine.SprintEngine`` / ``sprint_joint_engine.SprintJointEngine``.
"""

from __future__ import annotations

import logging
from typing import Any

import torch
import torch.nn as nn
from torch import Tensor

from ...sprint_engine import Route, SprintDims, SprintEngine
from ...sprint_joint_engine import SprintJointDims, SprintJointEngine
from .common import FlatArenaDenoiser, ModelOutput
from ...diffuse.utils import to_device
from .mmdit import DiTBlock, MMDiT, MMDiTBlock, MMDiTSingleStreamBlock, _LabelEmbed, _LastLayer


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
        assert not (n_classes is not None and context_embedder is not None), "n_classes and context_embedder cannot both be specified"
        if simple_dit and context_embedder is not None:
            raise NotImplementedError("diffulab_amd.SprintDiT: simple_dit=True takes class labels, not a context embedder")
        if simple_dit and n_single_stream_blocks > 0:
            raise NotImplementedError("diffulab_amd.SprintDiT: n_single_stream_blocks > 0 (with simple_dit=True the reference "
                                      "replaces the deep stack by MMDiTSingleStreamBlocks, sprint.py:147-151) is not built")
        if encoder_depth < 1 or deep_layers_depth < 1 or decoder_depth < 1 or n_single_stream_blocks > deep_layers_depth:
            raise NotImplementedError("diffulab_amd.SprintDiT: every stage needs at least one block")
        self.simple_dit = simple_dit
        self.patch_size = patch_size
        self.input_channels = input_channels
        self.output_channels = output_channels or input_channels
        self.context_embedder = context_embedder
        self.frequency_embedding = frequency_embedding
        self.rope_base = rope_base
        self.n_classes = n_classes
        self.classifier_free = classifier_free
        self.drop_rate = drop_rate
        self.use_checkpoint = use_checkpoint
        heads_dim = inner_dim // num_heads
        self._eval_routes: dict[tuple, Route] = {}
        if not simple_dit:
            self._init_joint(inner_dim, embedding_dim, num_heads, mlp_ratio, encoder_depth, deep_layers_depth, n_single_stream_blocks,
                             decoder_depth, rope_axes_dim, partial_rotary_factor, heads_dim)
            return
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

    def _init_joint(self, inner_dim: int, embedding_dim: int, num_heads: int, mlp_ratio: int, encoder_depth: int,
                    deep_layers_depth: int, n_single: int, decoder_depth: int, rope_axes_dim: list[int] | None,
                    partial_rotary_factor: float, heads_dim: int) -> None:
        """sprint.py:109-131,160-262 with a one-output context embedder"""
        ce = self.context_embedder
        assert ce is not None, "for dit with text context embedder must be provided"
        assert isinstance(ce.output_size, tuple) and all(isinstance(i, int) for i in ce.output_size), (
            "context_embedder.output_size must be a tuple of integers")
        if ce.n_output != 1:
            raise NotImplementedError("diffulab_amd.SprintDiT: context embedders with a pooled embedding are not built")
        if any(True for _ in ce.parameters()):
            raise NotImplementedError("diffulab_amd.SprintDiT: the context embedder must be parameter-free (precomputed embeddings)")
        self.pooled_embedding = False
        self.mlp_pooled_context = None
        self.context_embed = nn.Linear(ce.output_size[0], inner_dim, bias=False)
        if rope_axes_dim is None:
            rope_axes_dim = [int((partial_rotary_factor * heads_dim) // 3)] * 3
        self.rope_axes_dim = list(rope_axes_dim)
        self.dims = SprintJointDims(input_channels=self.input_channels, output_channels=self.output_channels, inner_dim=inner_dim,
                                    embedding_dim=embedding_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, patch_size=self.patch_size,
                                    rope_base=float(self.rope_base), frequency_embedding=self.frequency_embedding, n_classes=None,
                                    classifier_free=self.classifier_free, rope_axes_dim=self.rope_axes_dim,
                                    context_dim=ce.output_size[0], encoder_depth=encoder_depth, deep_layers_depth=deep_layers_depth,
                                    n_single_stream_blocks=n_single, decoder_depth=decoder_depth, drop_rate=self.drop_rate)
        self.dims.validate()
        self.mask_token = nn.Parameter(torch.zeros(1, 1, inner_dim))
        self.time_embed = nn.Sequential(nn.Linear(self.frequency_embedding, embedding_dim), nn.SiLU(),
                                        nn.Linear(embedding_dim, embedding_dim))
        self.conv_proj = nn.Conv2d(self.input_channels, inner_dim, kernel_size=self.patch_size, stride=self.patch_size, bias=False)
        self.fuse = nn.Linear(inner_dim * 2, inner_dim, bias=False)
        self.fuse_context = nn.Linear(2 * inner_dim, inner_dim, bias=False)
        self.last_layer = _LastLayer(embedding_dim, inner_dim, self.patch_size, self.output_channels)
        jb = lambda: MMDiTBlock(inner_dim, embedding_dim, mlp_ratio)  # noqa: E731
        self.layers = nn.ModuleList([jb() for _ in range(encoder_depth)])
        self.deep_layers = nn.ModuleList([jb() for _ in range(deep_layers_depth - n_single)]
                                         + [MMDiTSingleStreamBlock(inner_dim, embedding_dim, mlp_ratio) for _ in range(n_single)])
        self.decoder_layers = nn.ModuleList([jb() for _ in range(decoder_depth)])
        self.apply(MMDiT._init_weights)

    @property
    def precisions(self) -> tuple[str, ...]:  # the fp32-class regime exists for the class-conditional form
        return ("bf16", "fp32") if self.simple_dit else ("bf16",)

    def _make_engine(self, device: torch.device):
        if self.precision == "fp32":
            from ...sprint_engine_f32 import SprintEngineF32

            return SprintEngineF32(self.dims, device)
        return SprintEngine(self.dims, device) if self.simple_dit else SprintJointEngine(self.dims, device)

    # ------------------------------------------------------------------ random decisions of a step (device RNG, as the reference)
    def _draw_label_drop(self, y: Tensor, p: float) -> Tensor:
        return torch.where(torch.rand(y.size(), device=y.device) < p, self.n_classes, y)  # nn.py:149

    def _draw_scores(self, B: int, S: int, device: torch.device) -> Tensor:
        return torch.rand((B, S), device=device, dtype=torch.float32)  # sprint.py:343

    def _draw_path_drop(self, B: int, p: float, device: torch.device) -> Tensor:
        return torch.rand(B, device=device) < p  # sprint.py:384

    def _route(self, B: int, S: int, p: float, device: torch.device, n_ctx: int = 0) -> Route:
        """drop_tokens / restore_tokens bookkeeping (sprint.py:317-387) as index tensors for the routing kernels"""
        train = self.training
        k = self.dims.n_kept(S) if train else S
        static = not train and not (0 < p < 1)
        key = (B, S, p >= 1, str(device), n_ctx)
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
                if n_ctx:  # joint form: table rows of [text ; kept image] (text rows 0 .. n_ctx-1, image rows n_ctx + position)
                    r.pos_lat = torch.cat((torch.arange(n_ctx, device=device, dtype=torch.int32).expand(B, n_ctx),
                                           r.idx + n_ctx), dim=1).contiguous().view(-1)
        if static:
            self._eval_routes[key] = r
        return r

    # hipGraph replay: an eval-mode routing without a random path drop is one of the cached Route objects (persistent tensors),
    # so it is part of the capture key; a routing drawn for this call (0 < p < 1) runs eagerly
    def _graph_key(self, eng) -> tuple | None:
        r = eng.route
        return (id(r), r.skip_deep) if any(r is c for c in self._eval_routes.values()) else None

    def _graph_inputs(self, eng) -> tuple:
        return () if self.simple_dit else tuple(eng.context)

    def _graph_set_inputs(self, eng, tensors: tuple) -> None:
        if not self.simple_dit:
            eng.context = tuple(tensors)

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
        if self.simple_dit and initial_context is not None:
            raise NotImplementedError("simple_dit has no context stream")
        if intermediate_features:
            raise NotImplementedError("diffulab_amd.SprintDiT: intermediate_features (use forward hooks on .layers[i])")
        if p > 0:
            assert self.classifier_free, (
                "probability of dropping for classifier free guidance is only available if model is set up to be classifier free")
            if self.simple_dit:
                assert self.n_classes, (
                    "probability of dropping for classifier free guidance is only available if a number of classes is set")
        if x_context is not None:
            x = torch.cat([x, x_context], dim=1)
        eng = self.engine
        dev = eng.dev
        x = x.to(device=dev, dtype=torch.float32).contiguous()
        t = to_device(timesteps, dev, torch.float32)
        y_eff = None
        if not self.simple_dit:  # sprint.py:411-424: the embedder (context drop) first, then the routing draws
            assert self.context_embedder is not None, "for MMDiT context embedder must be provided"
            out = self.context_embedder(initial_context, p)
            keep = out.get("attn_mask", None)
            eng.context = (out["embeddings"].to(device=dev), keep.to(device=dev) if keep is not None else None)
            B, _, H, W = x.shape
            S = (H // self.patch_size) * (W // self.patch_size)
            eng.route = self._route(B, S, float(p), dev, n_ctx=out["embeddings"].shape[1])
        elif self.label_embed is not None:
            assert y is not None, "class-conditional DiT needs labels `y`"
            y_eff = y.to(device=dev, dtype=torch.int64)
            if p > 0:
                y_eff = self._draw_label_drop(y_eff, p)
            y_eff = y_eff.contiguous()
        if self.simple_dit:
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
