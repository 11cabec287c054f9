"""``MMDiT`` denoiser, MI355X build -- drop-in for ``diffulab.networks.denoisers.mmdit.MMDiT`` (simple_dit mode).

Same constructor kwargs (mmdit.py:604-625 of the reference), same ``forward`` kwargs (mmdit.py:903-912), same
``state_dict`` key names / shapes and the same initialisation scheme (mmdit.py:735-745), so Hydra ``_target_``
configs and ``denoiser.pt`` checkpoints carry over.  The ``nn.Module`` tree below only OWNS the parameters
(as views into one flat HBM arena); the arithmetic is the hand-written HIP path driven by
``diffulab_amd.engine.DiTEngine``.  There is no PyTorch/CPU fallback: calling ``forward`` without a GPU or
without ``libdiffulab_hip.so`` raises.
"""

from __future__ import annotations

import logging
from typing import Any

import torch
import torch.nn as nn
from torch import Tensor

from ...engine import DiTDims, DiTEngine
from .common import FlatArenaDenoiser, ModelOutput


class _RMSScale(nn.Module):
    def __init__(self, dim: int) -> None:
        super().__init__()
        self.scale = nn.Parameter(torch.ones(dim))


class _QKNorm(nn.Module):
    def __init__(self, dim: int) -> None:
        super().__init__()
        self.query_norm = _RMSScale(dim)
        self.key_norm = _RMSScale(dim)


class _Attention(nn.Module):
    def __init__(self, dim: int) -> None:
        super().__init__()
        self.qkv = nn.Linear(dim, 3 * dim, bias=False)
        self.qk_norm = _QKNorm(dim)
        self.proj_out = nn.Linear(dim, dim, bias=False)


class _Modulation(nn.Module):
    def __init__(self, emb: int, dim: int) -> None:
        super().__init__()
        self.lin = nn.Linear(emb, 6 * dim, bias=True)


class DiTBlock(nn.Module):
    """parameter container of one adaLN-zero block (reference mmdit.py:213-309); never called on its own"""

    def __init__(self, inner_dim: int, embedding_dim: int, mlp_ratio: int) -> None:
        super().__init__()
        self.modulation = _Modulation(embedding_dim, inner_dim)
        self.norm_1 = nn.LayerNorm(inner_dim)
        self.attention = _Attention(inner_dim)
        self.norm_2 = nn.LayerNorm(inner_dim)
        self.mlp_input = nn.Sequential(
            nn.Linear(inner_dim, mlp_ratio * inner_dim * 2, bias=False),
            nn.Identity(),  # PackedSwiGLU slot: keeps the index of the second linear at 2
            nn.Linear(mlp_ratio * inner_dim, inner_dim, bias=False),
        )

    def forward(self, *a: Any, **k: Any) -> Tensor:
        raise RuntimeError("DiTBlock parameters are consumed by the fused HIP engine; call the MMDiT module instead")


class _LabelEmbed(nn.Module):
    def __init__(self, n_classes: int, dim: int, cfg: bool) -> None:
        super().__init__()
        self.num_classes = n_classes
        self.embedding = nn.Embedding(n_classes + (1 if cfg else 0), dim)


class _LastLayer(nn.Module):
    def __init__(self, emb: int, dim: int, patch: int, out_ch: int) -> None:
        super().__init__()
        self.linear = nn.Linear(dim, patch * patch * out_ch)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(emb, 2 * dim))


class MMDiT(FlatArenaDenoiser):
    def __init__(
        self,
        simple_dit: bool = False,
        input_channels: int = 3,
        output_channels: int | None = None,
        inner_dim: int = 4096,
        embedding_dim: int = 4096,
        num_heads: int = 16,
        mlp_ratio: int = 4,
        patch_size: int = 16,
        depth: int = 38,
        n_single_stream_blocks: int = 0,
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
        if not simple_dit or context_embedder is not None:
            raise NotImplementedError(
                "diffulab_amd.MMDiT: only simple_dit=True (class/unconditional DiT) has a HIP path so far; the joint "
                "text-image MMDiT blocks (reference mmdit.py:107-210,312-532) are the next scope row (DESIGN.md)")
        if n_single_stream_blocks > 0:
            logging.warning("n_single_stream_blocks is ignored when simple_dit=True. All blocks are single-stream DiT blocks.")
        self.simple_dit = True
        self.patch_size = patch_size
        self.input_channels = input_channels
        self.output_channels = output_channels or input_channels
        self.context_embedder = None
        self.frequency_embedding = frequency_embedding
        self.rope_base = rope_base
        self.n_classes = n_classes
        self.classifier_free = classifier_free
        self.use_checkpoint = use_checkpoint  # activations are kept resident in HBM; nothing to checkpoint
        heads_dim = inner_dim // num_heads
        if rope_axes_dim is None:
            rope_axes_dim = [int((partial_rotary_factor * heads_dim) // 2)] * 2
        self.rope_axes_dim = list(rope_axes_dim)
        self.dims = DiTDims(input_channels=input_channels, output_channels=self.output_channels, inner_dim=inner_dim,
                            embedding_dim=embedding_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, patch_size=patch_size,
                            depth=depth, rope_base=float(rope_base), frequency_embedding=frequency_embedding,
                            n_classes=n_classes, classifier_free=classifier_free, rope_axes_dim=self.rope_axes_dim)
        self.dims.validate()

        self.label_embed = _LabelEmbed(n_classes, embedding_dim, classifier_free) if n_classes is not None else None
        self.last_layer = _LastLayer(embedding_dim, inner_dim, patch_size, self.output_channels)
        self.time_embed = nn.Sequential(nn.Linear(frequency_embedding, embedding_dim), nn.SiLU(),
                                        nn.Linear(embedding_dim, embedding_dim))
        self.conv_proj = nn.Conv2d(input_channels, inner_dim, kernel_size=patch_size, stride=patch_size, bias=False)
        self.layers = nn.ModuleList([DiTBlock(inner_dim, embedding_dim, mlp_ratio) for _ in range(depth)])
        self.apply(self._init_weights)

    # reference init: xavier on Linear/Conv2d, zero biases, zero adaLN (mmdit.py:735-745)
    @staticmethod
    def _init_weights(module: nn.Module) -> None:
        if isinstance(module, (nn.Linear, nn.Conv2d)):
            nn.init.xavier_uniform_(module.weight)
            if module.bias is not None:
                nn.init.constant_(module.bias, 0)
        if isinstance(module, _Modulation):
            for p in module.parameters():
                p.detach().zero_()
        if isinstance(module, _LastLayer):
            for p in module.adaLN_modulation.parameters():
                p.detach().zero_()

    def _make_engine(self, device: torch.device) -> DiTEngine:
        return DiTEngine(self.dims, device)

    # ------------------------------------------------------------------ forward (mmdit.py:903-928)
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
        t = timesteps.to(device=dev, dtype=torch.float32).contiguous()  # nn.py:110: timesteps[:, None].float()
        y_eff = None
        if self.label_embed is not None:
            assert y is not None, "class-conditional DiT needs labels `y`"
            y_eff = y.to(device=dev, dtype=torch.int64)
            if p > 0:  # LabelEmbed.drop_labels nn.py:149 -- torch device RNG, same draw as the reference
                y_eff = torch.where(torch.rand(y_eff.size(), device=dev) < p, self.n_classes, y_eff)
            y_eff = y_eff.contiguous()
        # forward hooks registered on a block (RePA: ``denoiser.layers[i].register_forward_hook``, repa.py:133-134) receive the
        # block's output as an extra differentiable output of the engine
        taps = tuple(i for i, layer in enumerate(self.layers) if layer._forward_hooks)
        if not taps:
            return {"x": self._run(x, t, y_eff)}
        pred, *feats = self._run(x, t, y_eff, taps)
        for i, f in zip(taps, feats):
            for hook in list(self.layers[i]._forward_hooks.values()):
                hook(self.layers[i], (), f)
        return {"x": pred}
