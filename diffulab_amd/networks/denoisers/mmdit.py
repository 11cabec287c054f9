"""``MMDiT`` denoiser, MI355X build -- drop-in for ``diffulab.networks.denoisers.mmdit.MMDiT`` (simple_dit mode).

Same constructor kwargs (mmdit.py:604-625 of the reference), same ``forward`` kwargs (mmdit.py:903-912), same
``state_dict`` key names / shapes and the same initialisation scheme (mmdit.py:735-745), so Hydra ``_target_``
configs and ``denoiser.pt`` checkpoints carry over.  The ``nn.Module`` tree below only OWNS the parameters
(as views into one flat HBM arena); the arithmetic is the hand-written HIP path driven by
``diffulab_amd.engine.DiTEngine``.  There is no PyTorch/CPU fallback: calling ``forward`` without a GPU or
without ``libdiffulab_hip.so`` raises.
"""

from __future__ import annotations

import copy
import logging
from typing import Any

import torch
import torch.nn as nn
from torch import Tensor

from ...engine import DiTDims, DiTEngine
from .common import Denoiser, ModelOutput


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


class _DiTFn(torch.autograd.Function):
    """autograd seam: forward/backward of the whole denoiser are the engine's launch sequences; parameter
    gradients are accumulated straight into the flat gradient arena (p.grad are views of it)."""

    @staticmethod
    def forward(ctx, module: "MMDiT", x: Tensor, t: Tensor, y_eff: Tensor | None, anchor: Tensor) -> Tensor:
        ctx.module = module
        ctx.set_materialize_grads(False)
        return module._engine.forward(x, t, y_eff, train=True).clone()

    @staticmethod
    def backward(ctx, dpred: Tensor | None):
        m: MMDiT = ctx.module
        if dpred is not None:
            m._prepare_grads()
            m._engine.backward(dpred.contiguous().float())
        return None, None, None, None, None


class MMDiT(Denoiser):
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
        object.__setattr__(self, "_engine", None)
        object.__setattr__(self, "_flat", None)
        object.__setattr__(self, "_flat_grad", None)
        object.__setattr__(self, "_anchor", None)

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

    # ------------------------------------------------------------------ flat arena management
    def __deepcopy__(self, memo):  # EMA wrappers deep-copy the module: copy parameters, not the engine/workspace
        saved = {k: self.__dict__.get(k) for k in ("_engine", "_flat", "_flat_grad", "_anchor")}
        for k in saved:
            object.__setattr__(self, k, None)
        try:
            cls = self.__class__
            new = cls.__new__(cls)
            memo[id(self)] = new
            for k, v in self.__dict__.items():
                new.__dict__[k] = copy.deepcopy(v, memo)
        finally:
            for k, v in saved.items():
                object.__setattr__(self, k, v)
        return new

    def _named(self) -> dict[str, nn.Parameter]:
        return dict(self.named_parameters())

    def _is_flat(self) -> bool:
        if self._flat is None or self._engine is None:
            return False
        lay = self._engine.layout
        base = self._flat.data_ptr()
        for name, p in self.named_parameters():
            if p.data_ptr() != base + 4 * lay.entries[name][0]:
                return False
        return True

    def flatten_parameters(self, device: torch.device | str | None = None) -> None:
        """(re)pack every parameter into the flat f32 arena on ``device`` and point ``.data`` / ``.grad`` at views."""
        named = self._named()
        dev = torch.device(device) if device is not None else next(iter(named.values())).device
        if dev.type != "cuda":
            raise RuntimeError("diffulab_amd.MMDiT runs on an MI355X only: move the module to 'cuda' (no CPU fallback)")
        if self._engine is None or self._engine.dev != dev:
            object.__setattr__(self, "_engine", DiTEngine(self.dims, dev))
        lay = self._engine.layout
        assert set(named) == set(lay.entries), set(named) ^ set(lay.entries)
        # the arena must be ordinary (version-tracked) tensors even when the first forward happens inside
        # torch.inference_mode() (Flow.denoise is decorated with it)
        with torch.inference_mode(False), torch.no_grad():
            flat = torch.zeros(lay.size, device=dev, dtype=torch.float32)
            grad = torch.zeros(lay.size, device=dev, dtype=torch.float32)
            for name, p in named.items():
                v = lay.view(flat, name)
                v.copy_(p.detach().to(device=dev, dtype=torch.float32))
                if p.grad is not None:
                    lay.view(grad, name).copy_(p.grad.to(device=dev, dtype=torch.float32))
                p.data = v
                p.grad = lay.view(grad, name)
            anchor = torch.zeros(1, device=dev, requires_grad=True)
        object.__setattr__(self, "_flat", flat)
        object.__setattr__(self, "_flat_grad", grad)
        object.__setattr__(self, "_anchor", anchor)
        self._engine.bind(flat, grad)

    def _prepare_grads(self) -> None:
        """called at the start of every backward: honour optimizer.zero_grad(set_to_none=True) (torch default) by
        zeroing the arena once and re-attaching the .grad views."""
        lay, grad = self._engine.layout, self._flat_grad
        first = next(iter(self.parameters()))
        if first.grad is None:
            grad.zero_()
        base = grad.data_ptr()
        for name, p in self.named_parameters():
            if p.grad is None or p.grad.data_ptr() != base + 4 * lay.entries[name][0]:
                p.grad = lay.view(grad, name)

    def zero_grad(self, set_to_none: bool = False) -> None:  # one memset instead of one kernel per tensor
        if self._flat_grad is not None and self._is_flat():
            self._flat_grad.zero_()
            self._prepare_grads()
        else:
            super().zero_grad(set_to_none=set_to_none)

    @property
    def engine(self) -> DiTEngine:
        if not self._is_flat():
            self.flatten_parameters()
        return self._engine

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
        need_grad = torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters())
        if need_grad:
            pred = _DiTFn.apply(self, x, t, y_eff, self._anchor)
        else:
            pred = eng.forward(x, t, y_eff, train=False).clone()
        return {"x": pred}
