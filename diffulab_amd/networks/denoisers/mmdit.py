"""``MMDiT`` denoiser, MI355X build -- drop-in for ``diffulab.networks.denoisers.mmdit.MMDiT``: the class-conditional
``simple_dit`` form and the joint text-image form (``simple_dit=False`` with a context embedder that returns one embedding and an
attention mask, e.g. ``PrecomputedEmbedder``; ``MMDiTBlock`` layers, ``n_single_stream_blocks == 0``).

Same constructor kwargs (mmdit.py:604-625 of the reference), same ``forward`` kwargs (mmdit.py:903-912), same
``state_dict`` key names / shapes and the same initialisation scheme (mmdit.py:735-745), so Hydra ``_target_``
configs and ``denoiser.pt`` checkpoints carry over.  The ``nn.Module`` tree below only OWNS the parameters
(as views into one flat HBM arena); the arithmetic is the hand-written HIP path driven by
``diffulab_amd.engine.DiTEngine`` / ``diffulab_amd.mmdit_engine.JointEngine``.  There is no PyTorch/CPU fallback: calling ``forward`` without a GPU or
without ``libdiffulab_hip.so`` raises.
"""

from __future__ import annotations

import logging
from typing import Any

import torch
import torch.nn as nn
from torch import Tensor

from ...engine import DiTDims, DiTEngine
from ...mmdit_engine import JointDims, JointEngine
from ...sprint_joint_engine import JointStackDims
from .common import FlatArenaDenoiser, ModelOutput
from ...diffuse.utils import to_device


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


class _JointAttention(nn.Module):  # parameter container of MMDiTAttention (mmdit.py:150-170)
    def __init__(self, dim: int) -> None:
        super().__init__()
        self.qkv_input = nn.Linear(dim, 3 * dim, bias=False)
        self.qkv_context = nn.Linear(dim, 3 * dim, bias=False)
        self.qk_norm_input = _QKNorm(dim)
        self.qk_norm_context = _QKNorm(dim)
        self.input_proj_out = nn.Linear(dim, dim, bias=False)
        self.context_proj_out = nn.Linear(dim, dim, bias=False)


class MMDiTBlock(nn.Module):
    """parameter container of one joint text-image block (reference mmdit.py:346-372); never called on its own"""

    def __init__(self, inner_dim: int, embedding_dim: int, mlp_ratio: int) -> None:
        super().__init__()
        mlp = lambda: nn.Sequential(nn.Linear(inner_dim, mlp_ratio * inner_dim * 2, bias=False), nn.Identity(),  # noqa: E731
                                    nn.Linear(mlp_ratio * inner_dim, inner_dim, bias=False))
        self.modulation_context = _Modulation(embedding_dim, inner_dim)
        self.modulation_input = _Modulation(embedding_dim, inner_dim)
        self.context_norm_1 = nn.LayerNorm(inner_dim)
        self.input_norm_1 = nn.LayerNorm(inner_dim)
        self.attention = _JointAttention(inner_dim)
        self.context_norm_2 = nn.LayerNorm(inner_dim)
        self.input_norm_2 = nn.LayerNorm(inner_dim)
        self.mlp_context = mlp()
        self.mlp_input = mlp()

    def forward(self, *a: Any, **k: Any) -> Tensor:
        raise RuntimeError("MMDiTBlock parameters are consumed by the fused HIP engine; call the MMDiT module instead")


class MMDiTSingleStreamBlock(nn.Module):
    """parameter container of a single-stream block (reference mmdit.py:442-470); never called on its own"""

    def __init__(self, inner_dim: int, embedding_dim: int, mlp_ratio: int) -> None:
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(inner_dim, mlp_ratio * inner_dim * 2, bias=False), nn.Identity(),
                                 nn.Linear(mlp_ratio * inner_dim, inner_dim, bias=False))
        self.attention = _Attention(inner_dim)
        self.modulation = nn.Sequential(nn.SiLU(), nn.Linear(embedding_dim, 3 * inner_dim))
        self.norm = nn.LayerNorm(inner_dim)

    def forward(self, *a: Any, **k: Any) -> Tensor:
        raise RuntimeError("MMDiTSingleStreamBlock parameters are consumed by the fused HIP engine; call the owning denoiser module")


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
    cfg_pair_capable = True  # `p` only reaches the label drop: guided sampler steps batch their two forwards (forward_cfg_pair)

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
        assert not (n_classes is not None and context_embedder is not None), "n_classes and context_embedder cannot both be specified"
        if simple_dit and context_embedder is not None:
            raise NotImplementedError("diffulab_amd.MMDiT: simple_dit=True takes class labels, not a context embedder")
        if n_single_stream_blocks > 0 and simple_dit:
            logging.warning("n_single_stream_blocks is ignored when simple_dit=True. All blocks are single-stream DiT blocks.")
        if n_single_stream_blocks > depth:
            raise ValueError("n_single_stream_blocks cannot exceed depth")
        self.n_single_stream_blocks = 0 if simple_dit else n_single_stream_blocks
        self.simple_dit = simple_dit
        self.patch_size = patch_size
        self.input_channels = input_channels
        self.output_channels = output_channels or input_channels
        self.context_embedder = context_embedder
        self.frequency_embedding = frequency_embedding
        self.rope_base = rope_base
        self.n_classes = n_classes
        self.classifier_free = classifier_free
        self.use_checkpoint = use_checkpoint  # activations are kept resident in HBM; nothing to checkpoint
        heads_dim = inner_dim // num_heads
        if not simple_dit:
            self._init_joint(inner_dim, embedding_dim, num_heads, mlp_ratio, depth, rope_axes_dim, partial_rotary_factor, heads_dim)
            return
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

    def _init_joint(self, inner_dim: int, embedding_dim: int, num_heads: int, mlp_ratio: int, depth: int,
                    rope_axes_dim: list[int] | None, partial_rotary_factor: float, heads_dim: int) -> None:
        """mmdit.py:641-664,678-734 with MMDiTBlock layers"""
        ce = self.context_embedder
        assert ce is not None, "for MMDiT context embedder must be provided"
        assert isinstance(ce.output_size, tuple) and all(isinstance(i, int) for i in ce.output_size), (
            "context_embedder.output_size must be a tuple of integers")
        if ce.n_output != 1:
            raise NotImplementedError("diffulab_amd.MMDiT: context embedders with a pooled embedding (n_output == 2, "
                                      "mmdit.py:646-654) are not built")
        if any(True for _ in ce.parameters()):
            raise NotImplementedError("diffulab_amd.MMDiT: the context embedder must be parameter-free (precomputed embeddings)")
        self.pooled_embedding = False
        self.mlp_pooled_context = None
        self.context_embed = nn.Linear(ce.output_size[0], inner_dim, bias=False)
        if rope_axes_dim is None:
            rope_axes_dim = [int((partial_rotary_factor * heads_dim) // 3)] * 3
        self.rope_axes_dim = list(rope_axes_dim)
        ns = self.n_single_stream_blocks
        common = dict(input_channels=self.input_channels, output_channels=self.output_channels, inner_dim=inner_dim,
                      embedding_dim=embedding_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, patch_size=self.patch_size, depth=depth,
                      rope_base=float(self.rope_base), frequency_embedding=self.frequency_embedding, n_classes=None,
                      classifier_free=self.classifier_free, rope_axes_dim=self.rope_axes_dim, context_dim=ce.output_size[0])
        self.dims = JointStackDims(n_single_stream_blocks=ns, **common) if ns else JointDims(**common)
        self.dims.validate()
        self.last_layer = _LastLayer(embedding_dim, inner_dim, self.patch_size, self.output_channels)
        self.time_embed = nn.Sequential(nn.Linear(self.frequency_embedding, embedding_dim), nn.SiLU(),
                                        nn.Linear(embedding_dim, embedding_dim))
        self.conv_proj = nn.Conv2d(self.input_channels, inner_dim, kernel_size=self.patch_size, stride=self.patch_size, bias=False)
        self.layers = nn.ModuleList([MMDiTBlock(inner_dim, embedding_dim, mlp_ratio) for _ in range(depth - ns)]
                                    + [MMDiTSingleStreamBlock(inner_dim, embedding_dim, mlp_ratio) for _ in range(ns)])
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

    @property
    def precisions(self) -> tuple[str, ...]:  # the fp32-class regime is built for the class-conditional DiT (engine_f32.py)
        return ("bf16", "fp32") if self.simple_dit else ("bf16",)

    def _make_engine(self, device: torch.device) -> DiTEngine:
        if self.simple_dit:
            if self.precision == "fp32":
                from ...engine_f32 import DiTEngineF32

                return DiTEngineF32(self.dims, device)
            return DiTEngine(self.dims, device)
        if self.n_single_stream_blocks:
            from ...sprint_joint_engine import JointStackEngine

            return JointStackEngine(self.dims, device)
        return JointEngine(self.dims, device)

    # the context tensors are per-call inputs of the joint launch sequence: a captured graph reads static copies of them
    def _graph_inputs(self, eng) -> tuple:
        return () if self.simple_dit else tuple(eng.context)

    def _graph_set_inputs(self, eng, tensors: tuple) -> None:
        if not self.simple_dit:
            eng.context = tuple(tensors)

    def _forward_joint(self, x: Tensor, timesteps: Tensor, initial_context: Any, p: float) -> ModelOutput:
        """mmdit.py:789-851: the embedder (context drop for classifier-free guidance) stays host-side torch"""
        assert self.context_embedder is not None, "for MMDiT context embedder must be provided"
        out = self.context_embedder(initial_context, p)
        eng = self.engine
        dev = eng.dev
        emb = out["embeddings"].to(device=dev)
        keep = out.get("attn_mask", None)
        eng.context = (emb, keep.to(device=dev) if keep is not None else None)
        x = x.to(device=dev, dtype=torch.float32).contiguous()
        t = to_device(timesteps, dev, torch.float32)
        taps = tuple(i for i, layer in enumerate(self.layers) if layer._forward_hooks)
        if not taps:
            return {"x": self._run(x, t, None)}
        pred, *feats = self._run(x, t, None, taps)
        for i, f in zip(taps, feats):
            for hook in list(self.layers[i]._forward_hooks.values()):
                hook(self.layers[i], (), f)
        return {"x": pred}

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
        if p > 0:
            assert self.classifier_free, (
                "probability of dropping for classifier free guidance is only available if model is set up to be classifier free")
        if x_context is not None:
            x = torch.cat([x, x_context], dim=1)
        if not self.simple_dit:
            return self._forward_joint(x, timesteps, initial_context, p)
        if initial_context is not None:
            raise NotImplementedError("simple_dit has no context stream")
        if p > 0:
            assert self.n_classes, (
                "probability of dropping for classifier free guidance is only available if a number of classes is set")
        eng = self.engine
        dev = eng.dev
        x = x.to(device=dev, dtype=torch.float32).contiguous()
        t = to_device(timesteps, dev, torch.float32)  # nn.py:110: timesteps[:, None].float()
        y_eff = None
        if self.label_embed is not None:
            assert y is not None, "class-conditional DiT needs labels `y`"
            y_eff = self._effective_labels(y.to(device=dev, dtype=torch.int64), p).contiguous()  # (drop_labels nn.py:149)
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
