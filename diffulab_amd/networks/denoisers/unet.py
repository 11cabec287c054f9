"""``UNetModel`` denoiser, MI355X build -- drop-in for ``diffulab.networks.denoisers.unet.UNetModel``.

Same constructor kwargs (unet.py:531-551 of the reference), same ``forward`` kwargs (unet.py:749-757), same ``state_dict``
key names / shapes (``input_blocks.{i}.{j}...``, ``middle_block.{j}...``, ``output_blocks.{i}.{j}...``, ``out.{0,2}``,
``time_embed.{0,2}``, ``label_embed.embedding``) and the same initialisation (torch defaults, ``zero_module`` on the second
conv of every ResBlock and on the output conv: unet.py:172,744), so Hydra ``_target_`` configs (configs/model/unet.yaml,
configs/train_mnist_ddpm.yaml) and ``denoiser.pt`` checkpoints carry over.  The ``nn.Module`` tree below only OWNS the
parameters (views into one flat HBM arena); the arithmetic is the hand-written HIP path driven by
``diffulab_amd.unet_engine.UNetEngine``.  No PyTorch/CPU fallback exists.

Covered: FiLM (``use_scale_shift_norm=True``, what ``configs/model/unet.yaml`` builds) and additive conditioning, ResBlock
resampling (``resblock_updown=True``) and the plain ``Downsample`` / ``Upsample`` modules with or without their 3x3 conv
(``conv_resample``), labels or unconditional -- the constructor defaults included.  The cross-attention Transformer blocks of a
context embedder and dropout > 0 raise ``NotImplementedError``.
"""

from __future__ import annotations

from typing import Any

import torch
import torch.nn as nn
from torch import Tensor

from ...unet_engine import UNetDims, UNetEngine, build_plan
from .common import FlatArenaDenoiser, ModelOutput
from ...diffuse.utils import to_device
from .mmdit import _LabelEmbed


class _ResBlock(nn.Module):  # parameter holder for unet.py:80-237
    def __init__(self, cin: int, cout: int, emb: int, film: bool) -> None:
        super().__init__()
        self.in_layers = nn.Sequential(nn.GroupNorm(32, cin), nn.SiLU(), nn.Conv2d(cin, cout, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb, 2 * cout if film else cout))
        conv = nn.Conv2d(cout, cout, 3, padding=1)
        for p in conv.parameters():  # zero_module (unet.py:172)
            p.detach().zero_()
        self.out_layers = nn.Sequential(nn.GroupNorm(32, cout), nn.SiLU(), nn.Dropout(0.0), conv)
        self.skip_connection = nn.Identity() if cin == cout else nn.Conv2d(cin, cout, 1)


class _Downsample(nn.Module):  # parameter holder for nn.py:59-88
    def __init__(self, c: int, use_conv: bool) -> None:
        super().__init__()
        self.op = nn.Conv2d(c, c, 3, stride=2, padding=1) if use_conv else nn.AvgPool2d(kernel_size=2, stride=2)


class _Upsample(nn.Module):  # parameter holder for nn.py:28-56
    def __init__(self, c: int, use_conv: bool) -> None:
        super().__init__()
        if use_conv:
            self.conv = nn.Conv2d(c, c, 3, padding=1)


class _AttentionBlock(nn.Module):  # parameter holder for unet.py:240-322
    def __init__(self, c: int) -> None:
        super().__init__()
        self.norm_x = nn.GroupNorm(32, c)
        self.norm_context = nn.GroupNorm(32, c)
        self.to_q = nn.Conv1d(c, c, 1)
        self.to_kv = nn.Conv1d(c, 2 * c, 1)
        self.to_out = nn.Sequential(nn.Conv1d(c, c, 1), nn.Dropout(0.0))


class UNetModel(FlatArenaDenoiser):
    cfg_pair_capable = True  # `p` only reaches the label drop: guided sampler steps batch their two forwards (forward_cfg_pair)

    def __init__(
        self,
        image_size: list[int],
        in_channels: int,
        model_channels: int,
        out_channels: int,
        num_res_blocks: int,
        attention_resolutions: list[int],
        dropout: float = 0,
        channel_mult: str = "1, 2, 4, 8",
        conv_resample: bool = True,
        use_checkpoint: bool = False,
        num_heads: int = 1,
        use_scale_shift_norm: bool = False,
        resblock_updown: bool = False,
        n_classes: int | None = None,
        classifier_free: bool = False,
        context_embedder: Any | None = None,
        transformer_depth: int = 1,
    ) -> None:
        super().__init__()
        assert not (n_classes is not None and context_embedder is not None), (
            "n_classes and context_embedder cannot both be specified")
        if context_embedder is not None:
            raise NotImplementedError("diffulab_amd.UNetModel: cross-attention context blocks (unet.py:358-465) are not built")
        if dropout:
            raise NotImplementedError("diffulab_amd.UNetModel: dropout > 0 is not built (every reference config uses 0)")
        self.image_size = list(image_size)
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = list(attention_resolutions)
        self.dropout = dropout
        self.channel_mult: list[int] = [int(v) for v in str(channel_mult).split(",")]
        self.conv_resample = conv_resample
        self.use_checkpoint = use_checkpoint  # activations stay resident in HBM; nothing to checkpoint
        self.num_heads = num_heads
        self.context_embedder = None
        self.classifier_free = classifier_free
        self.n_classes = n_classes
        self.time_embed_dim = model_channels * 4
        self.dims = UNetDims(image_size=tuple(self.image_size), in_channels=in_channels, model_channels=model_channels,
                             out_channels=out_channels, num_res_blocks=num_res_blocks,
                             attention_resolutions=tuple(self.attention_resolutions), channel_mult=tuple(self.channel_mult),
                             num_heads=num_heads, use_scale_shift_norm=use_scale_shift_norm, resblock_updown=resblock_updown,
                             conv_resample=conv_resample, n_classes=n_classes, classifier_free=classifier_free)
        self.dims.validate()

        te = self.time_embed_dim
        self.time_embed = nn.Sequential(nn.Linear(model_channels, te), nn.SiLU(), nn.Linear(te, te))
        self.label_embed = _LabelEmbed(n_classes, te, classifier_free) if n_classes is not None else None
        plan = build_plan(self.dims)

        def group(blocks) -> nn.Sequential:
            mods: list[nn.Module] = []
            for b in blocks:
                if b.kind == "conv":
                    mods.append(nn.Conv2d(b.cin, b.cout, 3, padding=1))
                elif b.kind == "res":
                    mods.append(_ResBlock(b.cin, b.cout, te, use_scale_shift_norm))
                elif b.kind == "down":
                    mods.append(_Downsample(b.cin, conv_resample))
                elif b.kind == "up":
                    mods.append(_Upsample(b.cin, conv_resample))
                else:
                    mods.append(_AttentionBlock(b.cin))
            return nn.Sequential(*mods)

        self.input_blocks = nn.ModuleList([group(g) for g in plan.input_blocks])
        self.middle_block = group(plan.middle)
        self.output_blocks = nn.ModuleList([group(g) for g in plan.output_blocks])
        conv = nn.Conv2d(int(self.channel_mult[0] * model_channels), out_channels, 3, padding=1)
        for p in conv.parameters():  # zero_module (unet.py:744)
            p.detach().zero_()
        self.out = nn.Sequential(nn.GroupNorm(32, plan.final_ch), nn.SiLU(), conv)

    precisions = ("bf16", "fp32")  # the fp32-class regime: unet_engine_f32.py (round 4)

    def _make_engine(self, device: torch.device) -> UNetEngine:
        if self.precision == "fp32":
            from ...unet_engine_f32 import UNetEngineF32

            return UNetEngineF32(self.dims, device)
        return UNetEngine(self.dims, device)

    # ------------------------------------------------------------------ forward (unet.py:749-853)
    def forward(
        self,
        x: Tensor,
        timesteps: Tensor,
        y: Tensor | None = None,
        context: Any | None = None,
        p: float = 0.0,
        x_context: Tensor | None = None,
    ) -> ModelOutput:
        assert (y is not None) == (self.n_classes is not None), "must specify y if and only if the model is class-conditional"
        assert context is None, "must specify context if and only if the model is context-conditional"
        if p > 0:
            assert self.classifier_free, (
                "probability of dropping for classifier free guidance is only available if model is set up to be classifier free")
            assert self.n_classes, (
                "probability of dropping for classifier free guidance is only available if a number of classes is set")
        if x_context is not None:
            x = torch.cat([x, x_context], dim=1)
        assert list(x.shape[2:]) == self.image_size, f"Input shape {x.shape[2:]} does not match model image size {self.image_size}"
        dev = self.engine.dev
        x = x.to(device=dev, dtype=torch.float32).contiguous()
        t = to_device(timesteps, dev, torch.float32)
        y_eff = None
        if self.label_embed is not None:
            y_eff = self._effective_labels(y.to(device=dev, dtype=torch.int64), p).contiguous()  # (drop_labels nn.py:149)
        return {"x": self._run(x, t, y_eff)}
