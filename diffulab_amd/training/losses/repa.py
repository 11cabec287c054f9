"""Representation-alignment (REPA) loss on the HIP path -- drop-in for ``diffulab.training.losses.repa.RepaLoss`` with
precomputed target features (``load_dino=False``; the DINOv2 / DINOv3 encoders need pretrained weights and are outside the hot
path).  ``use_resampler=True`` inserts the HIP-path Perceiver resampler between the MLP and the cosine.

Same constructor kwargs, ``set_model`` / forward-hook mechanics and arithmetic as training/losses/repa.py:96-198:
features of ``denoiser.layers[alignment_layer - 1]`` -> 3-layer SiLU MLP [-> resampler] -> ``coeff * (1 - mean(cosine_similarity(., dst_features)))``.
The MLP runs on the bf16 MFMA GEMMs (bias + SiLU fused in the epilogue, pre-activations kept for the backward), the cosine
rows and their gradient are one HIP kernel each; the feature gradient flows back into the DiT engine's residual stream.
"""

from __future__ import annotations

from typing import Any

import torch
import torch.nn as nn
from torch import Tensor
from torch.utils.hooks import RemovableHandle

from ... import ops
from ...networks.repa import PerceiverResampler
from .common import LossFunction


def _rup(v: int, m: int) -> int:
    return (v + m - 1) // m * m


class _ProjMLP(torch.autograd.Function):
    """proj MLP (repa.py:96-102) as one autograd node: explicit forward / backward launch sequences over the C ABI.
    Returns the projected features bf16 [B, N, E]."""

    @staticmethod
    def forward(ctx, feat: Tensor, w1, b1, w2, b2, w3, b3) -> Tensor:
        B, N, D = feat.shape
        M = B * N
        dev = feat.device
        x = feat.reshape(M, D)
        if x.dtype != torch.bfloat16 or not x.is_contiguous():
            x = x.to(torch.bfloat16).contiguous()
        bf = torch.bfloat16
        Hd, E = w1.shape[0], w3.shape[0]
        if D % 64 or Hd % 64 or E % 8 or M % 64:
            raise NotImplementedError("RepaLoss HIP head: denoiser_dimension / hidden_dim % 64, embedding_dim % 8, tokens % 64")
        shadows = []
        for w in (w1, w2, w3):  # bf16 shadows W [out, in] (forward) and W^T [in, rup64(out)] (data gradient)
            f = torch.empty(w.shape[0], w.shape[1], device=dev, dtype=bf)
            t = torch.zeros(w.shape[1], _rup(w.shape[0], 64), device=dev, dtype=bf)
            ops.cast_weight(w.detach(), f, t)
            shadows.append((f, t))
        pre1, h1 = torch.empty(M, Hd, device=dev, dtype=bf), torch.empty(M, Hd, device=dev, dtype=bf)
        pre2, h2 = torch.empty(M, Hd, device=dev, dtype=bf), torch.empty(M, Hd, device=dev, dtype=bf)
        proj = torch.empty(M, E, device=dev, dtype=bf)
        ops.gemm_nt(x, shadows[0][0], h1, bias=b1.detach(), act=ops.ACT_SILU, pre_out=pre1)
        ops.gemm_nt(h1, shadows[1][0], h2, bias=b2.detach(), act=ops.ACT_SILU, pre_out=pre2)
        ops.gemm_nt(h2, shadows[2][0], proj, bias=b3.detach(), M=M, N=E, K=Hd)
        ctx.save_for_backward(x, pre1, h1, pre2, h2)
        ctx.shadows, ctx.dims, ctx.feat_shape = shadows, (M, D, Hd, E), feat.shape
        return proj.view(B, N, E)

    @staticmethod
    def backward(ctx, dout: Tensor):
        x, pre1, h1, pre2, h2 = ctx.saved_tensors
        (f1, t1), (f2, t2), (f3, t3) = ctx.shadows
        M, D, Hd, E = ctx.dims
        dev, bf = x.device, torch.bfloat16
        E64 = _rup(E, 64)
        dproj = dout.reshape(M, E)
        if E64 != E or dproj.dtype != bf or not dproj.is_contiguous():  # K-padded for the dgrad GEMM
            pad = torch.zeros(M, E64, device=dev, dtype=bf)
            pad[:, :E] = dproj
            dproj = pad
        E8 = _rup(E, 8)
        scr = ops.shared_scratch(dev, 4 * max(Hd * Hd, Hd * D, E8 * Hd))  # partial images of the weight gradients (dl_gemm_tn_det)
        dw3, db3 = torch.zeros(E8, Hd, device=dev), torch.zeros(E, device=dev)
        ops.gemm_tn(dproj, h2, dw3, M=E8, N=Hd, scratch=scr)
        ops.colsum(dproj, db3, M, E)
        dh2 = torch.empty(M, Hd, device=dev)
        ops.gemm_nt(dproj, t3, dh2, M=M, N=Hd, K=E64)
        dpre2 = torch.empty(M, Hd, device=dev, dtype=bf)
        ops.silu_bwd(dh2, pre2, dpre2)
        dw2, db2 = torch.zeros(Hd, Hd, device=dev), torch.zeros(Hd, device=dev)
        ops.gemm_tn(dpre2, h1, dw2, scratch=scr)
        ops.colsum(dpre2, db2, M, Hd)
        dh1 = dh2  # reuse
        ops.gemm_nt(dpre2, t2, dh1)
        dpre1 = torch.empty(M, Hd, device=dev, dtype=bf)
        ops.silu_bwd(dh1, pre1, dpre1)
        dw1, db1 = torch.zeros(Hd, D, device=dev), torch.zeros(Hd, device=dev)
        ops.gemm_tn(dpre1, x, dw1, scratch=scr)
        ops.colsum(dpre1, db1, M, Hd)
        dfeat = torch.empty(M, D, device=dev, dtype=bf)
        ops.gemm_nt(dpre1, t1, dfeat)
        return dfeat.view(ctx.feat_shape), dw1, db1, dw2, db2, dw3[:E], db3


class _CosineLoss(torch.autograd.Function):
    """coeff * (1 - mean(cosine_similarity(p, dst, dim=-1))) (repa.py:184-186): one HIP kernel per direction"""

    @staticmethod
    def forward(ctx, p: Tensor, dst: Tensor, coeff: float) -> Tensor:
        E = p.shape[-1]
        M = p.numel() // E
        dev = p.device
        p2 = p.reshape(M, E)
        if p2.dtype != torch.bfloat16 or not p2.is_contiguous():
            p2 = p2.to(torch.bfloat16).contiguous()
        d = dst.reshape(M, E).to(device=dev, dtype=torch.float32).contiguous()
        cosv, pn2, dn2 = (torch.empty(M, device=dev) for _ in range(3))
        ops.cosine_rows_fwd(p2, d, cosv, pn2, dn2)
        s = torch.zeros(1, device=dev)
        ops.colsum(cosv.view(M, 1), s, M, 1)
        ctx.save_for_backward(p2, d, cosv, pn2, dn2)
        ctx.coeff, ctx.p_shape = coeff, p.shape
        return coeff * (1.0 - s[0] / M)

    @staticmethod
    def backward(ctx, gout: Tensor):
        p2, d, cosv, pn2, dn2 = ctx.saved_tensors
        M, E = p2.shape
        g = gout.detach().reshape(1).float().contiguous()
        dp = torch.empty_like(p2)
        ops.cosine_rows_bwd(p2, d, cosv, pn2, dn2, -ctx.coeff / M, g, dp)
        return dp.view(ctx.p_shape), None, None


class RepaLoss(LossFunction):
    name: str = "RepaLoss"
    encoder_registry: dict[str, Any] = {}  # DINOv2 / DINOv3 need pretrained weights: precomputed dst_features only

    def __init__(
        self,
        repa_encoder: str = "dinov2",
        encoder_args: dict[str, Any] = {},
        alignment_layer: int = 8,
        denoiser_dimension: int = 256,
        hidden_dim: int = 1024,
        load_dino: bool = True,
        embedding_dim: int = 768,
        use_resampler: bool = False,
        resampler_params: dict[str, Any] | None = None,
        coeff: float = 1.0,
    ) -> None:
        super().__init__()
        if load_dino:
            raise NotImplementedError("diffulab_amd.RepaLoss: pass load_dino=False and precomputed dst_features (the "
                                      f"{repa_encoder} encoder needs pretrained weights that are not available offline)")
        self.repa_encoder = None
        self.proj = nn.Sequential(nn.Linear(denoiser_dimension, hidden_dim), nn.SiLU(), nn.Linear(hidden_dim, hidden_dim),
                                  nn.SiLU(), nn.Linear(hidden_dim, embedding_dim))
        self.resampler: PerceiverResampler | None = None
        if use_resampler:
            assert resampler_params is not None, "Resampler parameters must be provided when using the perceiver resampler."
            self.resampler = PerceiverResampler(**resampler_params)
        self.alignment_layer = alignment_layer
        self._handles: dict[int, RemovableHandle] = {}
        self._captured_features: dict[int, Tensor] = {}
        self._active_model_id: int | None = None
        self._hook_layer_idx = self.alignment_layer - 1
        self.coeff = coeff

    def _make_hook(self, model_id: int):
        def _hook(_mod: nn.Module, _inp: tuple[Any, ...], out: Tensor) -> None:
            self._captured_features[model_id] = out

        return _hook

    def _attach_hook(self, model: nn.Module) -> None:
        model_id = id(model)
        if model_id in self._handles:
            return
        layer = model.layers[self._hook_layer_idx]
        self._handles[model_id] = layer.register_forward_hook(self._make_hook(model_id))

    def set_model(self, model: nn.Module) -> None:
        """attach the forward hook to ``model.layers[alignment_layer - 1]`` (once per model); a forward pass of ``model``
        must follow before the loss is evaluated"""
        self._attach_hook(model)
        self._active_model_id = id(model)

    def _unregister_all(self) -> None:
        for handle in self._handles.values():
            handle.remove()
        self._handles.clear()
        self._captured_features.clear()
        self._active_model_id = None

    def forward(self, x0: Tensor | None = None, dst_features: Tensor | None = None) -> Tensor:
        if self._active_model_id is None or self._active_model_id not in self._captured_features:
            raise RuntimeError(
                "REPA: no captured features for the active model. Did you call set_model(...) and run a forward pass?")
        assert x0 is not None or dst_features is not None, "Either x0 or dst_features must be provided."
        if dst_features is None:
            raise NotImplementedError("diffulab_amd.RepaLoss: dst_features must be precomputed (no encoder is loaded)")
        src = self._captured_features[self._active_model_id]
        if isinstance(src, tuple):
            src = src[0]
        p = self.proj
        projected = _ProjMLP.apply(src, p[0].weight, p[0].bias, p[2].weight, p[2].bias, p[4].weight, p[4].bias)
        if self.resampler is not None:
            projected = self.resampler(projected)
        return _CosineLoss.apply(projected, dst_features, float(self.coeff))
