"""Exponential moving average of the denoiser, one fused launch over the flat parameter arena.

Replaces ``ema_pytorch.EMA`` (third-party, pinned 0.7.7 in the reference's uv.lock, not vendored under /root/reference) as the
reference uses it in base_trainer.py:247-256 (``EMA(denoiser, beta, update_after_step, update_every)``, ``.update()``,
``.ema_model``).  Published update rule restated (ema_pytorch/ema_pytorch.py, ``update`` / ``get_current_decay`` /
``update_moving_average`` with its defaults inv_gamma=1, power=2/3, min_value=0):

    step = self.step ; self.step += 1
    if step % update_every: return
    if step <= update_after_step: ema <- model ; return            (plain copy)
    if not initted: ema <- model ; initted = True
    epoch = max(step - update_after_step - 1, 0)
    decay = 0 if epoch <= 0 else clamp(1 - (1 + epoch / inv_gamma) ** -power, min_value, beta)
    ema <- ema + (1 - decay) * (model - ema)                        (Tensor.lerp_)

``ema_model`` is a deep copy of the denoiser (its own flat arena), so the lerp over ALL parameters is one ``dl_ema_update``
launch instead of one ATen kernel per tensor.
"""

from __future__ import annotations

import copy

import torch

from .. import ops
from ..engine import bump_param_epoch


class EMA(torch.nn.Module):
    def __init__(self, model: torch.nn.Module, beta: float = 0.9999, update_after_step: int = 100, update_every: int = 10,
                 inv_gamma: float = 1.0, power: float = 2 / 3, min_value: float = 0.0) -> None:
        super().__init__()
        self.beta, self.update_after_step, self.update_every = beta, update_after_step, update_every
        self.inv_gamma, self.power, self.min_value = inv_gamma, power, min_value
        object.__setattr__(self, "online_model", model)  # not registered: the online model is not part of the EMA state
        self.ema_model = copy.deepcopy(model)
        self.ema_model.requires_grad_(False)
        self.step = 0
        self.initted = False

    @property
    def model(self) -> torch.nn.Module:
        return self.online_model

    def get_current_decay(self) -> float:
        epoch = max(self.step - self.update_after_step - 1, 0)
        if epoch <= 0:
            return 0.0
        value = 1 - (1 + epoch / self.inv_gamma) ** -self.power
        return min(max(value, self.min_value), self.beta)

    def _arenas(self):
        src, dst = self.online_model, self.ema_model
        if hasattr(src, "engine") and hasattr(dst, "engine"):
            src.engine, dst.engine  # noqa: B018 -- (re)flatten both if needed
            if src._flat is not None and dst._flat is not None and src._flat.numel() == dst._flat.numel():
                return src._flat, dst._flat
        return None

    @torch.no_grad()
    def _lerp(self, weight: float) -> None:
        """ema <- ema + weight * (model - ema); weight = 1 is the plain copy"""
        flat = self._arenas()
        if flat is not None:
            ops.ema_update(flat[1], flat[0], 1.0 - weight)
        else:
            for pe, pm in zip(self.ema_model.parameters(), self.online_model.parameters()):
                ops.ema_update(pe.data, pm.data, 1.0 - weight)
        bump_param_epoch()

    def copy_params_from_model_to_ema(self) -> None:
        self._lerp(1.0)

    def update(self) -> None:
        step = self.step
        self.step += 1
        if step % self.update_every != 0:
            return
        if step <= self.update_after_step:
            self.copy_params_from_model_to_ema()
            return
        if not self.initted:
            self.copy_params_from_model_to_ema()
            self.initted = True
        self._lerp(1.0 - self.get_current_decay())  # reads the already incremented step counter, like ema_pytorch

    def forward(self, *args, **kwargs):
        return self.ema_model(*args, **kwargs)
