"""DDIM step (reference samplers/gaussian_diffusion/ddim.py:9-103) on the fused `dl_ddim_step` kernel."""

from __future__ import annotations

import torch
from torch import Tensor

from .... import ops
from ...utils import f32_table
from ..common import StepResult
from .ddpm import _MEAN, DDPM


class DDIM(DDPM):
    name = "ddim"

    def _ddim_tables(self, device: torch.device) -> tuple[Tensor, Tensor]:
        key = ("ddim", device)
        if key not in self._dev_tables:
            self._dev_tables[key] = f32_table(torch.stack([self.sqrt_alphas_bar, self.alphas_bar, self.alphas_bar_prev]), device)
            self._dev_tables[("ddim_c", device)] = f32_table(
                torch.stack([self.posterior_mean_coef1, self.posterior_mean_coef2]), device)
        return self._dev_tables[key], self._dev_tables[("ddim_c", device)]

    def step(self, model_prediction: Tensor, timesteps: Tensor, xt: Tensor, clamp_x: bool = False, eta: float = 0.0,
             prediction_uncond: Tensor | None = None, guidance_scale: float = 0.0) -> StepResult:
        self._variance_tables()  # same var_type validation as DDPM._get_p_mean_var (ddim.py:89)
        tab, coefs = self._ddim_tables(xt.device)
        noise = torch.randn_like(xt)  # ddim.py:63
        xp, x0, mean, std, lp = ops.ddim_step(
            model_prediction.float().contiguous(), None if prediction_uncond is None else prediction_uncond.float().contiguous(),
            guidance_scale, xt.float().contiguous(), noise, timesteps.to(torch.int32).contiguous(), tab, coefs,
            _MEAN.index(self.mean_type), clamp_x, eta)
        out = StepResult(x_prev=xp, estimated_x0=x0, x_prev_mean=mean)
        if eta > 0:
            out["x_prev_std"] = std
            out["logprob"] = lp
        return out
