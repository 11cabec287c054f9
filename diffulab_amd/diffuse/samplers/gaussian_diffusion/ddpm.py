"""DDPM ancestral step (reference samplers/gaussian_diffusion/ddpm.py:25-363) on the fused `dl_ddpm_step` kernel.

Host side keeps the fp64 schedule tables exactly as the reference builds them (ddpm.py:49-85); the kernel gets
their fp32 casts (what `extract_into_tensor(...).float()` would produce per call)."""

from __future__ import annotations

import torch
from torch import Tensor

from .... import ops
from ...utils import f32_table
from ..common import Sampler, StepResult

_MEAN = ("epsilon", "xstart", "xprev")
_VAR = ("learned", "fixed_small", "fixed_large", "learned_range")


class GaussianSampler(Sampler):
    name: str


class DDPM(GaussianSampler):
    name = "ddpm"

    def __init__(self, mean_type: str = "epsilon", var_type: str = "fixed_small") -> None:
        super().__init__()
        if mean_type not in _MEAN:
            raise ValueError(f"mean_type must be one of {list(_MEAN)}")
        if var_type not in _VAR:
            raise ValueError(f"variance_type must be one of {list(_VAR)}")
        self.mean_type, self.var_type = mean_type, var_type
        self._dev_tables: dict[tuple, Tensor] = {}

    def set_steps(self, betas: Tensor) -> None:
        one = torch.ones_like(betas)
        self.betas = betas
        self.alphas = one - betas
        self.alphas_bar = self.alphas.cumprod(dim=0)
        self.alphas_bar_prev = torch.cat([torch.tensor([1.0], dtype=torch.float64), self.alphas_bar[:-1]])
        self.alphas_bar_next = torch.cat([self.alphas_bar[1:], torch.tensor([0.0], dtype=torch.float64)])
        self.sqrt_alphas_bar = self.alphas_bar.sqrt()
        self.posterior_variance = betas * (one - self.alphas_bar_prev) / (one - self.alphas_bar)
        self.posterior_log_variance_clipped = torch.log(torch.cat([self.posterior_variance[1:2], self.posterior_variance[1:]]))
        self.posterior_mean_coef1 = betas * self.alphas_bar_prev.sqrt() / (one - self.alphas_bar)
        self.posterior_mean_coef2 = (one - self.alphas_bar_prev) * self.alphas.sqrt() / (one - self.alphas_bar)
        self._dev_tables.clear()

    def _variance_tables(self) -> tuple[Tensor, Tensor]:
        if self.var_type == "fixed_small":
            return self.posterior_variance, self.posterior_log_variance_clipped
        if self.var_type == "fixed_large":
            v = torch.cat([self.posterior_variance[1:2], self.betas[1:]])
            return v, torch.log(v)
        raise NotImplementedError(
            f"var_type={self.var_type!r}: learned variances need a 2x-channel denoiser head; not on the HIP path yet")

    def _tables(self, device: torch.device) -> Tensor:
        key = ("ddpm", device, self.var_type)
        if key not in self._dev_tables:
            var, lv = self._variance_tables()
            t = torch.stack([self.sqrt_alphas_bar, self.alphas_bar, self.posterior_mean_coef1, self.posterior_mean_coef2, var, lv])
            self._dev_tables[key] = f32_table(t, device)
        return self._dev_tables[key]

    def step(self, model_prediction: Tensor, timesteps: Tensor, xt: Tensor, clamp_x: bool = False,
             prediction_uncond: Tensor | None = None, guidance_scale: float = 0.0) -> StepResult:
        noise = torch.randn_like(xt)  # ddpm.py:302 (drawn for every t, masked at t == 0)
        xp, x0, mean, std, lp = ops.ddpm_step(
            model_prediction.float().contiguous(), None if prediction_uncond is None else prediction_uncond.float().contiguous(),
            guidance_scale, xt.float().contiguous(), noise, timesteps.to(torch.int32).contiguous(), self._tables(xt.device),
            _MEAN.index(self.mean_type), clamp_x)
        return StepResult(x_prev=xp, estimated_x0=x0, x_prev_mean=mean, x_prev_std=std, logprob=lp)
