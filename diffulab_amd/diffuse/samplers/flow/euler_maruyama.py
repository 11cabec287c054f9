"""Euler-Maruyama SDE step (reference samplers/flow/euler_meruyama.py:8-57) on `dl_euler_maruyama_step`."""

from __future__ import annotations

import torch
from torch import Tensor

from .... import ops
from ..common import StepResult
from .euler import FlowSampler


class EulerMaruyama(FlowSampler):
    name = "euler_maruyama"

    def __init__(self, eta: float = 0.7) -> None:
        super().__init__()
        self.eta = eta
        self.tmax: float | None = None

    def set_steps(self, timesteps: list[float]) -> None:
        self.tmax = timesteps[1]

    def step(self, x_t: Tensor, v: Tensor, t_curr: float, t_prev: float, x_prev: Tensor | None = None,
             v_uncond: Tensor | None = None, guidance_scale: float = 0.0) -> StepResult:
        assert self.tmax is not None, "set_steps must be called before step"
        sigma = ((t_curr / (1 - min(t_curr, self.tmax))) ** 0.5) * self.eta  # host scalars, as in the reference
        noise = torch.randn_like(x_t) if x_prev is None else None         # device RNG draw, same call as :44
        xp, mean, x0, lp, std = ops.euler_maruyama_step(
            x_t.float().contiguous(), v.float().contiguous(), None if v_uncond is None else v_uncond.float().contiguous(),
            guidance_scale, noise, None if x_prev is None else x_prev.float().contiguous(), t_curr, t_prev, sigma)
        return StepResult(x_prev=xp, x_prev_mean=mean, x_prev_std=torch.tensor(std, device=x_t.device), estimated_x0=x0,
                          logprob=lp)
