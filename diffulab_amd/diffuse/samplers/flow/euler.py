"""Euler ODE step for rectified flow (reference samplers/flow/euler.py:22-41) on the HIP `dl_euler_step` kernel."""

from __future__ import annotations

from torch import Tensor

from .... import ops
from ..common import Sampler, StepResult


class FlowSampler(Sampler):
    name: str


class Euler(FlowSampler):
    name = "euler"

    def set_steps(self, timesteps: list[float]) -> None:
        pass

    def step(self, x_t: Tensor, v: Tensor, t_curr: float, t_prev: float, v_uncond: Tensor | None = None,
             guidance_scale: float = 0.0) -> StepResult:
        """x_prev = x_t - v (t_curr - t_prev); estimated_x0 = x_t - v t_curr.  `v_uncond` (extension) fuses the
        classifier-free combine v_u + g (v - v_u) of flow.py:257-259 into the same pass."""
        x_prev, x0 = ops.euler_step(x_t.float().contiguous(), v.float().contiguous(),
                                    None if v_uncond is None else v_uncond.float().contiguous(), guidance_scale,
                                    t_curr, t_prev)
        return StepResult(x_prev=x_prev, estimated_x0=x0)
