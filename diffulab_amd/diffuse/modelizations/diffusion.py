"""`Diffusion` plugin base (reference diffuse/modelizations/diffusion.py:13-244): owns the sampler built from
`sampler_registry`, the step grid, and declares the six methods a modelization implements."""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any

import torch
from torch import Tensor

from ... import ops
from ..samplers.common import Sampler, StepResult
from ..utils import SamplingOutput


class _MSEHead(torch.autograd.Function):
    """loss = mean((target - pred)^2) with target = a - b (flow) or a (eps) -- dl_mse_loss_fwd / _bwd"""

    @staticmethod
    def forward(ctx, pred: Tensor, a: Tensor, b: Tensor | None, mode: int) -> Tensor:
        pred = pred.float().contiguous()
        ctx.save_for_backward(pred, a, b)
        ctx.mode = mode
        return ops.mse_loss_fwd(pred, a, b, mode)

    @staticmethod
    def backward(ctx, gout: Tensor):
        pred, a, b = ctx.saved_tensors
        # the upstream gradient stays on the device (no .item() sync): the kernel multiplies by *gscale_dev
        gdev = gout.float().contiguous() if gout is not None else None
        return ops.mse_loss_bwd(pred, a, b, 1.0, ctx.mode, gscale_dev=gdev), None, None, None


def mse_head(pred: Tensor, a: Tensor, b: Tensor | None, mode: int) -> Tensor:
    a = a.float().contiguous()
    b = None if b is None else b.float().contiguous()
    if pred.requires_grad:
        return _MSEHead.apply(pred, a, b, mode)
    return ops.mse_loss_fwd(pred.float().contiguous(), a, b, mode)


class Diffusion(ABC):
    sampler_registry: dict[str, type[Sampler]]

    def __init__(self, n_steps: int, sampling_method: str = "euler", schedule: str = "linear",
                 latent_diffusion: bool = False, sampler_parameters: dict[str, Any] = {}) -> None:
        assert sampling_method in self.sampler_registry, (
            f"Unknown sampling method '{sampling_method}'. Available methods: {list(self.sampler_registry.keys())}")
        self.sampler = self.sampler_registry[sampling_method](**sampler_parameters)
        self.timesteps: list[float] = []
        self.steps: int = n_steps
        self.sampling_method = sampling_method
        self.schedule = schedule
        self.latent_diffusion = latent_diffusion
        self.set_steps(n_steps, schedule=schedule)

    @abstractmethod
    def set_steps(self, n_steps: int, schedule: str) -> None: ...

    @abstractmethod
    def one_step_denoise(self, model, model_inputs, guidance_scale: float, *args: Any, **kwargs: Any) -> StepResult: ...

    @abstractmethod
    def compute_loss(self, model, model_inputs, timesteps: Tensor, noise: Tensor | None = None, extra_losses=[],
                     extra_args: dict[str, Any] = {}) -> dict[str, Tensor]: ...

    @abstractmethod
    def add_noise(self, x: Tensor, timesteps: Tensor, noise: Tensor | None = None) -> tuple[Tensor, Tensor]: ...

    @abstractmethod
    def denoise(self, model, model_inputs, data_shape=None, use_tqdm: bool = True, clamp_x: bool = False,
                guidance_scale: float = 0, sampler_args: dict[str, Any] = {},
                return_intermediates: bool = False) -> SamplingOutput: ...

    @abstractmethod
    def draw_timesteps(self, batch_size: int) -> Tensor: ...

    # shared by both heads: stack what the sampler loop collected (flow.py:510-522 / gaussian_diffusion.py:436-445)
    @staticmethod
    def _pack(x: Tensor, xt, x0, mean, std, lp, std_dim: int) -> SamplingOutput:
        out: SamplingOutput = {"x": x}
        if xt is not None:
            out["xt"] = torch.stack(xt, dim=1)
            out["estimated_x0"] = torch.stack(x0, dim=1)
            if mean:
                out["xt_mean"] = torch.stack(mean, dim=1)
            if std:
                out["xt_std"] = torch.stack(std, dim=std_dim)
            if lp:
                out["logprob"] = torch.stack(lp, dim=1)
        return out
