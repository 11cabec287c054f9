"""`Diffuser` facade (reference diffuse/diffuser.py:14-239): picks the modelization from `model_registry` and
forwards draw_timesteps / compute_loss / set_steps / generate to it, same signatures."""

from __future__ import annotations

from typing import Any

from torch import Tensor

from .modelizations.diffusion import Diffusion
from .modelizations.flow import Flow
from .modelizations.gaussian_diffusion import GaussianDiffusion
from .utils import SamplingOutput


class Diffuser:
    model_registry: dict[str, type[Diffusion]] = {"rectified_flow": Flow, "gaussian_diffusion": GaussianDiffusion}

    def __init__(self, denoiser, sampling_method: str, model_type: str = "rectified_flow", n_steps: int = 1000,
                 vision_tower=None, extra_args: dict[str, Any] = {}, extra_losses: list = []) -> None:
        self.model_type = model_type
        self.denoiser = denoiser
        self.n_steps = n_steps
        self.vision_tower = vision_tower
        self.extra_losses = extra_losses
        if self.vision_tower:
            self.latent_scale = self.vision_tower.latent_scale
            self.latent_bias = self.vision_tower.latent_bias
        if self.model_type not in self.model_registry:
            raise NotImplementedError(f"Model type {self.model_type} is not implemented")
        self.diffusion = self.model_registry[self.model_type](
            n_steps=n_steps, sampling_method=sampling_method, latent_diffusion=self.vision_tower is not None, **extra_args)

    def eval(self) -> None:
        self.denoiser.eval()

    def train(self) -> None:
        self.denoiser.train()

    def draw_timesteps(self, batch_size: int) -> Tensor:
        return self.diffusion.draw_timesteps(batch_size=batch_size)

    def compute_loss(self, model_inputs, timesteps: Tensor | None = None, noise: Tensor | None = None,
                     extra_args: dict[str, Any] = {}, grpo: bool = False, grpo_args: dict[str, Any] = {}) -> dict[str, Tensor]:
        if grpo:
            assert isinstance(self.diffusion, Flow), "GRPO loss computation is only available for Flow-based models"
            return self.diffusion.compute_loss_grpo(self.denoiser, model_inputs, **grpo_args)
        assert timesteps is not None, "timesteps must be provided for loss computation"
        return self.diffusion.compute_loss(self.denoiser, model_inputs, timesteps, noise, self.extra_losses, extra_args)

    def set_steps(self, n_steps: int, schedule: str = "linear", *args: Any, **kwargs: Any) -> None:
        self.diffusion.set_steps(n_steps, schedule=schedule, *args, **kwargs)

    def generate(self, model_inputs, data_shape: tuple[int, ...] | None = None, use_tqdm: bool = True,
                 clamp_x: bool = False, guidance_scale: float = 0, sampler_args: dict[str, Any] = {},
                 return_intermediates: bool = False, return_latents: bool = False) -> SamplingOutput:
        out = self.diffusion.denoise(self.denoiser, model_inputs=model_inputs, data_shape=data_shape, use_tqdm=use_tqdm,
                                     clamp_x=clamp_x, guidance_scale=guidance_scale, sampler_args=sampler_args,
                                     return_intermediates=return_intermediates)
        if self.vision_tower and not return_latents:  # decode latents with the (external) VAE, diffuser.py:220-227
            scale, bias = self.latent_scale, self.latent_bias
            if isinstance(scale, Tensor):
                scale = scale.to(out["x"].device)
            if isinstance(bias, Tensor):
                bias = bias.to(out["x"].device)
            out["x"] = self.vision_tower.decode(out["x"] / scale + bias)
        return out
