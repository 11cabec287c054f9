"""DDPM / IDDPM modelization (reference diffuse/modelizations/gaussian_diffusion.py:18-447) over the HIP kernels.

Schedules are built on the host in fp64 with the same torch ops as the reference (bit-identical tables and
`timestep_map`); the device kernels read their fp32 casts.  Timesteps are int32 indices drawn on the CPU global
generator and are fed to the denoiser unscaled (after the `timestep_map` remap when the chain is respaced).
"""

from __future__ import annotations

import math
from typing import Any, Callable

import torch
from torch import Tensor

from ... import ops
from ..samplers.common import StepResult
from ..samplers.gaussian_diffusion import DDIM, DDPM
from ..utils import SamplingOutput, f32_table, to_device
from .diffusion import Diffusion, mse_head
from .utils import space_timesteps

try:
    from tqdm import tqdm
except ImportError:  # pragma: no cover
    def tqdm(it, **kw):
        return it


class GaussianDiffusion(Diffusion):
    sampler_registry = {"ddpm": DDPM, "ddim": DDIM}

    def __init__(self, n_steps: int = 1000, sampling_method: str = "ddpm", schedule: str = "linear",
                 latent_diffusion: bool = False, sampler_parameters: dict[str, Any] = {}) -> None:
        if sampling_method not in ["ddpm", "ddim"]:
            raise ValueError("sampling method must be one of ['ddpm', 'ddim']")
        self.training_steps = n_steps
        self._dev: dict[torch.device, tuple[Tensor, Tensor]] = {}
        super().__init__(n_steps=self.training_steps, sampling_method=sampling_method, schedule=schedule,
                         latent_diffusion=latent_diffusion, sampler_parameters=sampler_parameters)

    def set_diffusion_parameters(self, betas: Tensor) -> None:
        self.betas = betas
        self.alphas = torch.ones_like(betas) - betas
        self.alphas_bar = self.alphas.cumprod(dim=0)
        self.sqrt_alphas_bar = self.alphas_bar.sqrt()
        self._dev.clear()
        self.sampler.set_steps(betas)

    def set_steps(self, n_steps: int, schedule: str = "linear", section_counts: int | str | None = None) -> None:
        if n_steps != self.training_steps:
            section_counts = section_counts or n_steps
        self.steps = n_steps
        self.set_diffusion_parameters(self._get_variance_schedule(self.training_steps, schedule))
        self.timestep_map: list[int] = []
        if section_counts:
            keep = space_timesteps(num_timesteps=self.training_steps, section_counts=section_counts,
                                   ddim=self.sampling_method == "ddim")
            prev_bar = torch.tensor(1.0)
            respaced: list[Tensor] = []
            for i, bar in enumerate(self.alphas_bar):
                if i in keep:
                    respaced.append(torch.ones_like(bar) - bar / prev_bar)
                    prev_bar = bar
                    self.timestep_map.append(i)
            self.set_diffusion_parameters(torch.tensor(respaced))

    def _get_variance_schedule(self, n_steps: int, variance_schedule: str = "linear") -> Tensor:
        if variance_schedule == "linear":  # Ho et al., rescaled to any chain length
            k = 1000 / n_steps
            return torch.linspace(k * 0.0001, k * 0.02, n_steps, dtype=torch.float64, requires_grad=False)
        if variance_schedule == "cosine":
            return self._betas_for_alpha_bar(n_steps, lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2)
        raise NotImplementedError(f"unknown beta schedule: {variance_schedule}")

    def _betas_for_alpha_bar(self, n_steps: int, alpha_bar: Callable[[float], float], max_beta: float = 0.999) -> Tensor:
        vals = [min(1 - alpha_bar((i + 1) / n_steps) / alpha_bar(i / n_steps), max_beta) for i in range(n_steps)]
        return torch.tensor(vals, dtype=torch.float64, requires_grad=False)

    def draw_timesteps(self, batch_size: int) -> Tensor:
        return torch.randint(0, self.steps, (batch_size,), dtype=torch.int32)

    def _remap(self, timesteps: Tensor) -> Tensor:
        if self.timestep_map:
            table = torch.tensor(self.timestep_map, device=timesteps.device, dtype=timesteps.dtype)
            return table[timesteps]
        return timesteps

    def one_step_denoise(self, model, model_inputs, t: int, clamp_x: bool = False, guidance_scale: float = 0.0,
                         sampler_args: dict[str, Any] = {}) -> StepResult:
        device = next(model.parameters()).device
        timesteps = torch.full((model_inputs["x"].shape[0],), t, device=device, dtype=torch.int32)
        t_model = self._remap(timesteps)
        extra = {}
        pair = None
        if guidance_scale > 0:  # CFG combine fused into the step kernel; the two forwards as one where the denoiser can (see Flow)
            fn = getattr(model, "forward_cfg_pair", None)
            pair = fn(t_model, **model_inputs) if fn is not None else None
        if pair is not None:
            prediction, extra = pair[0], {"prediction_uncond": pair[1], "guidance_scale": guidance_scale}
        else:
            prediction = model(**{**model_inputs, "p": 0}, timesteps=t_model)["x"]
            if guidance_scale > 0:
                extra = {"prediction_uncond": model(**{**model_inputs, "p": 1}, timesteps=t_model)["x"],
                         "guidance_scale": guidance_scale}
        return self.sampler.step(model_prediction=prediction, timesteps=timesteps, xt=model_inputs["x"], clamp_x=clamp_x,
                                 **extra, **sampler_args)

    def compute_loss(self, model, model_inputs, timesteps: Tensor, noise: Tensor | None = None, extra_losses=[],
                     extra_args: dict[str, Any] = {}) -> dict[str, Tensor]:
        model_inputs["x"], noise = self.add_noise(model_inputs["x"], timesteps, noise)
        prediction = model(**model_inputs, timesteps=self._remap(timesteps))["x"]
        loss_dict = {"loss": mse_head(prediction, noise, None, ops.LOSS_EPS)}
        for extra_loss in extra_losses:
            loss_dict[extra_loss.name] = extra_loss(**extra_args)
        return loss_dict

    def add_noise(self, x: Tensor, timesteps: Tensor, noise: Tensor | None = None) -> tuple[Tensor, Tensor]:
        x = x.float().contiguous()
        if noise is None:
            noise = torch.randn_like(x)
        noise = noise.to(device=x.device, dtype=torch.float32).contiguous()
        assert noise.shape == x.shape
        assert timesteps.shape[0] == x.shape[0]
        if x.device not in self._dev:
            self._dev[x.device] = (f32_table(self.sqrt_alphas_bar, x.device), f32_table(self.alphas_bar, x.device))
        sab, ab = self._dev[x.device]
        return ops.ddpm_add_noise(x, noise, to_device(timesteps, x.device, torch.int32), sab, ab), noise

    # The reference leaves autograd on in this loop (gaussian_diffusion.py:344-447 carries no inference_mode / no_grad, unlike
    # Flow.denoise): outside the trainer's @no_grad image logging it would record a graph through all 1000 steps.  Here the
    # loop runs under no_grad: same values, and the denoiser takes its inference path (hipGraph replay, no activation keeping).
    @torch.no_grad()
    def denoise(self, model, model_inputs, data_shape: tuple[int, ...] | None = None, use_tqdm: bool = True,
                clamp_x: bool = False, guidance_scale: float = 0, sampler_args: dict[str, Any] = {},
                return_intermediates: bool = False) -> SamplingOutput:
        if "x" not in model_inputs:
            assert data_shape is not None, "'data_shape' must be provided if 'x' is not in model_inputs"
            p0 = next(model.parameters())
            model_inputs["x"] = torch.randn(data_shape, device=p0.device, dtype=p0.dtype)
        keep = return_intermediates
        xt = [model_inputs["x"]] if keep else None
        x0s: list[Tensor] = []
        means: list[Tensor] = []
        stds: list[Tensor] = []
        lps: list[Tensor] = []
        for t in tqdm(list(range(self.steps))[::-1], desc="generating image", total=self.steps, disable=not use_tqdm,
                      leave=False):
            out = self.one_step_denoise(model=model, model_inputs=model_inputs, t=t, clamp_x=clamp_x,
                                        guidance_scale=guidance_scale, sampler_args=sampler_args)
            model_inputs["x"] = out["x_prev"]
            if keep:
                xt.append(out["x_prev"])
                x0s.append(out["estimated_x0"])
                if "x_prev_mean" in out:
                    means.append(out["x_prev_mean"])
                if "x_prev_std" in out:
                    stds.append(out["x_prev_std"])
                if "logprob" in out:
                    lps.append(out["logprob"])
        return self._pack(model_inputs["x"], xt, x0s, means, stds, lps, std_dim=1)
