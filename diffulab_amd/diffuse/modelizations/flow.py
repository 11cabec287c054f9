"""Rectified-flow modelization (reference diffuse/modelizations/flow.py:16-524) over the HIP kernels.

Host logic (timestep grids, draws, CFG bookkeeping, the sampler loop) stays Python and follows the reference
statement by statement, including its quirks: timesteps are drawn on the CPU global
generator; `t` enters the network unscaled; a constructor `shift` only affects `draw_timesteps` because the
base class builds the grid before `Flow.__init__` stores it; `model_inputs` is mutated in place.
Device arithmetic (noising, loss, x->v, Euler / Euler-Maruyama updates) is `libdiffulab_hip.so`.
GRPO (flow.py:317-380) is out of scope (needs the VLM reward stack).
"""

from __future__ import annotations

from typing import Any

import torch
from torch import Tensor

from ... import ops
from ..samplers.common import StepResult
from ..samplers.flow import Euler, EulerMaruyama
from ..utils import SamplingOutput, to_device
from .diffusion import Diffusion, mse_head

try:
    from tqdm import tqdm
except ImportError:  # pragma: no cover
    def tqdm(it, **kw):
        return it


class _XToV(torch.autograd.Function):
    """v = (z - xhat) / t  (flow.py:300-303)"""

    @staticmethod
    def forward(ctx, xhat: Tensor, z: Tensor, t: Tensor) -> Tensor:
        ctx.save_for_backward(t)
        return ops.flow_x_to_v(z, xhat.float().contiguous(), t)

    @staticmethod
    def backward(ctx, dv: Tensor):
        (t,) = ctx.saved_tensors
        return ops.flow_x_to_v_bwd(dv.float().contiguous(), t), None, None


class Flow(Diffusion):
    sampler_registry = {"euler": Euler, "euler_maruyama": EulerMaruyama}

    def __init__(self, n_steps: int = 50, sampling_method: str = "euler", schedule: str = "linear",
                 latent_diffusion: bool = False, logits_normal: bool = False, shift: float | None = None,
                 sampler_parameters: dict[str, Any] = {}, prediction_type: str = "v") -> None:
        assert prediction_type in ["v", "x"], (
            "prediction_type must be either 'v' or 'x', noise prediction not supported yet for flow models")
        super().__init__(n_steps=n_steps, sampling_method=sampling_method, schedule=schedule,
                         latent_diffusion=latent_diffusion, sampler_parameters=sampler_parameters)
        self.logits_normal = logits_normal
        self.shift = shift  # after the base class already built the (unshifted) grid -- reference behaviour
        self.x_prediction = prediction_type == "x"

    @staticmethod
    def _shift_timestep(t, alpha: float):
        return alpha * t / (1 + (alpha - 1) * t)

    def set_steps(self, n_steps: int, schedule: str = "linear", shift: float | None = None) -> None:
        self.shift = shift
        if schedule != "linear":
            raise NotImplementedError("Only linear schedule is supported for the moment")
        self.schedule = schedule
        grid: list[float] = torch.linspace(1, 0, n_steps + 1).tolist()
        if self.shift is not None:
            grid = [self._shift_timestep(t, self.shift) for t in grid]
        self.timesteps = grid
        self.steps = n_steps
        self.sampler.set_steps(self.timesteps)

    def at(self, timesteps: Tensor) -> Tensor:
        return 1 - timesteps

    def bt(self, timesteps: Tensor) -> Tensor:
        return timesteps

    def draw_timesteps(self, batch_size: int) -> Tensor:
        if self.logits_normal:
            t = torch.sigmoid(torch.randn((batch_size), dtype=torch.float32))
        else:
            t = torch.rand((batch_size), dtype=torch.float32)
        if self.shift is not None:
            t = self._shift_timestep(t, self.shift)
        if self.x_prediction:
            t = t.clamp(min=0.05)
        return t

    def get_v(self, model, model_inputs, t_curr: float) -> Tensor:
        p0 = next(model.parameters())
        timesteps = torch.full((model_inputs["x"].shape[0],), t_curr, device=p0.device, dtype=p0.dtype)
        prediction = model(**model_inputs, timesteps=timesteps)["x"]
        if self.x_prediction:
            return (model_inputs["x"] - prediction) / max(t_curr, 0.05)
        return prediction

    def _get_v_pair(self, model, model_inputs, t_curr: float):
        fn = getattr(model, "forward_cfg_pair", None)
        if fn is None:
            return None
        p0 = next(model.parameters())
        timesteps = torch.full((model_inputs["x"].shape[0],), t_curr, device=p0.device, dtype=p0.dtype)
        pair = fn(timesteps, **model_inputs)
        if pair is None or not self.x_prediction:
            return pair
        return tuple((model_inputs["x"] - pred) / max(t_curr, 0.05) for pred in pair)

    def one_step_denoise(self, model, model_inputs, t_prev: float, t_curr: float, guidance_scale: float,
                         sampler_args: dict[str, Any] = {}) -> StepResult:
        if guidance_scale > 0:
            # flow.py:256-259 makes two forwards (p = 0, then p = 1: every label dropped).  A denoiser whose `p` only reaches the
            # label drop runs them as ONE forward over [x ; x] (FlatArenaDenoiser.forward_cfg_pair: same values per row, half the
            # weight streaming); the combine v_u + g (v - v_u) is fused into the step kernel either way.
            pair = self._get_v_pair(model, model_inputs, t_curr)
            if pair is None:
                pair = (self.get_v(model, {**model_inputs, "p": 0}, t_curr), self.get_v(model, {**model_inputs, "p": 1}, t_curr))
            return self.sampler.step(model_inputs["x"], pair[0], t_curr, t_prev, v_uncond=pair[1],
                                     guidance_scale=guidance_scale, **sampler_args)
        v = self.get_v(model, {**model_inputs, "p": 0}, t_curr)
        return self.sampler.step(model_inputs["x"], v, t_curr, t_prev, **sampler_args)

    def compute_loss(self, model, model_inputs, timesteps: Tensor, noise: Tensor | None = None, extra_losses=[],
                     extra_args: dict[str, Any] = {}) -> dict[str, Tensor]:
        x_0 = model_inputs["x"]  # add_noise writes a fresh tensor, so no clone is needed to keep x_0
        model_inputs["x"], noise = self.add_noise(x_0, timesteps, noise)
        prediction = model(**model_inputs, timesteps=timesteps)
        pred = prediction["x"]
        if self.x_prediction:
            t_dev = to_device(timesteps, pred.device, torch.float32)
            pred = _XToV.apply(pred, model_inputs["x"], t_dev) if pred.requires_grad else ops.flow_x_to_v(
                model_inputs["x"], pred.float().contiguous(), t_dev)
            prediction["x"] = pred
        # mean_b(mean_chw(((eps - x0) - v)^2)) == global mean for equal-size samples (flow.py:306-309)
        loss_dict = {"loss": mse_head(pred, noise, x_0, ops.LOSS_FLOW)}
        for extra_loss in extra_losses:
            loss_dict[extra_loss.name] = extra_loss(**extra_args)
        return loss_dict

    def compute_loss_grpo(self, *a: Any, **k: Any):
        raise NotImplementedError("GRPO fine-tuning (flow.py:317-380) is outside the MI355X hot-path scope (DESIGN.md)")

    def add_noise(self, x: Tensor, timesteps: Tensor, noise: Tensor | None = None) -> tuple[Tensor, Tensor]:
        x = x.float().contiguous()
        if noise is None:
            noise = torch.randn_like(x)
        noise = noise.to(device=x.device, dtype=torch.float32).contiguous()
        assert noise.shape == x.shape
        assert timesteps.shape[0] == x.shape[0]
        t = to_device(timesteps, x.device, torch.float32)
        return ops.flow_add_noise(x, noise, t), noise

    @torch.inference_mode()
    def denoise(self, model, model_inputs, data_shape: tuple[int, ...] | None = None, use_tqdm: bool = True,
                clamp_x: bool = False, guidance_scale: float = 0, sampler_args: dict[str, Any] = {},
                return_intermediates: bool = False) -> SamplingOutput:
        p0 = next(model.parameters())
        if "x" not in model_inputs:
            assert data_shape is not None, "'data_shape' must be provided if 'x' is not in model_inputs"
            model_inputs["x"] = torch.randn(data_shape, device=p0.device, dtype=p0.dtype)
        keep = return_intermediates
        xt = [model_inputs["x"]] if keep else None
        x0s: list[Tensor] = []
        means: list[Tensor] = []
        stds: list[Tensor] = []
        lps: list[Tensor] = []
        pairs = zip(self.timesteps[:-1], self.timesteps[1:])
        for t_curr, t_prev in tqdm(pairs, desc="generating image", total=self.steps, disable=not use_tqdm, leave=False):
            out = self.one_step_denoise(model, model_inputs, t_curr=t_curr, t_prev=t_prev, guidance_scale=guidance_scale,
                                        sampler_args=sampler_args)
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
        if clamp_x:
            model_inputs["x"] = model_inputs["x"].clamp(-1, 1)
        return self._pack(model_inputs["x"], xt, x0s, means, stds, lps, std_dim=0)  # std stacked on dim 0 (flow.py:520)
