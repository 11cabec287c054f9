"""Oracle: diffusion "modelizations" and sampler steps.  TEST INFRASTRUCTURE ONLY.

Plain numpy (fp64 schedules, integer index sets) and torch fp32 restatement of

  diffuse/modelizations/flow.py:85-135,168-197,199-260,262-315,382-408   (Flow)
  diffuse/modelizations/gaussian_diffusion.py:71-210,268-342             (GaussianDiffusion)
  diffuse/modelizations/utils.py:1-57                                    (space_timesteps)
  diffuse/utils.py:6-19                                                  (extract_into_tensor)
  diffuse/samplers/flow/euler.py:22-41 ; euler_meruyama.py:17-57
  diffuse/samplers/gaussian_diffusion/ddpm.py:49-363 ; ddim.py:28-103

All functions are pure: random draws are passed in (``noise=``) or drawn from the
global torch CPU generator in the same order the reference draws them, so that
``torch.manual_seed(s)`` + oracle == ``torch.manual_seed(s)`` + reference.
"""

from __future__ import annotations

import math

import numpy as np
import torch
from torch import Tensor

# ============================================================================ rectified flow


def flow_shift(t, alpha: float):
    """flow.py:85-99: s(alpha, t) = alpha t / (1 + (alpha - 1) t); works on floats and tensors."""
    return alpha * t / (1 + (alpha - 1) * t)


def flow_timesteps(n_steps: int, shift: float | None = None) -> list[float]:
    """flow.py:127-131: linspace(1, 0, n+1) in fp32 -> python floats, then the shift in fp64."""
    ts = [float(v) for v in torch.linspace(1, 0, n_steps + 1).tolist()]
    if shift is not None:
        ts = [flow_shift(v, shift) for v in ts]
    return ts


def flow_draw_timesteps(batch: int, logits_normal: bool = False, shift: float | None = None,
                        x_prediction: bool = False) -> Tensor:
    """flow.py:184-197: CPU global generator; sigmoid(randn) or rand; shift; clamp for x-pred."""
    if logits_normal:
        t = torch.sigmoid(torch.randn((batch,), dtype=torch.float32))
    else:
        t = torch.rand((batch,), dtype=torch.float32)
    if shift is not None:
        t = flow_shift(t, shift)
    if x_prediction:
        t = t.clamp(min=0.05)
    return t


def _bcast(v: Tensor, like: Tensor) -> Tensor:
    return v.reshape(-1, *([1] * (like.dim() - 1)))


def flow_add_noise(x: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """flow.py:404-407: z_t = (1 - t) x + t eps."""
    tt = _bcast(t, x).to(x.device)
    return (1 - tt) * x + tt * noise


def flow_loss(pred: Tensor, x0: Tensor, noise: Tensor) -> Tensor:
    """flow.py:306-309: mean_b( mean_chw( ((eps - x0) - v)^2 ) )."""
    d = ((noise - x0) - pred) ** 2
    return d.reshape(d.shape[0], -1).mean(dim=-1).mean()


def flow_x_to_v(z_t: Tensor, x_pred: Tensor, t: Tensor) -> Tensor:
    """flow.py:300-303 (x-prediction -> velocity)."""
    return (z_t - x_pred) / _bcast(t, z_t)


def cfg_combine(cond: Tensor, uncond: Tensor, g: float) -> Tensor:
    """flow.py:257-259 / gaussian_diffusion.py:253-255."""
    return uncond + g * (cond - uncond)


def euler_step(x_t: Tensor, v: Tensor, t_curr: float, t_prev: float) -> dict[str, Tensor]:
    """euler.py:37-41."""
    dt = t_curr - t_prev
    return {"x_prev": x_t - v * dt, "estimated_x0": x_t - v * t_curr}


def euler_maruyama_step(x_t: Tensor, v: Tensor, t_curr: float, t_prev: float, tmax: float, eta: float,
                        noise: Tensor | None = None, x_prev: Tensor | None = None) -> dict[str, Tensor]:
    """euler_meruyama.py:39-57.  ``tmax`` = timesteps[1] (:22)."""
    sigma = ((t_curr / (1 - min(t_curr, tmax))) ** 0.5) * eta
    mean = x_t - (v + sigma**2 / (2 * t_curr) * (x_t + (1 - t_curr) * v)) * (t_curr - t_prev)
    std = torch.tensor(sigma * (t_curr - t_prev) ** 0.5)
    if x_prev is None:
        if noise is None:
            noise = torch.randn_like(x_t)
        x_prev = mean + std * noise
    logprob = -((x_prev - mean) ** 2 / (2 * std**2) + torch.log(std) + 0.5 * torch.log(torch.tensor(2 * torch.pi)))
    return {"x_prev": x_prev, "x_prev_mean": mean, "x_prev_std": std, "estimated_x0": x_t - v * t_curr,
            "logprob": logprob}


# ============================================================================ gaussian diffusion


def beta_schedule(n_steps: int, kind: str = "linear") -> np.ndarray:
    """gaussian_diffusion.py:157-194, fp64."""
    if kind == "linear":
        scale = 1000 / n_steps
        # torch.linspace(fp64) == start + i*step evaluated symmetrically; reproduce with torch to stay bit-exact
        return torch.linspace(scale * 0.0001, scale * 0.02, n_steps, dtype=torch.float64).numpy().copy()
    if kind == "cosine":
        f = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2  # noqa: E731
        return np.array([min(1 - f((i + 1) / n_steps) / f(i / n_steps), 0.999) for i in range(n_steps)],
                        dtype=np.float64)
    raise NotImplementedError(kind)


def space_timesteps(num_timesteps: int, section_counts, ddim: bool = False) -> set[int]:
    """modelizations/utils.py:1-57 (IDDPM respacing).

    Quirk kept on purpose: in the ddim branch the ``raise`` sits inside the ``for`` body
    (utils.py:26-31), so only stride 1 is ever tried.
    """
    if ddim:
        assert isinstance(section_counts, int)
        for stride in range(1, num_timesteps):
            if len(range(0, num_timesteps, stride)) == section_counts:
                return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
    counts = [int(c) for c in section_counts.split(",")] if isinstance(section_counts, str) else [section_counts]
    base, extra = divmod(num_timesteps, len(counts))
    start, picked = 0, []
    for i, cnt in enumerate(counts):
        size = base + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError(f"cannot divide section of {size} steps into {cnt}")
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        pos = 0.0
        for _ in range(cnt):
            picked.append(start + round(pos))
            pos += stride
        start += size
    return set(picked)


class GaussianTables:
    """All fp64 tables of GaussianDiffusion.set_steps (gd.py:87-133) + DDPM.set_steps (ddpm.py:49-85)."""

    def __init__(self, training_steps: int = 1000, n_steps: int | None = None, schedule: str = "linear",
                 section_counts=None, ddim: bool = False) -> None:
        n_steps = training_steps if n_steps is None else n_steps
        if n_steps != training_steps:
            section_counts = section_counts or n_steps
        self.steps = n_steps
        betas = beta_schedule(training_steps, schedule)
        self.timestep_map: list[int] = []
        if section_counts:
            use = space_timesteps(training_steps, section_counts, ddim=ddim)
            abar = torch.from_numpy(betas).neg().add(1).cumprod(0)  # torch cumprod: same rounding as the reference
            last = torch.tensor(1.0)
            new = []
            for i, ab in enumerate(abar):
                if i in use:
                    new.append(torch.ones_like(ab) - ab / last)
                    last = ab
                    self.timestep_map.append(i)
            betas = torch.tensor(new).numpy().astype(np.float64)  # torch.tensor(list of 0-d fp64) keeps fp64
        self._set(betas)

    def _set(self, betas: np.ndarray) -> None:
        b = torch.from_numpy(np.asarray(betas, dtype=np.float64))
        one = torch.ones_like(b)
        self.betas = b
        self.alphas = one - b
        self.alphas_bar = self.alphas.cumprod(0)
        self.alphas_bar_prev = torch.cat([torch.tensor([1.0], dtype=torch.float64), self.alphas_bar[:-1]])
        self.sqrt_alphas_bar = self.alphas_bar.sqrt()
        self.posterior_variance = b * (one - self.alphas_bar_prev) / (one - self.alphas_bar)
        self.posterior_log_variance_clipped = torch.log(
            torch.cat([self.posterior_variance[1:2], self.posterior_variance[1:]]))
        self.posterior_mean_coef1 = b * self.alphas_bar_prev.sqrt() / (one - self.alphas_bar)
        self.posterior_mean_coef2 = (one - self.alphas_bar_prev) * self.alphas.sqrt() / (one - self.alphas_bar)


def gather(table: Tensor, t: Tensor, like: Tensor) -> Tensor:
    """diffuse/utils.py:16-19: fp64 gather -> .float() -> broadcast."""
    return _bcast(table[t.long()].float(), like)


def ddpm_draw_timesteps(batch: int, steps: int) -> Tensor:
    """gd.py:210."""
    return torch.randint(0, steps, (batch,), dtype=torch.int32)


def ddpm_add_noise(T: GaussianTables, x: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """gd.py:338-341: sqrt(abar)[t] from the fp64-sqrt table; sqrt(1 - abar[t]) evaluated in fp32."""
    return gather(T.sqrt_alphas_bar, t, x) * x + (torch.ones_like(noise) - gather(T.alphas_bar, t, x)).sqrt() * noise


def mse_loss(pred: Tensor, target: Tensor) -> Tensor:
    """gd.py:306."""
    return ((pred - target) ** 2).mean()


def ddpm_x_start(T: GaussianTables, out: Tensor, xt: Tensor, t: Tensor, mean_type: str) -> Tensor:
    """ddpm.py:87-154."""
    if mean_type == "xstart":
        return out
    if mean_type == "epsilon":
        sab = gather(T.sqrt_alphas_bar, t, xt)
        return (1.0 / sab) * xt - ((torch.ones_like(out) - gather(T.alphas_bar, t, xt)).sqrt() / sab) * out
    if mean_type == "xprev":
        c1 = gather(T.posterior_mean_coef1, t, xt)
        return (1.0 / c1) * out - (gather(T.posterior_mean_coef2, t, xt) / c1) * xt
    raise ValueError(mean_type)


def ddpm_variance(T: GaussianTables, t: Tensor, like: Tensor, var_type: str) -> tuple[Tensor, Tensor]:
    """ddpm.py:200-210 (fixed variants only; learned variances are not on any shipped config)."""
    if var_type == "fixed_small":
        return gather(T.posterior_variance, t, like), gather(T.posterior_log_variance_clipped, t, like)
    if var_type == "fixed_large":
        v = torch.cat([T.posterior_variance[1:2], T.betas[1:]])
        return gather(v, t, like), gather(torch.log(v), t, like)
    raise ValueError(var_type)


def ddpm_step(T: GaussianTables, pred: Tensor, t: Tensor, xt: Tensor, noise: Tensor, mean_type: str = "epsilon",
              var_type: str = "fixed_small", clamp_x: bool = False) -> dict[str, Tensor]:
    """ddpm.py:238-363 with the ``randn_like`` of :302 passed in as ``noise``."""
    x0 = ddpm_x_start(T, pred, xt, t, mean_type)
    if clamp_x:
        x0 = x0.clamp(-1, 1)
    mean = gather(T.posterior_mean_coef1, t, xt) * x0 + gather(T.posterior_mean_coef2, t, xt) * xt
    var, logvar = ddpm_variance(T, t, xt, var_type)
    mask = _bcast((t > 0).float(), xt)
    x_prev = mean + mask * noise * torch.exp(0.5 * logvar)
    vs = var.clamp_min(1e-20)
    logprob = (-((x_prev - mean) ** 2) / (2.0 * vs) - torch.log(2 * torch.pi * vs) * 0.5) * mask
    return {"x_prev": x_prev, "estimated_x0": x0, "x_prev_mean": mean, "x_prev_std": vs.sqrt().expand_as(xt),
            "logprob": logprob}


def ddim_step(T: GaussianTables, pred: Tensor, t: Tensor, xt: Tensor, noise: Tensor, eta: float = 0.0,
              mean_type: str = "epsilon", clamp_x: bool = False) -> dict[str, Tensor]:
    """ddim.py:28-103."""
    x0 = ddpm_x_start(T, pred, xt, t, mean_type)
    if clamp_x:
        x0 = x0.clamp(-1, 1)
    one = torch.ones_like(xt)
    ab, abp = gather(T.alphas_bar, t, xt), gather(T.alphas_bar_prev, t, xt)
    eps = ((1 / gather(T.sqrt_alphas_bar, t, xt)) * xt - x0) / (1 / ab - 1).sqrt()  # ddpm.py:324-326
    sigma = eta * ((one - abp) / (one - ab)).sqrt() * (one - ab / abp).sqrt()
    mean = x0 * abp.sqrt() + (one - abp - sigma**2).sqrt() * eps
    mask = _bcast((t > 0).float(), xt)
    x_prev = mean + mask * sigma * noise
    out = {"x_prev": x_prev, "estimated_x0": x0, "x_prev_mean": mean}
    if eta > 0:
        out["x_prev_std"] = sigma
        out["logprob"] = -((x_prev - mean) ** 2 / (2 * sigma**2) + torch.log(sigma)
                           + 0.5 * torch.log(torch.tensor(2 * torch.pi)))
    return out
