"""Generate the golden vectors under tests/golden/ by IMPORTING THE REAL REFERENCE.

Runs only in the build container (``/root/reference`` does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference (LouisRouss/DiffuLab @ 2026-01-28) ships no tests and no golden vectors
(SURVEY.md §4), so these files are what pins the oracle.  Only OUTPUTS of the reference are
stored; inputs / weights are regenerated from ``oracle.synth`` (numpy PCG64) by the tests.
Nothing of the reference's source text is copied here: the script only *calls* it.

Import shims: python 3.10 lacks typing.NotRequired; jaxtyping and the heavy
optional deps of the package ``__init__``s are not installed -> stub them before import.
"""

from __future__ import annotations

import os
import sys
import types
import typing

sys.dont_write_bytecode = True
REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import typing_extensions  # noqa: E402


def install_shims() -> None:
    for name in ("NotRequired", "Required"):
        if not hasattr(typing, name):
            setattr(typing, name, getattr(typing_extensions, name))

    class _Any:
        def __class_getitem__(cls, item):
            return typing.Any

    jt = types.ModuleType("jaxtyping")
    for n in ("Float", "Int", "Bool"):
        setattr(jt, n, _Any)
    sys.modules["jaxtyping"] = jt

    def ns(name: str, path: str) -> types.ModuleType:
        m = types.ModuleType(name)
        m.__path__ = [path]  # type: ignore[attr-defined]
        sys.modules[name] = m
        return m

    root = os.path.join(REF, "diffulab")
    ns("diffulab", root)
    ns("diffulab.networks", f"{root}/networks")
    ns("diffulab.networks.embedders", f"{root}/networks/embedders")
    ns("diffulab.networks.vision_towers", f"{root}/networks/vision_towers")
    ns("diffulab.training", f"{root}/training")
    ns("diffulab.training.losses", f"{root}/training/losses")
    ns("diffulab.diffuse", f"{root}/diffuse")
    vt = types.ModuleType("diffulab.networks.vision_towers.common")

    class VisionTower:  # only used as a type annotation on the path
        pass

    vt.VisionTower = VisionTower
    sys.modules["diffulab.networks.vision_towers.common"] = vt
    import importlib

    lc = importlib.import_module("diffulab.training.losses.common")
    sys.modules["diffulab.training.losses"].LossFunction = lc.LossFunction


install_shims()

from diffulab.diffuse.diffuser import Diffuser  # noqa: E402
from diffulab.diffuse.modelizations.flow import Flow  # noqa: E402
from diffulab.diffuse.modelizations.gaussian_diffusion import GaussianDiffusion  # noqa: E402
from diffulab.diffuse.modelizations.utils import space_timesteps  # noqa: E402
from diffulab.diffuse.samplers.flow import Euler, EulerMaruyama  # noqa: E402
from diffulab.diffuse.samplers.gaussian_diffusion import DDIM, DDPM  # noqa: E402
from diffulab.networks.denoisers.mmdit import DiTBlock, MMDiT  # noqa: E402
from diffulab.networks.denoisers.unet import AttentionBlock, ResBlock, UNetModel  # noqa: E402
from diffulab.networks.utils.nn import get_cos_sin_ndim_grid, timestep_embedding  # noqa: E402

from oracle import synth  # noqa: E402
from oracle.dit import DiTConfig, param_shapes  # noqa: E402
from oracle import unet as ounet  # noqa: E402

torch.set_num_threads(8)


def save(name: str, **arrs) -> None:
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays)")


# ------------------------------------------------------------------ (i)+(ii) schedules and draws
def gen_schedules() -> None:
    o: dict[str, object] = {}
    for n in (4, 50, 100):
        o[f"flow_ts_n{n}"] = np.array(Flow(n_steps=n).timesteps, dtype=np.float64)
        for sh in (4.63, 6.93):
            f = Flow(n_steps=n, shift=sh)
            # quirk: Diffusion.__init__ calls set_steps(n, schedule) BEFORE Flow.__init__ stores `shift`
            # (flow.py:72-81 vs :118), so a constructor shift never reaches `timesteps`; only set_steps(shift=) does.
            o[f"flow_ts_ctor_n{n}_shift{sh}"] = np.array(f.timesteps, dtype=np.float64)
            f.set_steps(n, shift=sh)
            o[f"flow_ts_n{n}_shift{sh}"] = np.array(f.timesteps, dtype=np.float64)
    for seed in (0, 1, 2):
        for B in (4, 64):
            for tag, kw in (("uniform", {}), ("logit", {"logits_normal": True}),
                            ("logit_shift", {"logits_normal": True, "shift": 4.63}),
                            ("xpred", {"prediction_type": "x"})):
                torch.manual_seed(seed)
                o[f"draw_flow_{tag}_s{seed}_b{B}"] = Flow(n_steps=10, **kw).draw_timesteps(B)
            torch.manual_seed(seed)
            o[f"draw_ddpm_s{seed}_b{B}"] = GaussianDiffusion(n_steps=1000).draw_timesteps(B)
    for sched in ("linear", "cosine"):
        g = GaussianDiffusion(n_steps=1000, schedule=sched)
        s = g.sampler
        for nm in ("betas", "alphas_bar", "sqrt_alphas_bar"):
            o[f"gd_{sched}_{nm}"] = getattr(g, nm)
        for nm in ("alphas_bar_prev", "posterior_variance", "posterior_log_variance_clipped",
                   "posterior_mean_coef1", "posterior_mean_coef2"):
            o[f"gd_{sched}_{nm}"] = getattr(s, nm)
    for n in (50, 100, 250):
        g = GaussianDiffusion(n_steps=1000)
        g.set_steps(n)
        o[f"respace_{n}_betas"] = g.betas
        o[f"respace_{n}_map"] = np.array(g.timestep_map, dtype=np.int64)
        o[f"respace_{n}_postvar"] = g.sampler.posterior_variance
    g = GaussianDiffusion(n_steps=1000)
    g.set_steps(30, section_counts="10,10,10")
    o["respace_sections_map"] = np.array(g.timestep_map, dtype=np.int64)
    o["space_1000_10"] = np.array(sorted(space_timesteps(1000, 10)), dtype=np.int64)
    try:
        space_timesteps(1000, 10, ddim=True)
        o["space_ddim_raises"] = np.array(0)
    except ValueError:
        o["space_ddim_raises"] = np.array(1)
    o["space_ddim_full"] = np.array(sorted(space_timesteps(1000, 1000, ddim=True)), dtype=np.int64)
    save("schedules", **o)


# ------------------------------------------------------------------ (iii) embeddings / rope
def gen_prims() -> None:
    t = synth.uniform("prims.t", (8,), lo=0.0, hi=1.0)
    ti = torch.tensor([0, 1, 17, 500, 999], dtype=torch.int32)
    pos = torch.stack(torch.meshgrid([torch.arange(16), torch.arange(16)], indexing="ij"), dim=-1).view(-1, 2)[None]
    cos, sin = get_cos_sin_ndim_grid(pos, base=10_000, axes_dim=[32, 32])
    pos2 = torch.stack(torch.meshgrid([torch.arange(3), torch.arange(5)], indexing="ij"), dim=-1).view(-1, 2)[None]
    cos2, sin2 = get_cos_sin_ndim_grid(pos2, base=2000, axes_dim=[8, 24])
    save("prims", temb_f=timestep_embedding(t, 256), temb_i=timestep_embedding(ti, 128), temb_odd=timestep_embedding(t, 9),
         rope_cos_16x16=cos[0], rope_sin_16x16=sin[0], rope_cos_3x5=cos2[0], rope_sin_3x5=sin2[0])


# ------------------------------------------------------------------ (iv)+(v) DiT block / model
SMALL = DiTConfig(input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4,
                  patch_size=2, depth=2, n_classes=10, classifier_free=True)
S2 = DiTConfig()  # DiT-S/2 measurement config


def build_ref(cfg: DiTConfig, seed: int) -> MMDiT:
    m = MMDiT(simple_dit=True, input_channels=cfg.input_channels, output_channels=cfg.output_channels,
              inner_dim=cfg.inner_dim, embedding_dim=cfg.embedding_dim, num_heads=cfg.num_heads,
              mlp_ratio=cfg.mlp_ratio, patch_size=cfg.patch_size, depth=cfg.depth, n_classes=cfg.n_classes,
              classifier_free=cfg.classifier_free)
    P = synth.dit_params(param_shapes(cfg), seed=seed)
    sd = m.state_dict()
    assert set(sd) == set(P), (set(sd) ^ set(P))
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), k
    m.load_state_dict(P)
    return m


def gen_block() -> None:
    cfg = SMALL
    m = build_ref(cfg, seed=3)
    blk: DiTBlock = m.layers[1]
    B, gh, gw = 2, 4, 4
    x = synth.normal("blk.x", (B, gh * gw, cfg.inner_dim)).requires_grad_(True)
    emb = synth.normal("blk.emb", (B, cfg.embedding_dim)).requires_grad_(True)
    pos = torch.stack(torch.meshgrid([torch.arange(gh), torch.arange(gw)], indexing="ij"), dim=-1).view(-1, 2)[None]
    cs = get_cos_sin_ndim_grid(pos.repeat(B, 1, 1), base=cfg.rope_base, axes_dim=cfg.rope_axes_dim)
    taps = {}
    hooks = [blk.attention.register_forward_hook(lambda mod, i, o: taps.__setitem__("attn_proj", o)),
             blk.norm_1.register_forward_hook(lambda mod, i, o: taps.__setitem__("norm1", o)),
             blk.mlp_input[1].register_forward_hook(lambda mod, i, o: taps.__setitem__("mlp_hidden", o))]
    y = blk(x, emb, cs)
    w = synth.normal("blk.dy", tuple(y.shape))
    (y * w).sum().backward()
    for h in hooks:
        h.remove()
    o = {"y": y, "dx": x.grad, "demb": emb.grad, **{"tap_" + k: v for k, v in taps.items()}}
    for n, p in blk.named_parameters():
        o["g_" + n] = p.grad
    save("dit_block", **o)


def flow_loss_ref(m: MMDiT, x0, t, y, noise, p=0.0):
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=50)
    return d.compute_loss({"x": x0.clone(), "y": y, "p": p}, timesteps=t, noise=noise)["loss"]


def gen_small_model() -> None:
    cfg = SMALL
    m = build_ref(cfg, seed=5)
    B, H = 3, 8
    x0 = synth.normal("small.x0", (B, cfg.input_channels, H, H))
    noise = synth.normal("small.noise", (B, cfg.input_channels, H, H))
    t = synth.uniform("small.t", (B,), lo=0.02, hi=0.98)
    y = synth.integers("small.y", (B,), cfg.n_classes)
    # plain forward at z_t (oracle test recomputes z_t itself)
    z = (1 - t.view(-1, 1, 1, 1)) * x0 + t.view(-1, 1, 1, 1) * noise
    pred = m(x=z, timesteps=t, y=y, p=0.0)["x"]
    loss = flow_loss_ref(m, x0, t, y, noise)
    loss.backward()
    o = {"pred": pred, "loss": loss}
    for n, p in m.named_parameters():
        o["g_" + n] = p.grad
    # unconditional (all labels dropped, p=1) forward for CFG parity
    o["pred_uncond"] = m(x=z, timesteps=t, y=y, p=1.0)["x"]
    # DDPM loss on the same net (timesteps are int32 indices fed unscaled, Appendix C.7)
    m.zero_grad()
    gd = Diffuser(m, sampling_method="ddpm", model_type="gaussian_diffusion", n_steps=1000)
    ti = torch.tensor([3, 500, 999], dtype=torch.int32)
    l2 = gd.compute_loss({"x": x0.clone(), "y": y, "p": 0.0}, timesteps=ti, noise=noise)["loss"]
    l2.backward()
    o["ddpm_loss"] = l2
    o["ddpm_g_conv_proj.weight"] = m.conv_proj.weight.grad
    o["ddpm_xt"] = gd.diffusion.add_noise(x0, ti, noise)[0]
    save("dit_small", **o)


def gen_small16() -> None:
    """same small net on a 16x16 latent grid (64 tokens): the smallest shape the HIP attention kernel accepts, so
    the GPU path itself can be compared with reference outputs (not only with the oracle)."""
    cfg = SMALL
    m = build_ref(cfg, seed=5)
    B, H = 4, 16
    x0 = synth.normal("s16.x0", (B, cfg.input_channels, H, H))
    noise = synth.normal("s16.noise", (B, cfg.input_channels, H, H))
    t = synth.uniform("s16.t", (B,), lo=0.02, hi=0.98)
    y = synth.integers("s16.y", (B,), cfg.n_classes)
    z = (1 - t.view(-1, 1, 1, 1)) * x0 + t.view(-1, 1, 1, 1) * noise
    o = {"pred": m(x=z, timesteps=t, y=y, p=0.0)["x"]}
    loss = flow_loss_ref(m, x0, t, y, noise)
    loss.backward()
    o["loss"] = loss
    for n, p in m.named_parameters():
        o["g_" + n] = p.grad
    m.eval()
    x_init = synth.normal("s16.init", (B, 4, H, H))
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    out = d.generate({"x": x_init.clone(), "y": y}, use_tqdm=False, guidance_scale=2.0, return_intermediates=True)
    o["loop_euler_x"], o["loop_euler_x0"] = out["x"], out["estimated_x0"]
    g = Diffuser(m, sampling_method="ddpm", model_type="gaussian_diffusion", n_steps=1000)
    g.set_steps(5)
    torch.manual_seed(23)
    with torch.no_grad():
        o["loop_ddpm_x"] = g.generate({"x": x_init.clone(), "y": y}, use_tqdm=False, guidance_scale=1.5, clamp_x=True)["x"]
    save("dit_small16", **o)


def gen_s2_model() -> None:
    cfg = S2
    m = build_ref(cfg, seed=7)
    B = 2
    x0 = synth.normal("s2.x0", (B, 4, 32, 32))
    noise = synth.normal("s2.noise", (B, 4, 32, 32))
    t = synth.uniform("s2.t", (B,), lo=0.05, hi=0.95)
    y = synth.integers("s2.y", (B,), 1000)
    z = (1 - t.view(-1, 1, 1, 1)) * x0 + t.view(-1, 1, 1, 1) * noise
    pred = m(x=z, timesteps=t, y=y, p=0.0)["x"]
    loss = flow_loss_ref(m, x0, t, y, noise)
    loss.backward()
    o = {"pred": pred, "loss": loss}
    names, norms = [], []
    for n, p in m.named_parameters():
        names.append(n)
        norms.append(p.grad.double().norm().item())
        if p.grad.numel() <= 4096:
            o["g_" + n] = p.grad
        else:
            o["gs_" + n] = p.grad.flatten()[:: max(1, p.grad.numel() // 512)][:512]  # strided sample
    o["grad_names"] = np.array(names)
    o["grad_norms"] = np.array(norms)
    save("dit_s2", **o)


def gen_autocast() -> None:
    """THE bf16 YARDSTICK, from the reference itself: the imported reference under
    ``torch.autocast("cpu", dtype=torch.bfloat16)`` -- what accelerate's bf16 mixed precision runs (trainers/common.py:103-109,
    SURVEY Appendix D) -- on the inputs of the `dit_small16` and `dit_s2` fixtures; stored: the autocast loss and, per parameter,
    the relative L2 error of the autocast gradient against the reference's OWN fp32 gradient.  tests/test_oracle_golden.py checks that ``oracle.dit.bf16_autocast()`` reproduces these numbers,
    tests/test_parity_bf16_gpu.py bounds the HIP path by them."""
    o = {}
    for tag, cfg, seed, B, H, lo in (("s16", SMALL, 5, 4, 16, 0.02), ("s2", S2, 7, 2, 32, 0.05)):
        x0 = synth.normal(f"{tag}.x0", (B, cfg.input_channels, H, H))
        noise = synth.normal(f"{tag}.noise", (B, cfg.input_channels, H, H))
        t = synth.uniform(f"{tag}.t", (B,), lo=lo, hi=1.0 - lo)
        y = synth.integers(f"{tag}.y", (B,), cfg.n_classes)
        legs = {}
        for leg in ("fp32", "autocast"):
            m = build_ref(cfg, seed=seed)
            if leg == "autocast":
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    loss = flow_loss_ref(m, x0, t, y, noise)
            else:
                loss = flow_loss_ref(m, x0, t, y, noise)
            loss.backward()
            legs[leg] = (loss.detach().float(), {n: p.grad.detach().float().clone() for n, p in m.named_parameters()})
        names = list(legs["fp32"][1])
        errs = [((legs["autocast"][1][n].double() - legs["fp32"][1][n].double()).norm() / legs["fp32"][1][n].double().norm()).item() for n in names]
        o[f"{tag}_names"], o[f"{tag}_err"] = np.array(names), np.array(errs)
        o[f"{tag}_loss_fp32"], o[f"{tag}_loss_autocast"] = legs["fp32"][0], legs["autocast"][0]
        e = np.array(errs)
        print(f"{tag}: loss fp32 {legs['fp32'][0].item():.6f} autocast {legs['autocast'][0].item():.6f}; per-tensor error median {np.median(e):.3e} "
              f"max {e.max():.3e}; {int((e > 1e-2).sum())} of {len(e)} tensors above 1e-2")
    save("dit_autocast", **o)


# ------------------------------------------------------------------ (vi) samplers
def gen_samplers() -> None:
    o = {}
    shp = (3, 4, 8, 8)
    xt, v, nz = synth.normal("smp.xt", shp), synth.normal("smp.v", shp), synth.normal("smp.noise", shp)
    r = Euler().step(xt, v, 0.75, 0.5)
    o["euler_x_prev"], o["euler_x0"] = r["x_prev"], r["estimated_x0"]
    em = EulerMaruyama(eta=0.7)
    em.set_steps(Flow(n_steps=10).timesteps)
    torch.manual_seed(11)
    r = em.step(xt, v, 0.6, 0.5)
    for k in ("x_prev", "x_prev_mean", "x_prev_std", "estimated_x0", "logprob"):
        o["em_" + k] = r[k]
    torch.manual_seed(11)
    o["em_noise"] = torch.randn_like(xt)
    r = em.step(xt, v, 1.0, 0.9, x_prev=nz)  # t_curr > tmax branch, given x_prev
    o["em2_logprob"], o["em2_mean"] = r["logprob"], r["x_prev_mean"]
    tt = torch.tensor([0, 7, 999], dtype=torch.int32)
    for mt in ("epsilon", "xstart", "xprev"):
        for vt in ("fixed_small", "fixed_large"):
            for clamp in (False, True):
                s = DDPM(mean_type=mt, var_type=vt)
                s.set_steps(GaussianDiffusion(n_steps=1000).betas)
                torch.manual_seed(13)
                r = s.step(v, tt, xt, clamp_x=clamp)
                tag = f"ddpm_{mt}_{vt}_{int(clamp)}_"
                for k in ("x_prev", "estimated_x0", "x_prev_mean", "x_prev_std", "logprob"):
                    o[tag + k] = r[k]
    torch.manual_seed(13)
    o["ddpm_noise"] = torch.randn_like(xt)
    for eta in (0.0, 0.5):
        s = DDIM()
        s.set_steps(GaussianDiffusion(n_steps=1000).betas)
        torch.manual_seed(17)
        r = s.step(v, tt, xt, eta=eta)
        for k in r:
            o[f"ddim_eta{eta}_{k}"] = r[k]
    torch.manual_seed(17)
    o["ddim_noise"] = torch.randn_like(xt)
    # full sampler loops on the small DiT: 4-step Euler with CFG, 5-step respaced DDPM
    m = build_ref(SMALL, seed=5).eval()
    y = synth.integers("smp.y", (2,), SMALL.n_classes)
    x_init = synth.normal("smp.init", (2, 4, 8, 8))
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    out = d.generate({"x": x_init.clone(), "y": y}, use_tqdm=False, guidance_scale=2.0, return_intermediates=True)
    o["loop_euler_x"], o["loop_euler_xt"], o["loop_euler_x0"] = out["x"], out["xt"], out["estimated_x0"]
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=3, extra_args={"shift": 4.63})
    d.set_steps(3, shift=4.63)
    o["loop_euler_shift_x"] = d.generate({"x": x_init.clone(), "y": y}, use_tqdm=False)["x"]
    g = Diffuser(m, sampling_method="ddpm", model_type="gaussian_diffusion", n_steps=1000)
    g.set_steps(5)
    torch.manual_seed(19)
    with torch.no_grad():
        out = g.generate({"x": x_init.clone(), "y": y}, use_tqdm=False, guidance_scale=1.5, clamp_x=True)
    o["loop_ddpm_x"] = out["x"]
    # replay the global-generator stream of that loop: per step the p=1 forward draws rand(B) for the label
    # drop (nn.py:149) BEFORE DDPM.step draws randn_like (ddpm.py:302)
    torch.manual_seed(19)
    nz = []
    for _ in range(5):
        torch.rand(2)
        nz.append(torch.randn(2, 4, 8, 8))
    o["loop_ddpm_noise"] = torch.stack(nz)
    save("samplers", **o)



# ------------------------------------------------------------------ (vii) UNet pieces and a small UNet
UNET_SMALL = ounet.UNetConfig(image_size=(16, 16), in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1,
                              attention_resolutions=(2,), channel_mult=(1, 2), num_heads=2, use_scale_shift_norm=True,
                              resblock_updown=True, n_classes=10, classifier_free=True)


def build_unet_ref(cfg, seed: int) -> UNetModel:
    m = UNetModel(image_size=list(cfg.image_size), in_channels=cfg.in_channels, model_channels=cfg.model_channels,
                  out_channels=cfg.out_channels, num_res_blocks=cfg.num_res_blocks,
                  attention_resolutions=list(cfg.attention_resolutions), channel_mult=", ".join(map(str, cfg.channel_mult)),
                  num_heads=cfg.num_heads, use_scale_shift_norm=cfg.use_scale_shift_norm,
                  resblock_updown=cfg.resblock_updown, conv_resample=cfg.conv_resample, n_classes=cfg.n_classes,
                  classifier_free=cfg.classifier_free)
    P = synth.generic_params(ounet.param_shapes(cfg), seed=seed)
    sd = m.state_dict()
    assert set(sd) == set(P), (set(sd) ^ set(P))
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), k
    m.load_state_dict(P)
    return m


def gen_unet() -> None:
    o = {}
    te = 128
    # ResBlock variants (scale-shift norm): channel change (1x1 skip), up, down
    for tag, kw, cin, cout in (("plain", {}, 64, 96), ("up", {"up": True}, 64, 64), ("down", {"down": True}, 64, 64)):
        blk = ResBlock(cin, te, 0.0, out_channels=cout, use_scale_shift_norm=True, **kw)
        shapes = {n: tuple(p.shape) for n, p in blk.named_parameters()}
        blk.load_state_dict(synth.generic_params({f"rb_{tag}." + n: s for n, s in shapes.items()}, seed=21)
                            and {n: synth.generic_params({f"rb_{tag}." + n: shapes[n]}, seed=21)[f"rb_{tag}." + n] for n in shapes})
        x = synth.normal(f"rb_{tag}.x", (4, cin, 8, 8)).requires_grad_(True)
        emb = synth.normal(f"rb_{tag}.emb", (4, te)).requires_grad_(True)
        y = blk(x, emb)
        (y * synth.normal(f"rb_{tag}.dy", tuple(y.shape))).sum().backward()
        o[f"rb_{tag}_y"], o[f"rb_{tag}_dx"], o[f"rb_{tag}_demb"] = y, x.grad, emb.grad
        for n, p in blk.named_parameters():
            o[f"rb_{tag}_g_{n}"] = p.grad
    # AttentionBlock at 8x8 (c=128, 2 heads; the MNIST config's 512/1024-wide blocks are covered oracle-vs-HIP on the GPU)
    ab = AttentionBlock(128, num_heads=2)
    shapes = {n: tuple(p.shape) for n, p in ab.named_parameters()}
    ab.load_state_dict({n: synth.generic_params({"ab." + n: shapes[n]}, seed=22)["ab." + n] for n in shapes})
    x = synth.normal("ab.x", (4, 128, 8, 8)).requires_grad_(True)
    y = ab(x)
    (y * synth.normal("ab.dy", tuple(y.shape))).sum().backward()
    o["ab_y"], o["ab_dx"] = y, x.grad
    for n, p in ab.named_parameters():
        if p.grad is not None:
            o["ab_g_" + n] = p.grad
    # small UNet, DDPM loss, fwd + bwd
    cfg = UNET_SMALL
    m = build_unet_ref(cfg, seed=23)
    B = 4
    x0 = synth.normal("un.x0", (B, 1, 16, 16))
    noise = synth.normal("un.noise", (B, 1, 16, 16))
    yl = synth.integers("un.y", (B,), 10)
    ti = torch.tensor([3, 500, 999, 0], dtype=torch.int32)
    gd = Diffuser(m, sampling_method="ddpm", model_type="gaussian_diffusion", n_steps=1000)
    xt = gd.diffusion.add_noise(x0, ti, noise)[0]
    o["un_pred"] = m(x=xt, timesteps=ti, y=yl, p=0.0)["x"]
    loss = gd.compute_loss({"x": x0.clone(), "y": yl, "p": 0.0}, timesteps=ti, noise=noise)["loss"]
    loss.backward()
    o["un_loss"] = loss
    names, norms = [], []
    for n, p in m.named_parameters():
        if p.grad is not None:
            names.append(n)
            norms.append(p.grad.double().norm().item())
            if p.grad.numel() <= 20000:
                o["un_g_" + n] = p.grad
    o["un_grad_names"], o["un_grad_norms"] = np.array(names), np.array(norms)
    save("unet", **o)


# the UNetModel constructor DEFAULTS (additive conditioning, Downsample / Upsample modules with their 3x3 convs: unet.py:541-545)
# and the conv-free resampling variant; three levels so that two resampling stages of each kind are crossed
UNET_VARIANTS = {
    "dflt": ounet.UNetConfig(image_size=(16, 16), in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1,
                             attention_resolutions=(4,), channel_mult=(1, 2, 2), num_heads=2, use_scale_shift_norm=False,
                             resblock_updown=False, conv_resample=True, n_classes=10, classifier_free=True),
    "pool": ounet.UNetConfig(image_size=(16, 16), in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1,
                             attention_resolutions=(4,), channel_mult=(1, 2, 2), num_heads=2, use_scale_shift_norm=True,
                             resblock_updown=False, conv_resample=False, n_classes=None, classifier_free=False),
}


def gen_unet_variants() -> None:
    o = {}
    B = 4
    for tag, cfg in UNET_VARIANTS.items():
        m = build_unet_ref(cfg, seed=29)
        x0 = synth.normal(f"uv.{tag}.x0", (B, 1, 16, 16))
        noise = synth.normal(f"uv.{tag}.noise", (B, 1, 16, 16))
        yl = synth.integers(f"uv.{tag}.y", (B,), 10) if cfg.n_classes else None
        ti = torch.tensor([7, 250, 999, 0], dtype=torch.int32)
        gd = Diffuser(m, sampling_method="ddpm", model_type="gaussian_diffusion", n_steps=1000)
        xt = gd.diffusion.add_noise(x0, ti, noise)[0]
        kw = {"y": yl, "p": 0.0} if cfg.n_classes else {}
        o[f"{tag}_pred"] = m(x=xt, timesteps=ti, **kw)["x"]
        loss = gd.compute_loss({"x": x0.clone(), **kw}, timesteps=ti, noise=noise)["loss"]
        loss.backward()
        o[f"{tag}_loss"] = loss
        names, norms = [], []
        for n, p in m.named_parameters():
            if p.grad is not None:
                names.append(n)
                norms.append(p.grad.double().norm().item())
                if p.grad.numel() <= 20000:
                    o[f"{tag}_g_" + n] = p.grad
        o[f"{tag}_grad_names"], o[f"{tag}_grad_norms"] = np.array(names), np.array(norms)
    save("unet_variants", **o)


def gen_unet_full() -> None:
    """The reference UNet at configs/model/unet.yaml dims (276.7 M parameters), B = 2, DDPM epsilon loss, on the inputs of
    tests/test_full_dims_gpu.py::test_unet_at_config1_dims_against_the_oracle: (a) its fp32 prediction / loss / per-parameter
    gradient NORMS (pins the oracle at these dims: the full tensors would be 1.1 GB) and a few small gradient tensors in full;
    (b) THE bf16 YARDSTICK at these dims: the same model under ``torch.autocast("cpu", dtype=torch.bfloat16)`` -- relative L2 error
    of its prediction and of every gradient tensor against its own fp32 leg."""
    out = {}
    for B, pre in ((2, ""), (128, "b128_")):  # B = 128: the batch of configs/train_mnist_ddpm.yaml (the UNet's benched shape)
        out.update(_unet_full_legs(B, pre))
    save("unet_full", **out)


def _unet_full_legs(B: int, pre: str) -> dict:
    cfg = ounet.UNetConfig()
    x0, noise = synth.normal("fd.x0", (B, 1, 32, 32)), synth.normal("fd.noise", (B, 1, 32, 32))
    y = synth.integers("fd.y", (B,), 10)
    ti = torch.tensor([17, 940], dtype=torch.int32) if B == 2 else synth.integers("fd.t", (B,), 1000).to(torch.int32)
    legs = {}
    for leg in ("fp32", "autocast"):
        m = build_unet_ref(cfg, seed=41)
        gd = Diffuser(m, sampling_method="ddpm", model_type="gaussian_diffusion", n_steps=1000)
        xt = gd.diffusion.add_noise(x0, ti, noise)[0]
        if leg == "autocast":
            with torch.autocast("cpu", dtype=torch.bfloat16):
                pred = m(x=xt, timesteps=ti, y=y, p=0.0)["x"].detach().float()
                loss = gd.compute_loss({"x": x0.clone(), "y": y, "p": 0.0}, timesteps=ti, noise=noise)["loss"]
        else:
            pred = m(x=xt, timesteps=ti, y=y, p=0.0)["x"].detach()
            loss = gd.compute_loss({"x": x0.clone(), "y": y, "p": 0.0}, timesteps=ti, noise=noise)["loss"]
        loss.backward()
        legs[leg] = (pred, loss.detach().float(), {n: p.grad.detach().float().clone() for n, p in m.named_parameters()})
    f32, ac = legs["fp32"], legs["autocast"]
    names = list(f32[2])
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()  # noqa: E731
    o = {"pred": f32[0], "loss": f32[1], "names": np.array(names), "grad_norms": np.array([f32[2][n].double().norm().item() for n in names]),
         "ac_pred_err": np.float64(rel(ac[0], f32[0])), "ac_loss": ac[1], "ac_err": np.array([rel(ac[2][n], f32[2][n]) for n in names])}
    for n in names:
        if f32[2][n].numel() <= (4096 if B == 2 else 1024):
            o["g_" + n] = f32[2][n]
    e = o["ac_err"][o["grad_norms"] > 1e-6 * o["grad_norms"].max()]
    print(f"unet_full B={B}: loss fp32 {f32[1].item():.6f} autocast {ac[1].item():.6f}; autocast prediction error {o['ac_pred_err']:.3e}; per-tensor "
          f"gradient error median {np.median(e):.3e} max {e.max():.3e}")
    return {pre + k: v for k, v in o.items()}


# ------------------------------------------------------------------ (viii) loss curve, DiT-S/2 + AdamW
def gen_loss_curve() -> None:
    cfg = S2
    m = build_ref(cfg, seed=7)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8)
    B, steps = 4, 20  # 20 steps
    x0 = synth.normal("curve.x0", (B, 4, 32, 32))
    y = synth.integers("curve.y", (B,), 1000)
    losses = []
    for s in range(steps):
        noise = synth.normal(f"curve.noise{s}", (B, 4, 32, 32))
        t = synth.uniform(f"curve.t{s}", (B,), lo=0.02, hi=0.98)
        opt.zero_grad()
        loss = flow_loss_ref(m, x0, t, y, noise)
        loss.backward()
        opt.step()
        losses.append(loss.item())
        print(f"  step {s}: {loss.item():.6f}")
    save("loss_curve", losses=np.array(losses, dtype=np.float64),
         final_qkv0=m.layers[0].attention.qkv.weight.detach().flatten()[::577][:256])


def gen_curve_autocast() -> None:
    """The 20-step AdamW loop of `gen_loss_curve` with the imported reference under ``torch.autocast("cpu", bfloat16)`` -- the
    regime accelerate's bf16 mixed precision runs it in (trainers/common.py:103-109; loop base_trainer.py:138-151: zero_grad ->
    loss -> backward -> optimizer.step, parameters and AdamW state stay f32).  Stored: the autocast curve and its per-step relative
    error against the reference's OWN fp32 curve (`loss_curve.npz`): the bound `test_loss_curve_against_reference` holds the HIP
    bf16 regime to (no builder-chosen constant)."""
    cfg = S2
    m = build_ref(cfg, seed=7)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8)
    B, steps = 4, 20
    x0 = synth.normal("curve.x0", (B, 4, 32, 32))
    y = synth.integers("curve.y", (B,), 1000)
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "loss_curve.npz"))["losses"]
    losses = []
    for s in range(steps):
        noise = synth.normal(f"curve.noise{s}", (B, 4, 32, 32))
        t = synth.uniform(f"curve.t{s}", (B,), lo=0.02, hi=0.98)
        opt.zero_grad()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            loss = flow_loss_ref(m, x0, t, y, noise)
        loss.backward()
        opt.step()
        losses.append(loss.detach().float().item())
        print(f"  step {s}: autocast {losses[-1]:.6f}  fp32 {ref[s]:.6f}  rel {abs(losses[-1] - ref[s]) / ref[s]:.3e}")
    losses = np.array(losses, dtype=np.float64)
    err = np.abs(losses - ref) / ref
    print(f"reference under autocast vs its own fp32 curve: max {err.max():.3e} mean {err.mean():.3e}")
    save("loss_curve_autocast", losses=losses, rel_err_vs_fp32=err)


# ------------------------------------------------------------------ (ix) REPA loss hooked into a small DiT
def gen_repa() -> None:
    """the reference RepaLoss (training/losses/repa.py) with precomputed target features, hooked on layers[0] of the small DiT,
    evaluated through Flow.compute_loss(extra_losses=[...]).  Its encoder / resampler imports need torchvision (absent): those
    three modules are stubbed, the loss code itself is the reference's."""
    import importlib

    root = os.path.join(REF, "diffulab")
    rp = types.ModuleType("diffulab.networks.repa")
    rp.__path__ = [f"{root}/networks/repa"]  # type: ignore[attr-defined]
    sys.modules["diffulab.networks.repa"] = rp
    for mod, cls in (("common", "REPA"), ("dinov2", "DinoV2"), ("perceiver_resampler", "PerceiverResampler")):
        m = types.ModuleType(f"diffulab.networks.repa.{mod}")
        setattr(m, cls, type(cls, (), {}))
        sys.modules[f"diffulab.networks.repa.{mod}"] = m
    RepaLoss = importlib.import_module("diffulab.training.losses.repa").RepaLoss
    from oracle import repa as orepa

    cfg = SMALL
    m = build_ref(cfg, seed=5)
    rl = RepaLoss(repa_encoder="dinov2", alignment_layer=1, denoiser_dimension=cfg.inner_dim, hidden_dim=128, load_dino=False,
                  embedding_dim=64, coeff=0.5)
    rl.load_state_dict(synth.generic_params(orepa.param_shapes(cfg.inner_dim, 128, 64), seed=41))
    rl.set_model(m)
    B, H = 4, 16
    x0, noise = synth.normal("rp.x0", (B, 4, H, H)), synth.normal("rp.noise", (B, 4, H, H))
    y, t = synth.integers("rp.y", (B,), 10), synth.uniform("rp.t", (B,), lo=0.05, hi=0.95)
    dst = synth.normal("rp.dst", (B, (H // cfg.patch_size) ** 2, 64))
    flow = Flow(n_steps=4, sampling_method="euler")
    losses = flow.compute_loss(m, {"x": x0.clone(), "y": y, "p": 0.0}, t, noise, extra_losses=[rl], extra_args={"dst_features": dst})
    sum(losses.values()).backward()
    o = {"loss": losses["loss"], "repa": losses["RepaLoss"]}
    for n, p in rl.named_parameters():
        o["g_" + n] = p.grad
    for n, p in m.named_parameters():
        o["gd_" + n] = p.grad
    save("repa", **o)


# ------------------------------------------------------------------ (x) Perceiver resampler of the REPA config
def gen_resampler() -> None:
    """the reference PerceiverResampler (networks/repa/perceiver_resampler.py) on seeded inputs: output, input gradient and
    every parameter gradient (dim 128, depth 2, 2 heads x 64, 256 latents, 64 input tokens = an 8x8 grid)"""
    import importlib

    root = os.path.join(REF, "diffulab")
    if "diffulab.networks.repa" not in sys.modules:
        rp = types.ModuleType("diffulab.networks.repa")
        rp.__path__ = [f"{root}/networks/repa"]  # type: ignore[attr-defined]
        sys.modules["diffulab.networks.repa"] = rp
    sys.modules.pop("diffulab.networks.repa.perceiver_resampler", None)  # (gen_repa registers a stub of this name)
    PR = importlib.import_module("diffulab.networks.repa.perceiver_resampler").PerceiverResampler
    from oracle import repa as orepa

    kw = dict(dim=128, depth=2, head_dim=64, num_heads=2, ff_mult=4, num_latents=256)
    m = PR(**kw)
    shapes = orepa.resampler_param_shapes(**kw)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == shapes
    m.load_state_dict(synth.generic_params(shapes, seed=51))
    x = synth.normal("rs.x", (3, 64, 128)).requires_grad_(True)
    # As shipped, forward(x) with cos_sin=None raises IndexError: it builds [S, d/2] tables from un-batched position ids and
    # _apply_rotary (utils/nn.py:342) indexes them as [B, S, d/2].  The module's own cos_sin argument with the same helper on
    # batched ids ([1, S, 2]) is the evident intent and is what the fixture pins.
    from diffulab.networks.utils.nn import get_cos_sin_ndim_grid

    ids = torch.stack(torch.meshgrid(torch.arange(8), torch.arange(8), indexing="ij"), dim=-1).view(1, -1, 2)
    y = m(x, cos_sin=get_cos_sin_ndim_grid(ids, base=m.rope_base, axes_dim=m.rope_axes_dim))
    (y * synth.normal("rs.dy", tuple(y.shape))).sum().backward()
    o = {"y": y, "dx": x.grad}
    for n, p in m.named_parameters():
        o["g_" + n] = p.grad
    save("resampler", **o)


# ------------------------------------------------------------------ (xi) SprintDiT, simple_dit (configs/model/sprint.yaml)
SPRINT_SMALL = dict(input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4, patch_size=2,
                    encoder_depth=1, deep_layers_depth=2, decoder_depth=1, n_classes=10, classifier_free=True, drop_rate=0.75)


class _RandRecorder:
    """records every torch.rand draw of a forward (label drop nn.py:149, token scores sprint.py:343, path drop sprint.py:384)
    so that the oracle / the HIP path can be fed the same decisions"""

    def __enter__(self):
        self.draws, self._orig = [], torch.rand

        def rec(*a, **k):
            out = self._orig(*a, **k)
            self.draws.append(out.clone())
            return out

        torch.rand = rec
        return self

    def __exit__(self, *exc):
        torch.rand = self._orig


def gen_sprint() -> None:
    import importlib

    from oracle import sprint as osprint

    SprintDiT = importlib.import_module("diffulab.networks.denoisers.sprint").SprintDiT
    cfg = osprint.SprintConfig(**SPRINT_SMALL)
    m = SprintDiT(simple_dit=True, **SPRINT_SMALL)
    shapes = osprint.param_shapes(cfg)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == shapes, set(m.state_dict()) ^ set(shapes)
    P = synth.dit_params({k: v for k, v in shapes.items() if k != "mask_token"}, seed=61)
    P["mask_token"] = synth.normal("sp.mask", shapes["mask_token"]) * 0.5
    m.load_state_dict(P)
    B, H = 4, 32
    x = synth.normal("sp.x", (B, 4, H, H))
    t = synth.uniform("sp.t", (B,), lo=0.05, hi=0.95)
    y = synth.integers("sp.y", (B,), 10)
    dy = synth.normal("sp.dy", (B, 4, H, H))
    o = {}
    # (a) training step, p = 0: token drop only
    m.train()
    torch.manual_seed(3)
    with _RandRecorder() as r:
        pred = m(x=x, timesteps=t, y=y, p=0.0)["x"]
    assert len(r.draws) == 1 and r.draws[0].shape == (B, 256)
    o["a_scores"], o["a_pred"] = r.draws[0], pred
    (pred * dy).sum().backward()
    big = ("fuse.weight", "conv_proj.weight", "layers.0.attention.qkv.weight", "layers.0.modulation.lin.weight",
           "deep_layers.0.mlp_input.0.weight", "deep_layers.1.attention.proj_out.weight", "deep_layers.1.mlp_input.2.weight",
           "decoder_layers.0.attention.qkv.weight", "decoder_layers.0.mlp_input.2.weight", "last_layer.adaLN_modulation.1.weight")
    for n, p in m.named_parameters():  # every vector / small tensor and one matrix of each kind (the oracle carries the rest)
        if p.numel() <= 16384 or n in big:
            o["a_g_" + n] = p.grad.clone()
    m.zero_grad()
    # (b) training step, p = 0.5: label drop + token drop + per-sample path drop (seed chosen so both outcomes occur)
    torch.manual_seed(4)
    with _RandRecorder() as r:
        pred = m(x=x, timesteps=t, y=y, p=0.5)["x"]
    assert [tuple(d.shape) for d in r.draws] == [(B,), (B, 256), (B,)]
    assert 0 < int((r.draws[2] < 0.5).sum()) < B, r.draws[2]
    o["b_label_u"], o["b_scores"], o["b_path_u"], o["b_pred"] = r.draws[0], r.draws[1], r.draws[2], pred
    (pred * dy).sum().backward()
    for n in ("mask_token", "fuse.weight", "layers.0.attention.qkv.weight", "deep_layers.1.mlp_input.2.weight",
              "decoder_layers.0.modulation.lin.weight", "label_embed.embedding.weight"):
        o["b_g_" + n] = dict(m.named_parameters())[n].grad.clone()
    m.zero_grad()
    # (c) / (d) eval: no token drop; p = 1 skips the deep layers (the unconditional branch of classifier-free sampling)
    m.eval()
    with torch.no_grad():
        o["c_pred"] = m(x=x, timesteps=t, y=y, p=0.0)["x"]
        o["d_pred"] = m(x=x, timesteps=t, y=y, p=1.0)["x"]
    # (e) 4-step Euler sampling with guidance through the Diffuser
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    out = d.generate({"x": synth.normal("sp.init", (B, 4, H, H)), "y": y}, use_tqdm=False, guidance_scale=2.0)
    o["e_loop_x"] = out["x"]
    save("sprint", **o)


# ------------------------------------------------------------------ (xii) MMDiT joint text-image blocks (simple_dit=False)
JOINT_SMALL = dict(input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4, patch_size=2,
                   depth=2, rope_axes_dim=[16, 24, 24], rope_base=2000, classifier_free=True)


def gen_mmdit_joint() -> None:
    """MMDiT(simple_dit=False) behind the reference PrecomputedEmbedder: 64 text tokens (ragged valid lengths) + 64 image tokens"""
    import importlib
    import tempfile

    from oracle import mmdit as ommdit

    PE = importlib.import_module("diffulab.networks.embedders.precomputed").PrecomputedEmbedder
    Lc, Cd, B, H = 64, 96, 4, 16
    null = synth.normal("mj.null", (1, Lc, Cd)) * 0.5
    with tempfile.NamedTemporaryFile(suffix=".pt") as f:
        torch.save(null, f.name)
        emb = PE(f.name, null_embedding_seq_len=7)
    m = MMDiT(simple_dit=False, context_embedder=emb, **JOINT_SMALL)
    cfg = ommdit.JointConfig(context_dim=Cd, **{k: v for k, v in JOINT_SMALL.items()})
    shapes = ommdit.param_shapes(cfg)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == shapes, set(m.state_dict()) ^ set(shapes)
    m.load_state_dict(synth.dit_params(shapes, seed=71))
    x = synth.normal("mj.x", (B, 4, H, H))
    t = synth.uniform("mj.t", (B,), lo=0.05, hi=0.95)
    ctx = synth.normal("mj.ctx", (B, Lc, Cd))
    keep = torch.arange(Lc)[None, :] < torch.tensor([64, 20, 41, 5])[:, None]
    dy = synth.normal("mj.dy", (B, 4, H, H))
    o = {}
    m.train()
    pred = m(x=x, timesteps=t, initial_context={"embeddings": ctx, "attn_mask": keep}, p=0.0)["x"]
    o["a_pred"] = pred
    (pred * dy).sum().backward()
    for n, p in m.named_parameters():  # (the context branch of the LAST block feeds nothing: those gradients are None)
        if p.grad is not None:
            o["a_g_" + n] = p.grad.clone()
    m.zero_grad()
    torch.manual_seed(2)
    with _RandRecorder() as r:
        pred = m(x=x, timesteps=t, initial_context={"embeddings": ctx, "attn_mask": keep}, p=0.5)["x"]
    assert [tuple(d.shape) for d in r.draws] == [(B,)] and 0 < int((r.draws[0] < 0.5).sum()) < B, r.draws
    o["b_u"], o["b_pred"] = r.draws[0], pred
    m.eval()
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    out = d.generate({"x": synth.normal("mj.init", (B, 4, H, H)), "initial_context": {"embeddings": ctx, "attn_mask": keep}},
                     use_tqdm=False, guidance_scale=2.0)
    o["e_loop_x"] = out["x"]
    save("mmdit_joint", **o)


# ------------------------------------------------------------------ (xiii) SprintDiT joint form with single-stream deep blocks
SPRINT_JOINT = dict(input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4, patch_size=1,
                    encoder_depth=1, deep_layers_depth=3, n_single_stream_blocks=2, decoder_depth=2, rope_axes_dim=[16, 24, 24],
                    rope_base=2000, classifier_free=True, drop_rate=0.75)


def gen_sprint_joint() -> None:
    """SprintDiT(simple_dit=False): joint encoder block, deep stage = one joint + two single-stream blocks on 64 of 256 image
    tokens, fuse / fuse_context, two joint decoder blocks; 64 text tokens with a ragged mask (the shape of
    configs/train_imagenet_repa_txt_to_img_sprint.yaml at small width)"""
    import importlib
    import tempfile

    from oracle import sprint as osprint

    PE = importlib.import_module("diffulab.networks.embedders.precomputed").PrecomputedEmbedder
    SprintDiT = importlib.import_module("diffulab.networks.denoisers.sprint").SprintDiT
    Lc, Cd, B, H = 64, 96, 4, 16
    null = synth.normal("sj.null", (1, Lc, Cd)) * 0.5
    with tempfile.NamedTemporaryFile(suffix=".pt") as f:
        torch.save(null, f.name)
        emb = PE(f.name, null_embedding_seq_len=7)
    m = SprintDiT(simple_dit=False, context_embedder=emb, **SPRINT_JOINT)
    cfg = osprint.SprintJointConfig(context_dim=Cd, **SPRINT_JOINT)
    shapes = osprint.joint_param_shapes(cfg)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == shapes, set(m.state_dict()) ^ set(shapes)
    P = synth.dit_params({k: v for k, v in shapes.items() if k != "mask_token"}, seed=81)
    P["mask_token"] = synth.normal("sj.mask", shapes["mask_token"]) * 0.5
    m.load_state_dict(P)
    x = synth.normal("sj.x", (B, 4, H, H))
    t = synth.uniform("sj.t", (B,), lo=0.05, hi=0.95)
    ctx = synth.normal("sj.ctx", (B, Lc, Cd))
    keep = torch.arange(Lc)[None, :] < torch.tensor([64, 20, 41, 5])[:, None]
    dy = synth.normal("sj.dy", (B, 4, H, H))
    ic = {"embeddings": ctx, "attn_mask": keep}
    o = {}
    m.train()
    torch.manual_seed(5)
    with _RandRecorder() as r:
        pred = m(x=x, timesteps=t, initial_context=ic, p=0.0)["x"]
    assert [tuple(d.shape) for d in r.draws] == [(B,), (B, 256)], [tuple(d.shape) for d in r.draws]  # embedder draw (p = 0), scores
    o["a_scores"], o["a_pred"] = r.draws[1], pred
    (pred * dy).sum().backward()
    big = ("fuse.weight", "fuse_context.weight", "context_embed.weight", "conv_proj.weight", "layers.0.attention.qkv_context.weight",
           "layers.0.mlp_input.2.weight", "deep_layers.0.attention.input_proj_out.weight", "deep_layers.0.mlp_context.0.weight",
           "deep_layers.1.attention.qkv.weight", "deep_layers.1.mlp.0.weight", "deep_layers.2.attention.proj_out.weight",
           "deep_layers.2.mlp.2.weight", "deep_layers.2.modulation.1.weight", "decoder_layers.0.mlp_context.2.weight",
           "decoder_layers.1.attention.qkv_input.weight", "decoder_layers.1.modulation_context.lin.weight")
    for n, p in m.named_parameters():  # every vector / small tensor and one matrix of each kind (the oracle carries the rest)
        if p.grad is not None and (p.numel() <= 16384 or n in big):
            o["a_g_" + n] = p.grad.clone()
    o["a_none"] = np.array(sorted(n for n, p in m.named_parameters() if p.grad is None))
    m.zero_grad()
    for seed in range(6, 64):  # first seed whose draws drop some (not all) contexts and some (not all) deep paths
        torch.manual_seed(seed)
        with _RandRecorder() as r, torch.no_grad():
            pred = m(x=x, timesteps=t, initial_context=ic, p=0.5)["x"]
        assert [tuple(d.shape) for d in r.draws] == [(B,), (B, 256), (B,)]
        if 0 < int((r.draws[0] < 0.5).sum()) < B and 0 < int((r.draws[2] < 0.5).sum()) < B:
            break
    else:
        raise AssertionError("no seed with mixed drops")
    o["b_ctx_u"], o["b_scores"], o["b_path_u"], o["b_pred"] = r.draws[0], r.draws[1], r.draws[2], pred
    m.eval()
    with torch.no_grad():
        o["c_pred"] = m(x=x, timesteps=t, initial_context=ic, p=0.0)["x"]
        o["d_pred"] = m(x=x, timesteps=t, initial_context=ic, p=1.0)["x"]
    save("sprint_joint", **o)


# ------------------------------------------------------------------ DDT / SPRINT configurations at their own yaml dims
def gen_yaml_dims() -> None:
    """The reference's DDT and SprintDiT at the dims of this repo's configs/model/{ddt,sprint,ddt_txt,sprint_txt}.yaml (mirrors of the
    reference's own model nodes) on the inputs of tests/test_full_dims_gpu.py: (a) the fp32 prediction, every gradient norm and the
    recorded token scores of the SPRINT forms (pins the oracle at these dims); (b) the bf16 yardstick: the same step under
    ``torch.autocast("cpu", dtype=torch.bfloat16)`` -- relative L2 error of the prediction and of every gradient tensor against the
    model's own fp32 leg."""
    import importlib
    import tempfile

    import yaml

    PE = importlib.import_module("diffulab.networks.embedders.precomputed").PrecomputedEmbedder
    DDT = importlib.import_module("diffulab.networks.denoisers.ddt").DDT
    SprintDiT = importlib.import_module("diffulab.networks.denoisers.sprint").SprintDiT
    from oracle import ddt as oddt
    from oracle import sprint as osprint

    def node(name):
        with open(os.path.join(HERE, "..", "..", "configs", "model", name + ".yaml")) as f:
            kw = yaml.safe_load(f)
        kw.pop("_target_")
        return kw

    def okw(kw):
        return {k: v for k, v in kw.items() if k not in ("simple_dit", "simple_ddt", "use_checkpoint")}

    Lc, Dc = 128, 1024  # embedder node of configs/train_imagenet_repa_txt_to_img*.yaml

    def embedder(tag):
        with tempfile.NamedTemporaryFile(suffix=".pt") as f:
            torch.save(synth.normal(f"{tag}.null", (1, Lc, Dc)) * 0.5, f.name)
            return PE(f.name, null_embedding_seq_len=7)

    o = {}
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()  # noqa: E731
    for tag in ("ddt", "sprint", "ddt_txt", "sprint_txt"):
        kw = node(tag)
        joint = tag.endswith("_txt")
        legs = {}
        for leg in ("fp32", "autocast"):
            if tag == "ddt":
                m, shapes = DDT(**kw), oddt.param_shapes(oddt.DDTConfig(**okw(kw)))
                P = synth.dit_params(shapes, seed=111)
            elif tag == "ddt_txt":
                m, shapes = DDT(context_embedder=embedder("ft"), **kw), oddt.joint_param_shapes(oddt.DDTJointConfig(context_dim=Dc, **okw(kw)))
                P = synth.dit_params(shapes, seed=117)
            else:
                if joint:
                    m, shapes = SprintDiT(context_embedder=embedder("f5"), **kw), osprint.joint_param_shapes(osprint.SprintJointConfig(context_dim=Dc, **okw(kw)))
                else:
                    m, shapes = SprintDiT(**kw), osprint.param_shapes(osprint.SprintConfig(**okw(kw)))
                P = synth.dit_params({k: v for k, v in shapes.items() if k != "mask_token"}, seed=91 if joint else 113)
                P["mask_token"] = synth.normal("f5.mask" if joint else "fc.mask", shapes["mask_token"]) * 0.5
            assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == shapes, set(m.state_dict()) ^ set(shapes)
            m.load_state_dict(P)
            m.train()
            if joint:
                pre, B = ("ft", 2) if tag == "ddt_txt" else ("f5", 2)
                x, dy = synth.normal(f"{pre}.x", (B, 128, 32, 32)), synth.normal(f"{pre}.dy", (B, 128, 32, 32))
                t = torch.tensor([0.27, 0.88] if tag == "ddt_txt" else [0.31, 0.83])
                keep = torch.arange(Lc)[None, :] < torch.tensor([51, Lc] if tag == "ddt_txt" else [Lc, 37])[:, None]
                args = dict(x=x, timesteps=t, initial_context={"embeddings": synth.normal(f"{pre}.ctx", (B, Lc, Dc)) * 0.5, "attn_mask": keep}, p=0.0)
            else:
                B = 4
                x, dy = synth.normal("fc.x", (B, 3, 32, 32)), synth.normal("fc.dy", (B, 3, 32, 32))
                args = dict(x=x, timesteps=synth.uniform("fc.t", (B,), lo=0.05, hi=0.95), y=synth.integers("fc.y", (B,), 10), p=0.0)
            torch.manual_seed(5)  # both legs draw the same token scores
            with _RandRecorder() as r:
                if leg == "autocast":
                    with torch.autocast("cpu", dtype=torch.bfloat16):
                        pred = m(**args)["x"]
                        loss = (pred.float() * dy).sum()
                else:
                    pred = m(**args)["x"]
                    loss = (pred * dy).sum()
            loss.backward()
            scores = [d for d in r.draws if d.dim() == 2]
            legs[leg] = (pred.detach().float(), {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None},
                         scores[0] if scores else None)
        f32, ac = legs["fp32"], legs["autocast"]
        names = list(f32[1])
        assert f32[2] is None or torch.equal(f32[2], ac[2])
        o[f"{tag}_pred"] = f32[0]
        if f32[2] is not None:
            o[f"{tag}_scores"] = f32[2]
        o[f"{tag}_names"] = np.array(names)
        o[f"{tag}_grad_norms"] = np.array([f32[1][n].double().norm().item() for n in names])
        o[f"{tag}_ac_pred_err"] = np.float64(rel(ac[0], f32[0]))
        o[f"{tag}_ac_err"] = np.array([rel(ac[1][n], f32[1][n]) for n in names])
        for n in names:  # the small tensors in full (every norm scale / bias)
            if f32[1][n].numel() <= 1024:
                o[f"{tag}_g_{n}"] = f32[1][n]
        e = o[f"{tag}_ac_err"]
        print(f"{tag}: autocast prediction error {o[tag + '_ac_pred_err']:.3e}; per-tensor gradient error median {np.median(e):.3e} max {e.max():.3e}")
    save("yaml_dims", **o)


# ------------------------------------------------------------------ (xiv) DDT, simple_ddt (configs/model/ddt.yaml)
DDT_SMALL = dict(input_channels=4, output_channels=4, inner_dim=128, num_heads=2, mlp_ratio=4, patch_size=2, encoder_depth=2,
                 decoder_depth=2, n_classes=10, classifier_free=True)


def gen_ddt() -> None:
    import importlib

    from oracle import ddt as oddt

    DDT = importlib.import_module("diffulab.networks.denoisers.ddt").DDT
    cfg = oddt.DDTConfig(**DDT_SMALL)
    m = DDT(simple_ddt=True, **DDT_SMALL)
    shapes = oddt.param_shapes(cfg)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == shapes, set(m.state_dict()) ^ set(shapes)
    m.load_state_dict(synth.dit_params(shapes, seed=91))
    B, H = 4, 16
    x = synth.normal("dd.x", (B, 4, H, H))
    t = synth.uniform("dd.t", (B,), lo=0.05, hi=0.95)
    y = synth.integers("dd.y", (B,), 10)
    dy = synth.normal("dd.dy", (B, 4, H, H))
    o = {}
    m.train()
    pred = m(x=x, timesteps=t, y=y, p=0.0)["x"]
    o["pred"] = pred
    (pred * dy).sum().backward()
    big = ("conv_proj_encoder.weight", "conv_proj_decoder.weight", "layers.0.attention.qkv.weight", "layers.1.mlp_input.0.weight",
           "layers.1.modulation.lin.weight", "decoder_layers.0.modulation.lin.weight", "decoder_layers.0.attention.proj_out.weight",
           "decoder_layers.1.attention.qkv.weight", "decoder_layers.1.mlp_input.2.weight", "last_layer.adaLN_modulation.1.weight",
           "time_embed.2.weight")
    for n, p in m.named_parameters():  # every vector / small tensor and one matrix of each kind (the oracle carries the rest)
        if p.numel() <= 16384 or n in big:
            o["g_" + n] = p.grad.clone()
    m.eval()
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    out = d.generate({"x": synth.normal("dd.init", (B, 4, H, H)), "y": y}, use_tqdm=False, guidance_scale=2.0)
    o["loop_x"] = out["x"]
    save("ddt", **o)


# ------------------------------------------------------------------ (xv) DDT with the joint text-image encoder
DDT_JOINT = dict(input_channels=4, output_channels=4, inner_dim=128, num_heads=2, mlp_ratio=4, patch_size=2, encoder_depth=2,
                 decoder_depth=2, rope_axes_dim=[16, 24, 24], rope_base=1000, classifier_free=True)


def gen_ddt_joint() -> None:
    """DDT(simple_ddt=False) behind the reference PrecomputedEmbedder (configs/train_imagenet_repa_txt_to_img.yaml at small width)"""
    import importlib
    import tempfile

    from oracle import ddt as oddt

    PE = importlib.import_module("diffulab.networks.embedders.precomputed").PrecomputedEmbedder
    DDT = importlib.import_module("diffulab.networks.denoisers.ddt").DDT
    Lc, Cd, B, H = 64, 96, 4, 16
    null = synth.normal("dj.null", (1, Lc, Cd)) * 0.5
    with tempfile.NamedTemporaryFile(suffix=".pt") as f:
        torch.save(null, f.name)
        emb = PE(f.name, null_embedding_seq_len=7)
    m = DDT(simple_ddt=False, context_embedder=emb, **DDT_JOINT)
    cfg = oddt.DDTJointConfig(context_dim=Cd, **DDT_JOINT)
    shapes = oddt.joint_param_shapes(cfg)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == shapes, set(m.state_dict()) ^ set(shapes)
    m.load_state_dict(synth.dit_params(shapes, seed=101))
    x = synth.normal("dj.x", (B, 4, H, H))
    t = synth.uniform("dj.t", (B,), lo=0.05, hi=0.95)
    ctx = synth.normal("dj.ctx", (B, Lc, Cd))
    keep = torch.arange(Lc)[None, :] < torch.tensor([64, 20, 41, 5])[:, None]
    dy = synth.normal("dj.dy", (B, 4, H, H))
    ic = {"embeddings": ctx, "attn_mask": keep}
    o = {}
    m.train()
    pred = m(x=x, timesteps=t, initial_context=ic, p=0.0)["x"]
    o["pred"] = pred
    (pred * dy).sum().backward()
    big = ("context_embed.weight", "conv_proj_decoder.weight", "layers.0.attention.qkv_context.weight", "layers.1.mlp_input.0.weight",
           "decoder_layers.0.modulation.lin.weight", "decoder_layers.1.attention.qkv.weight", "decoder_layers.1.mlp_input.2.weight",
           "last_layer.adaLN_modulation.1.weight")
    for n, p in m.named_parameters():
        if p.grad is not None and (p.numel() <= 16384 or n in big):
            o["g_" + n] = p.grad.clone()
    o["none"] = np.array(sorted(n for n, p in m.named_parameters() if p.grad is None))
    torch.manual_seed(2)
    with _RandRecorder() as r, torch.no_grad():
        o["b_pred"] = m(x=x, timesteps=t, initial_context=ic, p=0.5)["x"]
    assert 0 < int((r.draws[0] < 0.5).sum()) < B
    o["b_u"] = r.draws[0]
    save("ddt_joint", **o)


# ------------------------------------------------------------------ (xvi) MMDiT with single-stream blocks at the end of the stack
def gen_mmdit_single() -> None:
    import importlib
    import tempfile

    from oracle import mmdit as ommdit

    PE = importlib.import_module("diffulab.networks.embedders.precomputed").PrecomputedEmbedder
    Lc, Cd, B, H = 64, 96, 4, 16
    null = synth.normal("ms.null", (1, Lc, Cd)) * 0.5
    with tempfile.NamedTemporaryFile(suffix=".pt") as f:
        torch.save(null, f.name)
        emb = PE(f.name, null_embedding_seq_len=7)
    kw = dict(JOINT_SMALL, depth=3, n_single_stream_blocks=2)
    m = MMDiT(simple_dit=False, context_embedder=emb, **kw)
    cfg = ommdit.JointConfig(context_dim=Cd, **kw)
    shapes = ommdit.param_shapes(cfg)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == shapes, set(m.state_dict()) ^ set(shapes)
    m.load_state_dict(synth.dit_params(shapes, seed=111))
    x = synth.normal("ms.x", (B, 4, H, H))
    t = synth.uniform("ms.t", (B,), lo=0.05, hi=0.95)
    ctx = synth.normal("ms.ctx", (B, Lc, Cd))
    keep = torch.arange(Lc)[None, :] < torch.tensor([64, 20, 41, 5])[:, None]
    dy = synth.normal("ms.dy", (B, 4, H, H))
    o = {}
    m.train()
    pred = m(x=x, timesteps=t, initial_context={"embeddings": ctx, "attn_mask": keep}, p=0.0)["x"]
    o["pred"] = pred
    (pred * dy).sum().backward()
    big = ("context_embed.weight", "layers.0.attention.qkv_context.weight", "layers.0.mlp_input.0.weight",
           "layers.1.attention.qkv.weight", "layers.1.mlp.2.weight", "layers.1.modulation.1.weight",
           "layers.2.attention.proj_out.weight", "layers.2.mlp.0.weight")
    for n, p in m.named_parameters():
        if p.grad is not None and (p.numel() <= 16384 or n in big):
            o["g_" + n] = p.grad.clone()
    o["none"] = np.array(sorted(n for n, p in m.named_parameters() if p.grad is None))
    save("mmdit_single", **o)


def gen_datasets() -> None:
    """the reference's MNIST idx / CIFAR-10 pickle readers (datasets/mnist.py, datasets/cifar10.py) on small synthetic files written
    by oracle.synth.write_fake_mnist / write_fake_cifar10 (seeded bytes in the real on-disk formats): items 0, 3 and the last"""
    import importlib
    import tempfile

    dn = types.ModuleType("diffulab.datasets")
    dn.__path__ = [os.path.join(REF, "diffulab", "datasets")]
    sys.modules["diffulab.datasets"] = dn
    mn = importlib.import_module("diffulab.datasets.mnist")
    cf = importlib.import_module("diffulab.datasets.cifar10")
    o = {}
    with tempfile.TemporaryDirectory() as td:
        synth.write_fake_mnist(td, n_train=12, n_test=5, seed=11)
        synth.write_fake_cifar10(td, batches={"data_batch_1": 6, "data_batch_2": 4}, seed=12)
        for tag, ds in (("mnist_train", mn.MNISTDataset(td, train=True)), ("mnist_test", mn.MNISTDataset(td, train=False)),
                        ("cifar", cf.CIFAR10Dataset(td, batches_to_load=["data_batch_1", "data_batch_2"]))):
            o[f"{tag}_len"] = len(ds)
            for i in (0, 3, len(ds) - 1):
                it = ds[i]["model_inputs"]
                o[f"{tag}_x{i}"], o[f"{tag}_y{i}"] = it["x"], it["y"]
    save("datasets", **o)


def gen_multiar() -> None:
    """the reference's MultiARBatchSampler and collate_fn (datasets/imagenet.py:177-236) -- integer / index work on Python's
    `random`: the epoch's batch lists for four (shuffle, drop_last) settings after random.seed(1234), two consecutive epochs of the
    shuffling ones, __len__, and the collated tensors of one mixed batch.  `streaming` and `torchvision` are absent here: stubbed,
    the two objects under test never touch them."""
    import importlib
    import random

    st = types.ModuleType("streaming")
    st.StreamingDataset = object
    sys.modules.setdefault("streaming", st)
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvt.ToTensor = lambda: (lambda im: im)
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    dn = types.ModuleType("diffulab.datasets")
    dn.__path__ = [os.path.join(REF, "diffulab", "datasets")]
    sys.modules["diffulab.datasets"] = dn
    im = importlib.import_module("diffulab.datasets.imagenet")

    class DS:
        buckets = synth.multiar_buckets()

    o = {}
    for shuffle in (True, False):
        for drop_last in (True, False):
            tag = f"s{int(shuffle)}d{int(drop_last)}"
            smp = im.MultiARBatchSampler(DS(), batch_size=4, shuffle=shuffle, drop_last=drop_last)
            random.seed(1234)
            for ep in range(2):
                batches = list(smp)
                o[f"{tag}_e{ep}_flat"] = np.array([i for b in batches for i in b], np.int64)
                o[f"{tag}_e{ep}_lens"] = np.array([len(b) for b in batches], np.int64)
            o[f"{tag}_len"] = len(smp)
    g = torch.Generator().manual_seed(5)
    items = [{"model_inputs": {"x": torch.randn(4, 6, 10, generator=g), "initial_context": f"caption {i}"},
              "extra": {"dst_features": torch.randn(7, 12, generator=g)} if i != 1 else {}} for i in range(3)]
    del items[2]["model_inputs"]["initial_context"]  # (a missing caption collates to "")
    c = im.collate_fn(items)
    o["col_x"], o["col_feats"] = c["model_inputs"]["x"], c["extra"]["dst_features"]
    o["col_ctx"] = np.array(c["model_inputs"]["initial_context"])
    save("multiar", **o)


if __name__ == "__main__":
    which = sys.argv[1:] or ["unet_full", "yaml_dims", "datasets", "multiar", "autocast", "schedules", "prims", "block", "small", "small16", "s2", "samplers", "unet", "unet_variants", "curve", "curve_autocast", "repa", "resampler", "sprint", "mmdit_joint", "sprint_joint", "ddt", "ddt_joint", "mmdit_single"]
    fns = {"unet_full": gen_unet_full, "yaml_dims": gen_yaml_dims, "datasets": gen_datasets, "multiar": gen_multiar, "autocast": gen_autocast, "repa": gen_repa, "resampler": gen_resampler, "sprint": gen_sprint, "mmdit_joint": gen_mmdit_joint, "sprint_joint": gen_sprint_joint, "ddt": gen_ddt, "ddt_joint": gen_ddt_joint, "mmdit_single": gen_mmdit_single, "schedules": gen_schedules, "prims": gen_prims, "block": gen_block, "small": gen_small_model, "small16": gen_small16,
           "s2": gen_s2_model, "samplers": gen_samplers, "curve": gen_loss_curve, "curve_autocast": gen_curve_autocast, "unet": gen_unet, "unet_variants": gen_unet_variants}
    for w in which:
        print("==", w)
        fns[w]()
