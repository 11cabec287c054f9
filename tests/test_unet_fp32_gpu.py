"""The fp32-class regime of the UNet (`precision_type="no"`: the reference's default, which BASELINE configuration 1 --
train_mnist_ddpm.yaml -- inherits): the f32 kernels of csrc/f32.hip (3x3 convolution as im2col + exact-f32 MFMA GEMM with its two
gradients, GroupNorm32 / FiLM, 2x2 resampling, the AttentionBlock over the strided batched GEMM) and `unet_engine_f32.UNetEngineF32`
against the committed outputs of the imported reference (tests/golden/unet.npz, unet_variants.npz) and torch fp64 autograd.

Bar: per-tensor relative L2 <= 1e-5 ("HIP fp32-mode kernels <= 1e-5"), against 2e-2 .. 8e-2 in the bf16 regime.
"""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import diffusion as od  # noqa: E402
from oracle import synth  # noqa: E402
from oracle import unet as ounet  # noqa: E402

DEV = "cuda"
TOL = 1e-5


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def nhwc(x):  # [B,C,H,W] cpu -> [B*H*W, C] f32 cuda
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().float().to(DEV)


def nchw(t, B, H, W):
    return t.double().cpu().reshape(B, H, W, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("B,H,W,ci,co", [(2, 8, 8, 16, 24), (2, 16, 16, 1, 32), (3, 4, 4, 48, 7), (2, 6, 10, 5, 3)])
def test_conv3x3_f32_forward_and_both_gradients(B, H, W, ci, co):
    """nn.Conv2d(3x3, padding=1) as dl_f32_im2col3x3 + dl_f32_gemm on the weight's native layout; data gradient through
    dl_f32_col2im3x3; weight gradient accumulated: vs F.conv2d under fp64 autograd (non-square images, 1-channel stem, odd widths)"""
    from diffulab_amd import ops

    g = torch.Generator().manual_seed(B * H + ci * co)
    x, w, b = torch.randn(B, ci, H, W, generator=g), torch.randn(co, ci, 3, 3, generator=g) * 0.2, torch.randn(co, generator=g)
    dy = torch.randn(B, co, H, W, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y = F.conv2d(xr, wr, b.double(), padding=1)
    (y * dy.double()).sum().backward()
    M = B * H * W
    xd, wd, bd, dyd = nhwc(x), w.to(DEV), b.to(DEV), nhwc(dy)
    cols, out = torch.empty(M, 9 * ci, device=DEV), torch.empty(M, co, device=DEV)
    ops.f32_im2col3x3(xd, cols, B, H, W, ci)
    ops.f32_linear(cols, wd.view(co, 9 * ci), out, bias=bd)
    assert rel(nchw(out, B, H, W), y) < 2e-6
    gw = torch.ones(co, 9 * ci, device=DEV)
    ops.f32_gemm(dyd, cols, gw, co, 9 * ci, M, lda=co, ldb=9 * ci, ldc=9 * ci, ta=True, tb=True, accumulate=True)
    assert rel(gw.view(co, ci, 3, 3) - 1.0, wr.grad) < 2e-6
    dcols, dx = torch.empty(M, 9 * ci, device=DEV), torch.empty(M, ci, device=DEV)
    ops.f32_gemm(dyd, wd.view(co, 9 * ci), dcols, M, 9 * ci, co, lda=co, ldb=9 * ci, ldc=9 * ci, tb=True)
    ops.f32_col2im3x3(dcols, dx, B, H, W, ci)
    assert rel(nchw(dx, B, H, W), xr.grad) < 2e-6


@pytest.mark.parametrize("C,film,silu,H", [(64, True, True, 8), (32, False, True, 8), (96, True, False, 4), (384, False, True, 4),
                                           (1536, True, True, 2), (128, True, True, 16)])
def test_groupnorm32_f32_forward_backward(C, film, silu, H):
    from diffulab_amd import ops

    B, W = 3, H
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, C, H, W, generator=g) * 2 + 0.5
    w, b = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    fs = 0.3 * torch.randn(B, 2 * C, generator=g)
    dy, dres = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    xr, wr, br, fr = (t.double().requires_grad_(True) for t in (x, w, b, fs))
    y = ounet.group_norm32(xr, wr, br)
    if film:
        y = y * (1 + fr[:, :C, None, None]) + fr[:, C:, None, None]
    if silu:
        y = F.silu(y)
    ((y * dy.double()).sum() + (xr * dres.double()).sum()).backward()
    HW = H * W
    xd, wd, bd, fd = nhwc(x), w.to(DEV), b.to(DEV), fs.to(DEV)
    stats, out = torch.empty(B, 32, 2, device=DEV), torch.empty(B * HW, C, device=DEV)
    f_s, f_h = (fd[:, :C], fd[:, C:]) if film else (None, None)
    ops.f32_gn_stats(xd, stats, B, HW, C)
    ops.f32_gn_apply_fwd(xd, stats, wd, bd, f_s, f_h, silu, out, B, HW, C)
    assert rel(nchw(out, B, H, W), y) < 2e-6
    dx, part, dfd = torch.empty(B * HW, C, device=DEV), torch.empty(2, B, C, device=DEV), torch.zeros(B, 2 * C, device=DEV)
    ops.f32_gn_bwd(nhwc(dy), xd, stats, wd, bd, f_s, f_h, silu, nhwc(dres), dx, part[0], part[1], dfd[:, :C] if film else None,
                   dfd[:, C:] if film else None, B, HW, C)
    assert rel(nchw(dx, B, H, W), xr.grad) < 5e-6
    assert rel(part[0].sum(0), wr.grad) < 5e-6 and rel(part[1].sum(0), br.grad) < 5e-6
    if film:
        assert rel(dfd, fr.grad) < 5e-6


def test_resampling_layout_and_attention_f32():
    from diffulab_amd import ops
    from diffulab_amd.unet_engine_f32 import _F32Ops

    g = torch.Generator().manual_seed(5)
    B, C, H, W = 3, 20, 6, 10
    x = torch.randn(B, C, H, W, generator=g)
    t = torch.empty(B * H * W, C, device=DEV)
    ops.f32_nchw_to_nhwc(x.to(DEV), t, B, C, H * W)
    assert torch.equal(t.cpu(), nhwc(x).cpu())
    back = torch.empty(B, C, H, W, device=DEV)
    ops.f32_nhwc_to_nchw(t, back, B, C, H * W)
    assert torch.equal(back.cpu(), x)
    small = torch.empty(B * (H // 2) * (W // 2), C, device=DEV)
    _F32Ops.reduce2x2(t, small, B, H // 2, W // 2, C, 0.25)
    assert rel(nchw(small, B, H // 2, W // 2), F.avg_pool2d(x.double(), 2)) < 1e-6
    _F32Ops.pick2x2(t, small, B, H // 2, W // 2, C)
    assert torch.equal(nchw(small, B, H // 2, W // 2).float(), x[:, :, ::2, ::2])
    big = torch.empty(B * 4 * H * W, C, device=DEV)
    _F32Ops.expand2x2(t, big, B, H, W, C, 1.0)
    assert torch.equal(nchw(big, B, 2 * H, 2 * W).float(), F.interpolate(x, scale_factor=2, mode="nearest"))
    _F32Ops.stuff2x2(t, big, B, H, W, C)
    z = torch.zeros(B, C, 2 * H, 2 * W)
    z[:, :, ::2, ::2] = x
    assert torch.equal(nchw(big, B, 2 * H, 2 * W).float(), z)
    # additive conditioning + its pixel sum
    e = torch.randn(B, 64, generator=g).to(DEV)
    o = torch.empty_like(t)
    ops.f32_rowbias_add(t, e[:, 8 : 8 + C], o, B, H * W, C)
    assert rel(nchw(o, B, H, W), x.double() + e[:, 8 : 8 + C].double().cpu()[:, :, None, None]) < 1e-6
    de = torch.zeros(B, 64, device=DEV)
    ops.f32_rowbias_bwd(t, de[:, 8 : 8 + C], B, H * W, C)
    assert rel(de[:, 8 : 8 + C], x.double().sum((2, 3))) < 1e-6 and float(de[:, :8].abs().sum()) == 0.0
    # AttentionBlock core (heads = dh-wide column blocks; k, v are the halves of one kv matrix)
    Bn, n, nh, dh = 2, 16, 2, 12
    c = nh * dh
    q, kv, dout = torch.randn(Bn * n, c, generator=g), torch.randn(Bn * n, 2 * c, generator=g), torch.randn(Bn * n, c, generator=g)
    qr, kvr = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    hd = lambda z: z.reshape(Bn, n, nh, dh).transpose(1, 2)  # noqa: E731
    att = torch.softmax(hd(qr) @ hd(kvr[:, :c]).transpose(-1, -2) * dh**-0.5, -1) @ hd(kvr[:, c:])
    ref = att.transpose(1, 2).reshape(Bn * n, c)
    (ref * dout.double()).sum().backward()
    qd, kvd = q.to(DEV), kv.to(DEV)
    out, probs = torch.empty(Bn * n, c, device=DEV), torch.empty(Bn, nh, n, n, device=DEV)
    _F32Ops.attn_small_fwd(qd, kvd[:, :c], kvd[:, c:], out, probs, Bn, n, nh, dh)
    assert rel(out, ref) < 2e-6

    class _E:  # (scratch provider of the backward)
        def _scr(self, key, numel, dtype):
            return torch.empty(numel, device=DEV)

    dq, dkv = torch.empty(Bn * n, c, device=DEV), torch.empty(Bn * n, 2 * c, device=DEV)
    _F32Ops(_E()).attn_small_bwd(qd, kvd[:, :c], kvd[:, c:], dout.to(DEV), probs, dq, dkv[:, :c], dkv[:, c:], Bn, n, nh, dh)
    assert rel(dq, qr.grad) < 5e-6 and rel(dkv, kvr.grad) < 5e-6


UNET_VARIANTS = {  # tests/golden/make_golden.py::UNET_VARIANTS (+ the first fixture's configuration)
    "un": dict(image_size=(16, 16), in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1, attention_resolutions=(2,),
               channel_mult=(1, 2), num_heads=2, use_scale_shift_norm=True, resblock_updown=True, n_classes=10, classifier_free=True),
    "dflt": dict(image_size=(16, 16), in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1, attention_resolutions=(4,),
                 channel_mult=(1, 2, 2), num_heads=2, use_scale_shift_norm=False, resblock_updown=False, conv_resample=True,
                 n_classes=10, classifier_free=True),
    "pool": dict(image_size=(16, 16), in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1, attention_resolutions=(4,),
                 channel_mult=(1, 2, 2), num_heads=2, use_scale_shift_norm=True, resblock_updown=False, conv_resample=False,
                 n_classes=None, classifier_free=False),
}


@pytest.mark.parametrize("tag", ["un", "dflt", "pool"])
def test_unet_fp32_against_reference_fixtures(golden, tag):
    """UNetModel in the fp32 regime under the DDPM epsilon loss -- ResBlock resampling + FiLM + attention ("un"), the constructor
    defaults with additive conditioning and Downsample / Upsample convolutions ("dflt"), the conv-free pooling variant without labels
    ("pool") -- prediction, loss and EVERY parameter gradient against the REFERENCE's fp32 outputs; identical bits on a second run"""
    from diffulab_amd import Diffuser
    from diffulab_amd.networks.denoisers import UNetModel

    g = golden("unet" if tag == "un" else "unet_variants")
    kw = UNET_VARIANTS[tag]
    cfg = ounet.UNetConfig(**kw)
    mk = dict(kw, image_size=list(kw["image_size"]), attention_resolutions=list(kw["attention_resolutions"]),
              channel_mult=", ".join(map(str, kw["channel_mult"])))
    m = UNetModel(**mk)
    m.load_state_dict(synth.generic_params(ounet.param_shapes(cfg), seed=23 if tag == "un" else 29))
    m = m.set_precision("fp32").to(DEV)
    B = 4
    pre = "un" if tag == "un" else f"uv.{tag}"
    x0, noise = synth.normal(f"{pre}.x0", (B, 1, 16, 16)), synth.normal(f"{pre}.noise", (B, 1, 16, 16))
    yl = synth.integers(f"{pre}.y", (B,), 10) if cfg.n_classes else None
    ti = torch.tensor([3, 500, 999, 0] if tag == "un" else [7, 250, 999, 0], dtype=torch.int32)
    xt = od.ddpm_add_noise(od.GaussianTables(1000), x0, ti, noise)
    cond = {"y": yl.to(DEV), "p": 0.0} if yl is not None else {}
    with torch.no_grad():
        pred = m(x=xt.to(DEV), timesteps=ti.to(DEV), **cond)["x"]
    assert type(m.engine).__name__ == "UNetEngineF32"
    assert rel(pred, g[f"{tag}_pred"]) < TOL
    gd = Diffuser(m, sampling_method="ddpm", model_type="gaussian_diffusion", n_steps=1000)
    loss = gd.compute_loss({"x": x0.to(DEV), **cond}, timesteps=ti.to(DEV), noise=noise.to(DEV))["loss"]
    loss.backward()
    assert abs(loss.item() - float(g[f"{tag}_loss"])) / float(g[f"{tag}_loss"]) < TOL
    norms = dict(zip(g[f"{tag}_grad_names"].tolist(), g[f"{tag}_grad_norms"].tolist()))
    floor = 1e-6 * max(norms.values())  # conv biases in front of a GroupNorm have an exactly-zero gradient: f32 round-off here
    named = dict(m.named_parameters())
    assert set(named) == set(norms)
    worst = []
    for n, ref in norms.items():
        got = named[n].grad.double().norm().item()
        if ref <= floor:
            assert got <= 1e-4 * max(norms.values()), (n, got)
        else:
            assert abs(got - ref) / ref < TOL, (n, got, ref)
    for k in g:
        n = k[len(tag) + 3:]
        if k.startswith(f"{tag}_g_") and norms[n] > floor:
            worst.append((rel(named[n].grad, g[k]), n))
    worst.sort(reverse=True)
    print(f"UNet fp32 regime ({tag}): largest per-tensor gradient errors vs the reference:", worst[:4])
    assert worst[0][0] < 2e-5, worst[:4]  # (cancelling 1-D sums over 1024 pixels: a few 1e-6 of f32 round-off)
    g1 = m._flat_grad.clone()
    m.zero_grad()
    gd.compute_loss({"x": x0.to(DEV), **cond}, timesteps=ti.to(DEV), noise=noise.to(DEV))["loss"].backward()
    assert torch.equal(g1, m._flat_grad), "no atomics: a step is bit-reproducible"


def test_unet_fp32_ddpm_sampling_and_trainer_switch(tmp_path):
    """`BaseTrainer()` (precision_type "no", the reference's default) switches a UNet to its fp32 launch sequences at prepare(); a
    respaced DDPM sampling loop (hipGraph replay from the second step on) equals the oracle's loop"""
    from diffulab_amd import Diffuser
    from diffulab_amd.networks.denoisers import UNetModel
    from diffulab_amd.training import BaseTrainer, FusedAdamW

    kw = UNET_VARIANTS["un"]
    cfg = ounet.UNetConfig(**kw)
    mk = dict(kw, image_size=list(kw["image_size"]), attention_resolutions=list(kw["attention_resolutions"]),
              channel_mult=", ".join(map(str, kw["channel_mult"])))
    m = UNetModel(**mk)
    P = synth.generic_params(ounet.param_shapes(cfg), seed=23)
    m.load_state_dict(P)
    d = Diffuser(m, sampling_method="ddpm", model_type="gaussian_diffusion", n_steps=1000)
    tr = BaseTrainer(n_epoch=1, save_path=tmp_path, project_name="u32", use_ema=False)
    tr.prepare(d, FusedAdamW(m.parameters(), lr=1e-3))
    assert m.precision == "fp32"
    B = 2
    x = synth.normal("us.x", (B, 1, 16, 16))
    y = synth.integers("us.y", (B,), 10)
    with torch.no_grad():
        for tv in (999.0, 500.0, 3.0):
            t = torch.full((B,), tv)
            got = m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV))["x"]
            assert rel(got, ounet.unet_forward(P, x, t, y, cfg)) < TOL
