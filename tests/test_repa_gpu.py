"""GPU parity of the REPA row: RepaLoss hooked on a DiT block next to the flow loss, through the
plugin API (Diffuser.compute_loss(extra_losses) -> MMDiT forward hook -> HIP projection MLP + cosine kernels -> feature gradient
back into the DiT engine), against outputs of the reference's own RepaLoss (tests/golden/repa.npz) and the CPU oracle."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import dit as odit  # noqa: E402
from oracle import repa as orepa  # noqa: E402
from oracle import synth  # noqa: E402

DEV = "cuda"
SMALL = dict(input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4,
             patch_size=2, depth=2, n_classes=10, classifier_free=True)


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def test_cosine_rows_kernels():
    from diffulab_amd import ops

    M, E = 192, 72
    p = synth.normal("cs.p", (M, E)).to(torch.bfloat16).float().requires_grad_(True)
    d = synth.normal("cs.d", (M, E))
    cos = F.cosine_similarity(p, d, dim=-1)
    (0.5 * (1 - cos.mean())).backward()
    pd = p.detach().to(DEV, torch.bfloat16)
    cosv, pn2, dn2 = (torch.empty(M, device=DEV) for _ in range(3))
    ops.cosine_rows_fwd(pd, d.to(DEV), cosv, pn2, dn2)
    assert rel(cosv, cos) < 1e-5
    dp = torch.empty(M, E, device=DEV, dtype=torch.bfloat16)
    ops.cosine_rows_bwd(pd, d.to(DEV), cosv, pn2, dn2, -0.5 / M, torch.ones(1, device=DEV), dp)
    assert rel(dp.float(), p.grad) < 4e-3


def test_repa_loss_with_flow_loss_against_reference_fixture(golden):
    from diffulab_amd import Diffuser, MMDiT
    from diffulab_amd.training.losses import RepaLoss

    g = golden("repa")
    cfg = odit.DiTConfig(**SMALL)
    m = MMDiT(simple_dit=True, **SMALL)
    m.load_state_dict(synth.dit_params(odit.param_shapes(cfg), seed=5))
    m = m.to(DEV)
    rl = RepaLoss(repa_encoder="dinov2", alignment_layer=1, denoiser_dimension=128, hidden_dim=128, load_dino=False,
                  embedding_dim=64, coeff=0.5)
    assert {k: tuple(v.shape) for k, v in rl.state_dict().items()} == orepa.param_shapes(128, 128, 64)
    rl.load_state_dict(synth.generic_params(orepa.param_shapes(128, 128, 64), seed=41))
    rl = rl.to(DEV)
    rl.set_model(m)
    B, H = 4, 16
    x0, noise = synth.normal("rp.x0", (B, 4, H, H)), synth.normal("rp.noise", (B, 4, H, H))
    y, t = synth.integers("rp.y", (B,), 10), synth.uniform("rp.t", (B,), lo=0.05, hi=0.95)
    dst = synth.normal("rp.dst", (B, 64, 64))
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4, extra_losses=[rl])
    losses = d.compute_loss({"x": x0.to(DEV), "y": y.to(DEV), "p": 0.0}, timesteps=t, noise=noise.to(DEV),
                            extra_args={"dst_features": dst.to(DEV)})
    assert set(losses) == {"loss", "RepaLoss"}
    sum(losses.values()).backward()
    assert abs(losses["loss"].item() - float(g["loss"])) / float(g["loss"]) < 2e-3
    assert abs(losses["RepaLoss"].item() - float(g["repa"])) / float(g["repa"]) < 2e-3
    for n, p in rl.named_parameters():
        assert rel(p.grad, g["g_" + n]) < 2.5e-2, n
    for n, p in m.named_parameters():
        assert rel(p.grad, g["gd_" + n]) < 2.5e-2, n
    # validation path: no autograd, features still delivered to the hook
    with torch.no_grad():
        v = d.compute_loss({"x": x0.to(DEV), "y": y.to(DEV), "p": 0.0}, timesteps=t, noise=noise.to(DEV),
                           extra_args={"dst_features": dst.to(DEV)})
    assert abs(v["RepaLoss"].item() - float(g["repa"])) / float(g["repa"]) < 2e-3
    rl._unregister_all()
    assert not m.layers[0]._forward_hooks


def test_repa_config_dims_against_oracle():
    """dims of configs/train_imagenet_flow_matching_repa.yaml (DC-AE latents 32x8x8, patch 1 -> 64 tokens, inner 768, 12 heads,
    embedding 256) at depth 3, REPA hooked on block 2 with hidden 1024 / target dim 1024 (no resampler): losses and every
    gradient against the CPU oracle -- covers the D = 768 row kernels and the 64-token attention"""
    from diffulab_amd import Diffuser, MMDiT
    from diffulab_amd.training.losses import RepaLoss
    from oracle import diffusion as od

    kw = dict(input_channels=32, output_channels=32, inner_dim=768, embedding_dim=256, num_heads=12, mlp_ratio=4, patch_size=1,
              depth=3, n_classes=1000, classifier_free=True)
    cfg = odit.DiTConfig(**kw)
    P = synth.dit_params(odit.param_shapes(cfg), seed=13)
    R = synth.generic_params(orepa.param_shapes(768, 1024, 1024), seed=43)
    m = MMDiT(simple_dit=True, **kw)
    m.load_state_dict(P)
    m = m.to(DEV)
    rl = RepaLoss(alignment_layer=2, denoiser_dimension=768, hidden_dim=1024, load_dino=False, embedding_dim=1024, coeff=0.5)
    rl.load_state_dict(R)
    rl = rl.to(DEV)
    rl.set_model(m)
    B = 4
    x0, noise = synth.normal("rb.x0", (B, 32, 8, 8)), synth.normal("rb.noise", (B, 32, 8, 8))
    y, t = synth.integers("rb.y", (B,), 1000), synth.uniform("rb.t", (B,), lo=0.05, hi=0.95)
    dst = synth.normal("rb.dst", (B, 64, 1024))
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4, extra_losses=[rl])
    losses = d.compute_loss({"x": x0.to(DEV), "y": y.to(DEV), "p": 0.0}, timesteps=t, noise=noise.to(DEV),
                            extra_args={"dst_features": dst.to(DEV)})
    sum(losses.values()).backward()
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    Rr = {k: v.clone().requires_grad_(True) for k, v in R.items()}
    taps: dict = {}
    pred = odit.dit_forward(Pr, od.flow_add_noise(x0, t, noise), t, y, cfg, taps=taps)
    ref_loss = od.flow_loss(pred, x0, noise)
    ref_repa = orepa.repa_loss(Rr, taps["layer1"], dst, coeff=0.5)
    (ref_loss + ref_repa).backward()
    assert abs(losses["loss"].item() - ref_loss.item()) / ref_loss.item() < 2e-3
    assert abs(losses["RepaLoss"].item() - ref_repa.item()) / ref_repa.item() < 2e-3
    for n, p in rl.named_parameters():
        assert rel(p.grad, Rr[n].grad) < 3e-2, n
    bad = [(n, rel(p.grad, Pr[n].grad)) for n, p in m.named_parameters() if Pr[n].grad.norm() > 0]
    bad = [(n, e) for n, e in bad if e > 3e-2]
    assert not bad, bad
