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
