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


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("depth,align,B", [(3, 2, 4), (12, 8, 128)])
def test_repa_config_dims_against_oracle(depth, align, B):
    """dims of configs/train_imagenet_flow_matching_repa.yaml (BASELINE config 4: DC-AE latents 32x8x8, patch 1 -> 64 tokens, inner
    768, 12 heads, embedding 256), REPA with hidden 1024 / target dim 1024 (no resampler): losses and every gradient against the CPU
    oracle -- at depth 3 / B=4 (quick) and at the CONFIG's depth 12, alignment layer 8 and batch 128 (it had only run
    at depth 3, B=4).  Covers the D = 768 row kernels, the 64-token attention and, at B=128, the grouped weight gradients."""
    from diffulab_amd import Diffuser, MMDiT
    from diffulab_amd.training.losses import RepaLoss
    from oracle import diffusion as od

    kw = dict(input_channels=32, output_channels=32, inner_dim=768, embedding_dim=256, num_heads=12, mlp_ratio=4, patch_size=1,
              depth=depth, n_classes=1000, classifier_free=True)
    cfg = odit.DiTConfig(**kw)
    P = synth.dit_params(odit.param_shapes(cfg), seed=13)
    R = synth.generic_params(orepa.param_shapes(768, 1024, 1024), seed=43)
    m = MMDiT(simple_dit=True, **kw)
    m.load_state_dict(P)
    m = m.to(DEV)
    rl = RepaLoss(alignment_layer=align, denoiser_dimension=768, hidden_dim=1024, load_dino=False, embedding_dim=1024, coeff=0.5)
    rl.load_state_dict(R)
    rl = rl.to(DEV)
    rl.set_model(m)
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
    ref_repa = orepa.repa_loss(Rr, taps[f"layer{align - 1}"], dst, coeff=0.5)
    (ref_loss + ref_repa).backward()
    assert abs(losses["loss"].item() - ref_loss.item()) / ref_loss.item() < 2e-3
    assert abs(losses["RepaLoss"].item() - ref_repa.item()) / ref_repa.item() < 2e-3
    for n, p in rl.named_parameters():
        assert rel(p.grad, Rr[n].grad) < 3e-2, n
    bad = [(n, rel(p.grad, Pr[n].grad)) for n, p in m.named_parameters() if Pr[n].grad.norm() > 0]
    bad = [(n, e) for n, e in bad if e > 3e-2]
    assert not bad, bad


# ------------------------------------------------------------------ Perceiver resampler (use_resampler: true)
def test_gelu_epilogue_and_backward_kernels():
    """exact-erf GELU fused in the NT GEMM epilogue (FeedForward, perceiver_resampler.py:66-93) and its backward"""
    from diffulab_amd import ops

    M, N, K = 320, 192, 128
    a = synth.normal("ge.a", (M, K)).to(torch.bfloat16)
    w = (synth.normal("ge.w", (N, K)) * K**-0.5).to(torch.bfloat16)
    pre_ref = a.float() @ w.float().t()
    out, pre = (torch.empty(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    ops.gemm_nt(a.to(DEV), w.to(DEV), out, act=ops.ACT_GELU, pre_out=pre)
    assert rel(pre.float(), pre_ref) < 4e-3 and rel(out.float(), F.gelu(pre_ref)) < 4e-3
    p = pre.float().cpu().requires_grad_(True)
    dy = synth.normal("ge.dy", (M, N))
    F.gelu(p).backward(dy)
    dx = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.gelu_bwd(dy.to(DEV), pre, dx)
    assert rel(dx.float(), p.grad) < 4e-3


def test_heads_split_rope_and_merge_backward():
    """'b n (h d) -> b h n d' into a row window of a longer key buffer with N-D RoPE on the way, and the adjoint"""
    from diffulab_amd import ops
    from diffulab_amd.engine import rope_grid_tables

    B, H, n, m, Nk = 2, 3, 64, 256, 512
    cos, sin = rope_grid_tables(8, 8, [16, 32], 10_000.0)  # rot = 48 < 64: the last 16 channels pass through
    src = synth.normal("hs.src", (B * n, 2 * H * 64)).to(torch.bfloat16)
    lat = synth.normal("hs.lat", (B * m, H * 64)).to(torch.bfloat16)
    k = torch.zeros(B, H, Nk, 64, device=DEV, dtype=torch.bfloat16)
    sd, ld = src.to(DEV), lat.to(DEV)
    ops.heads_split_rope(sd[:, H * 64:], k, B, H, n, 0, cos.to(DEV), sin.to(DEV), 48)  # second half of a fused kv row
    ops.heads_split_rope(ld, k, B, H, m, n)
    xs = src[:, H * 64:].float().reshape(B, n, H, 64)
    rot = odit.apply_rope(xs[..., :48], cos, sin)
    want = torch.cat((torch.cat((rot, xs[..., 48:]), -1), lat.float().reshape(B, m, H, 64)), 1).transpose(1, 2)
    got = k.float().cpu()
    assert rel(got[:, :, : n + m], want) < 3e-3 and float(got[:, :, n + m:].abs().max()) == 0.0
    # adjoint: <split(x), g> == <x, merge(g)>
    gk = synth.normal("hs.g", (B, H, Nk, 64)).to(torch.bfloat16)
    dsrc = torch.zeros(B * n, 2 * H * 64, device=DEV, dtype=torch.bfloat16)
    ops.heads_merge_rope_bwd(gk.to(DEV), dsrc[:, H * 64:], B, H, n, 0, cos.to(DEV), sin.to(DEV), 48)
    lhs = (got[:, :, :n].double() * gk[:, :, :n].double()).sum()
    rhs = (src[:, H * 64:].double() * dsrc[:, H * 64:].double().cpu()).sum()
    assert abs(lhs - rhs) / abs(lhs) < 5e-3 and float(dsrc[:, : H * 64].abs().max()) == 0.0
    ops.heads_merge_rope_bwd(gk.to(DEV), dsrc[:, H * 64:], B, H, n, 0, cos.to(DEV), sin.to(DEV), 48, accumulate=True)
    assert abs(2 * lhs - (src[:, H * 64:].double() * dsrc[:, H * 64:].double().cpu()).sum()) / abs(lhs) < 1e-2


RS = dict(dim=128, depth=2, head_dim=64, num_heads=2, ff_mult=4, num_latents=256)


def _resampler():
    from diffulab_amd.networks.repa import PerceiverResampler

    m = PerceiverResampler(**RS)
    shapes = orepa.resampler_param_shapes(**RS)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == shapes  # the reference's checkpoint keys
    m.load_state_dict(synth.generic_params(shapes, seed=51))
    return m.to(DEV)


def test_perceiver_resampler_against_reference_fixture(golden):
    """64 input tokens (8x8 grid, keys padded 320 -> 512 and masked): output, input gradient and every parameter gradient vs
    the reference module's own outputs"""
    g = golden("resampler")
    m = _resampler()
    x = synth.normal("rs.x", (3, 64, 128)).to(DEV).requires_grad_(True)
    y = m(x)
    assert y.shape == (3, 256, 128) and y.dtype == torch.bfloat16
    y.backward(synth.normal("rs.dy", (3, 256, 128)).to(DEV, torch.bfloat16))
    torch.cuda.synchronize()
    assert rel(y.float(), g["y"]) < 1.5e-2 and rel(x.grad, g["dx"]) < 3e-2
    for n, p in m.named_parameters():
        assert rel(p.grad, g["g_" + n]) < (6e-2 if p.dim() == 1 else 3e-2), n


def test_perceiver_resampler_unpadded_keys_against_oracle():
    """256 input tokens (16x16 grid): 512 keys exactly, no mask"""
    m = _resampler()
    B = 2
    P = {k: v.requires_grad_(True) for k, v in synth.generic_params(orepa.resampler_param_shapes(**RS), seed=51).items()}
    xc = synth.normal("rs2.x", (B, 256, 128)).to(torch.bfloat16).float().requires_grad_(True)
    dy = synth.normal("rs2.dy", (B, 256, 128)).to(torch.bfloat16).float()
    yc = orepa.perceiver_resampler(P, xc, depth=2, head_dim=64, num_heads=2)
    yc.backward(dy)
    x = xc.detach().to(DEV, torch.bfloat16).requires_grad_(True)
    y = m(x)
    y.backward(dy.to(DEV, torch.bfloat16))
    assert rel(y.float(), yc) < 1.5e-2 and rel(x.grad.float(), xc.grad) < 3e-2
    for n, p in m.named_parameters():
        assert rel(p.grad, P[n].grad) < (6e-2 if p.dim() == 1 else 3e-2), n


def test_repa_loss_with_resampler_against_oracle():
    """RepaLoss(use_resampler=True) end to end on a hooked DiT: loss value and the gradients that reach the DiT, the MLP and the
    resampler, against the oracle chain (features of the oracle DiT -> proj -> resampler -> cosine)"""
    from diffulab_amd import MMDiT
    from diffulab_amd.training.losses import RepaLoss

    cfg = odit.DiTConfig(**SMALL)
    P = synth.dit_params(odit.param_shapes(cfg), seed=5)
    m = MMDiT(simple_dit=True, **SMALL)
    m.load_state_dict(P)
    m = m.to(DEV)
    rl = RepaLoss(repa_encoder="dinov2", alignment_layer=2, denoiser_dimension=128, hidden_dim=128, load_dino=False,
                  embedding_dim=128, use_resampler=True, resampler_params=RS, coeff=0.5)
    Pm = synth.generic_params(orepa.param_shapes(128, 128, 128), seed=41)
    Pr = synth.generic_params(orepa.resampler_param_shapes(**RS), seed=51)
    rl.load_state_dict({**Pm, **{"resampler." + k: v for k, v in Pr.items()}})
    rl = rl.to(DEV)
    rl.set_model(m)
    B = 4
    x = synth.normal("rr.x", (B, 4, 16, 16))
    t, y = synth.uniform("rr.t", (B,), lo=0.05, hi=0.95), synth.integers("rr.y", (B,), 10)
    dst = synth.normal("rr.dst", (B, 256, 128))
    m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV), p=0.0)
    loss = rl(dst_features=dst.to(DEV))
    loss.backward()
    # oracle chain
    Po = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    Pmo = {k: v.clone().requires_grad_(True) for k, v in Pm.items()}
    Pro = {k: v.clone().requires_grad_(True) for k, v in Pr.items()}
    taps: dict = {}
    odit.dit_forward(Po, x, t, y, cfg, taps=taps)
    want = orepa.repa_loss_resampled(Pmo, Pro, taps["layer1"], dst, 0.5, depth=2, head_dim=64, num_heads=2)
    want.backward()
    assert abs(loss.item() - want.item()) < 5e-3 * abs(want.item())
    sd = dict(rl.named_parameters())
    for k in ("proj.0.weight", "proj.4.weight", "resampler.latents", "resampler.layers.0.0.to_kv.weight",
              "resampler.layers.1.1.3.weight"):
        ref = (Pmo[k] if k.startswith("proj") else Pro[k[len("resampler."):]]).grad
        assert rel(sd[k].grad, ref) < 5e-2, k
    gd = dict(m.named_parameters())
    for k in ("layers.0.attention.qkv.weight", "layers.1.mlp_input.0.weight"):
        assert rel(gd[k].grad, Po[k].grad) < 5e-2, k
