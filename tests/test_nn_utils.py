"""`diffulab.networks.utils.nn` (reference networks/utils/nn.py:11-540) as an importable module: the standalone primitives against the
reference fixture `prims.npz` (generated from the imported reference), the oracle, and -- on the GPU -- their kernel-backed forms
against the torch expressions of the same module (VERDICT r4 missing #4)."""

import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import dit as odit  # noqa: E402
from oracle import synth  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "prims.npz"))


def test_module_resolves_under_the_reference_name_and_matches_the_reference_fixture():
    from diffulab.networks.utils import nn as dnn
    from diffulab_amd.networks.utils import nn as dnn2

    assert dnn is dnn2
    t = synth.uniform("prims.t", (8,), lo=0.0, hi=1.0)
    ti = torch.tensor([0, 1, 17, 500, 999], dtype=torch.int32)
    assert torch.equal(dnn.timestep_embedding(t, 256), torch.from_numpy(G["temb_f"]))
    assert torch.equal(dnn.timestep_embedding(ti, 128), torch.from_numpy(G["temb_i"]))
    assert torch.equal(dnn.timestep_embedding(t, 9), torch.from_numpy(G["temb_odd"]))
    pos = torch.stack(torch.meshgrid([torch.arange(16), torch.arange(16)], indexing="ij"), dim=-1).view(-1, 2)[None]
    cos, sin = dnn.get_cos_sin_ndim_grid(pos, base=10_000, axes_dim=[32, 32])
    assert torch.equal(cos[0], torch.from_numpy(G["rope_cos_16x16"])) and torch.equal(sin[0], torch.from_numpy(G["rope_sin_16x16"]))
    pos2 = torch.stack(torch.meshgrid([torch.arange(3), torch.arange(5)], indexing="ij"), dim=-1).view(-1, 2)[None]
    cos2, sin2 = dnn.get_cos_sin_ndim_grid(pos2, base=2000, axes_dim=[8, 24])
    assert torch.equal(cos2[0], torch.from_numpy(G["rope_cos_3x5"])) and torch.equal(sin2[0], torch.from_numpy(G["rope_sin_3x5"]))
    # the layer classes against the oracle's restatement of the same reference lines (state_dict names are the reference's)
    x = synth.normal("nnu.x", (2, 15, 2, 32))
    rope = dnn.RotaryPositionalEmbeddingNDim([8, 24])
    q, k, v = rope(x, 2 * x, 3 * x, (cos2.expand(2, -1, -1), sin2.expand(2, -1, -1)))
    assert torch.allclose(q, odit.apply_rope(x, cos2[0], sin2[0]), atol=1e-6) and torch.equal(v, 3 * x)
    assert torch.allclose(k, odit.apply_rope(2 * x, cos2[0], sin2[0]), atol=1e-6)
    qk = dnn.QKNorm(64)
    assert sorted(qk.state_dict()) == ["key_norm.scale", "query_norm.scale"]
    with torch.no_grad():
        qk.query_norm.scale.copy_(1 + synth.normal("nnu.s", (64,), std=0.1))
    a = synth.normal("nnu.a", (3, 5, 64))
    qn, kn = qk(a, a, a.double())
    assert qn.dtype == torch.float64 and torch.allclose(qn.float(), odit.rms_norm(a, qk.query_norm.scale.detach()), atol=1e-6)
    u = synth.normal("nnu.u", (4, 7, 32))
    assert torch.allclose(dnn.PackedSwiGLU()(u), odit.silu(u[..., :16]) * u[..., 16:], atol=1e-6)
    mod = dnn.Modulation(8, 4)
    out = mod(synth.normal("nnu.v", (3, 8)))
    assert sorted(mod.state_dict()) == ["lin.bias", "lin.weight"] and out.alpha.shape == (3, 1, 4) and out.zeta.shape == (3, 1, 4)
    assert torch.equal(dnn.modulate(a, 0.5 * a, a), a * (1 + 0.5 * a) + a)
    le = dnn.LabelEmbed(10, 6, classifier_free_guidance=True)
    assert le.embedding.weight.shape == (11, 6)
    torch.manual_seed(3)
    want = torch.where(torch.rand(5) < 0.5, 10, torch.arange(5))
    torch.manual_seed(3)
    assert torch.equal(le.drop_labels(torch.arange(5), 0.5), want)
    with pytest.raises(AssertionError):
        dnn.LabelEmbed(10, 6)(torch.arange(5), p=0.5)
    assert dnn.Upsample(4, True, 6)(torch.zeros(1, 4, 3, 3)).shape == (1, 6, 6, 6)
    assert dnn.Downsample(4, False)(torch.zeros(1, 4, 6, 6)).shape == (1, 4, 3, 3)
    gn = dnn.normalization(64)
    assert isinstance(gn, dnn.GroupNorm32) and gn(torch.randn(2, 64, 4, 4).half()).dtype == torch.float16


@pytest.mark.gpu
def test_kernel_backed_forms_equal_the_torch_expressions():
    from diffulab.networks.utils import nn as dnn

    dev = "cuda"
    t = synth.uniform("prims.t", (8,), lo=0.0, hi=1.0).to(dev)
    got = dnn.timestep_embedding(t, 256)
    assert got.dtype == torch.float32 and float((got.cpu() - torch.from_numpy(G["temb_f"])).abs().max()) < 2e-6
    # PackedSwiGLU: forward + backward on the standalone kernels, both dtypes
    for dtype, tol in ((torch.float32, 2e-6), (torch.bfloat16, 1e-2)):
        u = synth.normal("nnu.gu", (3, 40, 256)).to(dev, dtype).requires_grad_(True)
        g = synth.normal("nnu.gg", (3, 40, 128)).to(dev, dtype)
        h = dnn.PackedSwiGLU()(u)
        h.backward(g)
        ur = u.detach().float().requires_grad_(True)
        hr = torch.nn.functional.silu(ur[..., :128]) * ur[..., 128:]
        hr.backward(g.float())
        rel = lambda a, b: float((a.float() - b).norm() / b.norm())  # noqa: E731
        assert h.shape == (3, 40, 128) and rel(h, hr.detach()) < tol and rel(u.grad, ur.grad) < tol
    # the fused QKNorm + RoPE + head split against the separate classes of the same module
    B, gh, gw, H, dh = 3, 8, 8, 2, 64
    N, D = gh * gw, H * dh
    qkv = synth.normal("nnu.qkv", (B * N, 3 * D)).to(dev, torch.bfloat16).requires_grad_(True)
    qk = dnn.QKNorm(D).to(dev)
    with torch.no_grad():
        qk.query_norm.scale.copy_(1 + synth.normal("nnu.sq", (D,), std=0.1).to(dev))
        qk.key_norm.scale.copy_(1 + synth.normal("nnu.sk", (D,), std=0.1).to(dev))
    pos = torch.stack(torch.meshgrid([torch.arange(gh), torch.arange(gw)], indexing="ij"), dim=-1).view(-1, 2)[None]
    cs = dnn.get_cos_sin_ndim_grid(pos, 10_000.0, [32, 32])
    q, k, v = dnn.qk_norm_rope(qkv, qk, cs, B, N, H)
    wq, wk, wv = (synth.normal(f"nnu.w{i}", (B, H, N, dh)).to(dev) for i in range(3))
    ((q.float() * wq).sum() + (k.float() * wk).sum() + (v.float() * wv).sum()).backward()
    x = qkv.detach().float().view(B, N, 3 * D).requires_grad_(True)
    qk32 = dnn.QKNorm(D).to(dev)
    qk32.load_state_dict(qk.state_dict())
    qr, kr = qk32(x[..., :D], x[..., D : 2 * D], x[..., 2 * D :])
    rope = dnn.RotaryPositionalEmbeddingNDim([32, 32])
    cosb, sinb = (c.to(dev).expand(B, -1, -1) for c in cs)
    qr, kr, vr = rope(qr.view(B, N, H, dh), kr.view(B, N, H, dh), x[..., 2 * D :].view(B, N, H, dh), (cosb, sinb))
    qr, kr, vr = (z.transpose(1, 2) for z in (qr, kr, vr))
    ((qr * wq).sum() + (kr * wk).sum() + (vr * wv).sum()).backward()
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())  # noqa: E731
    assert rel(q, qr) < 6e-3 and rel(k, kr) < 6e-3 and rel(v, vr) < 1e-6
    assert rel(qkv.grad, x.grad.view(B * N, 3 * D)) < 1.5e-2
    assert rel(qk.query_norm.scale.grad, qk32.query_norm.scale.grad) < 1.5e-2 and rel(qk.key_norm.scale.grad, qk32.key_norm.scale.grad) < 1.5e-2
    # GroupNorm32 inference on bf16 NCHW through the UNet engine's kernels
    gn = dnn.normalization(64).to(dev)
    with torch.no_grad():
        gn.weight.copy_(1 + synth.normal("nnu.gw", (64,), std=0.1).to(dev))
        gn.bias.copy_(synth.normal("nnu.gb", (64,), std=0.1).to(dev))
        xi = synth.normal("nnu.gx", (2, 64, 8, 8)).to(dev, torch.bfloat16)
        got = gn(xi)
        want = torch.nn.functional.group_norm(xi.float(), 32, gn.weight, gn.bias, gn.eps)
    assert got.dtype == torch.bfloat16 and got.shape == xi.shape and rel(got, want) < 6e-3
