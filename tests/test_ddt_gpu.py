"""GPU parity of the DDT row (per-token-modulation decoder; configs/model/ddt.yaml = simple_ddt): the
per-token LayerNorm-modulate backward and the decoder-conditioning kernels through the C ABI against torch autograd, and DDT end to
end against outputs of the reference module (tests/golden/ddt.npz)."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import ddt as oddt  # noqa: E402
from oracle import synth  # noqa: E402

DEV = "cuda"
KW = dict(input_channels=4, output_channels=4, inner_dim=128, num_heads=2, mlp_ratio=4, patch_size=2, encoder_depth=2, decoder_depth=2,
          n_classes=10, classifier_free=True)


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("D,affine", [(128, True), (768, True), (384, False)])
def test_ln_modulate_per_token_forward_backward(D, affine):
    """modulate(LN(x + gate_prev * t_prev)) with one (scale, shift, gate) row per token, and its backward incl. the fused backward of
    the gated residual that follows (dt = gate * dx, dgate = dx * t per token) against autograd"""
    from diffulab_amd import ops

    M, R = 320, 3 * D  # modulation matrix [M, R]: columns [scale | shift | gate]
    x, dout, dres = (bf(synth.normal(f"lt.{n}{D}", (M, D))) for n in ("x", "do", "dr"))
    mod = bf(synth.normal(f"lt.mod{D}", (M, R)) * 0.3)
    gt = bf(synth.normal(f"lt.gt{D}", (M, D)))
    w = (1 + 0.1 * synth.normal(f"lt.w{D}", (D,))) if affine else None
    b = (0.05 * synth.normal(f"lt.b{D}", (D,))) if affine else None
    xr, modr = x.float().requires_grad_(True), mod.float().requires_grad_(True)
    wr = w.clone().requires_grad_(True) if affine else None
    br = b.clone().requires_grad_(True) if affine else None
    y = F.layer_norm(xr, (D,), wr, br, 1e-5) * (1 + modr[:, :D]) + modr[:, D : 2 * D]
    out = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    md = mod.to(DEV)
    ops.ln_modulate_fwd(x.to(DEV), w.to(DEV) if affine else None, b.to(DEV) if affine else None, md[:, :D], md[:, D : 2 * D], 1, 1e-5,
                        out, mean, rstd)
    assert rel(out.float(), y) < 4e-3
    # loss = <y, dout> + <x, dres> (the residual path) ; the gate backward consumes dx
    (y * dout.float()).sum().add((xr * dres.float()).sum()).backward()
    dx = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    dmod = torch.zeros(M, R, device=DEV, dtype=torch.bfloat16)
    dt = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    part = torch.zeros(8, 2, D, device=DEV) if affine else None
    ops.ln_modulate_bwd_tok(dout.to(DEV), x.to(DEV), w.to(DEV) if affine else None, b.to(DEV) if affine else None, md[:, :D], mean, rstd,
                            dres.to(DEV), dx, dmod[:, :D], dmod[:, D : 2 * D], part, gate_t=gt.to(DEV), gate=md[:, 2 * D :], dt=dt,
                            dgate=dmod[:, 2 * D :])
    assert rel(dx.float(), xr.grad) < 6e-3
    assert rel(dmod[:, :D].float(), modr.grad[:, :D]) < 6e-3 and rel(dmod[:, D : 2 * D].float(), modr.grad[:, D : 2 * D]) < 6e-3
    dxs = dx.float().cpu()
    assert rel(dt.float(), dxs * mod.float()[:, 2 * D :]) < 6e-3 and rel(dmod[:, 2 * D :].float(), dxs * gt.float()) < 6e-3
    if affine:
        assert rel(part.sum(0)[0], wr.grad) < 6e-3 and rel(part.sum(0)[1], br.grad) < 6e-3


def test_ddt_conditioning_kernels():
    from diffulab_amd import ops

    B, N, D = 3, 64, 128
    enc = bf(synth.normal("dc.enc", (B * N, D)))
    temb = synth.normal("dc.t", (B, D))
    er, tr = enc.float().requires_grad_(True), temb.clone().requires_grad_(True)
    sz = F.silu(F.silu(er.view(B, N, D) + tr[:, None, :])).view(B * N, D)
    out = torch.empty(B * N, D, device=DEV, dtype=torch.bfloat16)
    ops.ddt_cond_fwd(enc.to(DEV), temb.to(DEV), B, N, out)
    assert rel(out.float(), sz) < 4e-3
    d = bf(synth.normal("dc.d", (B * N, D)))
    sz.backward(d.float())
    denc = torch.empty(B * N, D, device=DEV, dtype=torch.bfloat16)
    dtemb = torch.zeros(B, D, device=DEV)
    ops.ddt_cond_bwd(d.to(DEV), enc.to(DEV), temb.to(DEV), B, N, denc, dtemb)
    assert rel(denc.float(), er.grad) < 5e-3 and rel(dtemb, tr.grad) < 2e-3


def _model():
    from diffulab_amd import DDT

    cfg = oddt.DDTConfig(**KW)
    shapes = oddt.param_shapes(cfg)
    m = DDT(simple_ddt=True, **KW)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == shapes
    m.load_state_dict(synth.dit_params(shapes, seed=91))
    return m.to(DEV)


def test_ddt_training_step_against_reference_fixture(golden):
    g = {k: torch.as_tensor(v) for k, v in golden("ddt").items()}
    m = _model()
    B, H = 4, 16
    x, t, y = synth.normal("dd.x", (B, 4, H, H)), synth.uniform("dd.t", (B,), lo=0.05, hi=0.95), synth.integers("dd.y", (B,), 10)
    dy = synth.normal("dd.dy", (B, 4, H, H))
    m.train()
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV), p=0.0)["x"]
    assert rel(pred, g["pred"]) < 1.5e-2
    (pred * dy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    # the fixture stores every vector and one matrix of each kind; the oracle (pinned by the same fixture) covers every parameter
    cfg = oddt.DDTConfig(**KW)
    Pr = {k: v.requires_grad_(True) for k, v in synth.dit_params(oddt.param_shapes(cfg), seed=91).items()}
    (oddt.ddt_forward(Pr, x, t, y, cfg) * dy).sum().backward()
    bad = []
    for n, p in m.named_parameters():
        tol = 8e-2 if n.endswith(("bias", "scale")) or "norm" in n else 4e-2
        if "g_" + n in g and rel(p.grad, g["g_" + n]) > tol:
            bad.append((n, "fixture", rel(p.grad, g["g_" + n])))
        if rel(p.grad, Pr[n].grad) > tol:
            bad.append((n, "oracle", rel(p.grad, Pr[n].grad)))
    assert not bad, bad


def test_ddt_guided_sampling_and_optimizer_steps(golden):
    from diffulab_amd import Diffuser
    from diffulab_amd.training import FusedAdamW

    g = {k: torch.as_tensor(v) for k, v in golden("ddt").items()}
    m = _model()
    y = synth.integers("dd.y", (4,), 10)
    m.eval()
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    out = d.generate({"x": synth.normal("dd.init", (4, 4, 16, 16)).to(DEV), "y": y.to(DEV)}, use_tqdm=False, guidance_scale=2.0)
    assert rel(out["x"], g["loop_x"]) < 3e-2
    m.train()
    opt = FusedAdamW(m.parameters(), lr=2e-3, weight_decay=0.0)
    x0 = synth.normal("dd.x0", (8, 4, 16, 16)).to(DEV)
    yy = synth.integers("dd.yy", (8,), 10).to(DEV)
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = d.compute_loss({"x": x0.clone(), "y": yy, "p": 0.1}, timesteps=d.draw_timesteps(8))["loss"]
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(v == v for v in losses) and sum(losses[-5:]) < sum(losses[:5])


# ------------------------------------------------------------------ joint text-image encoder form (simple_ddt=False)
JKW = dict(input_channels=4, output_channels=4, inner_dim=128, num_heads=2, mlp_ratio=4, patch_size=2, encoder_depth=2, decoder_depth=2,
           rope_axes_dim=[16, 24, 24], rope_base=1000, classifier_free=True)


def test_ddt_joint_encoder_against_reference_fixture_and_oracle(golden):
    """prediction vs the reference; gradients vs the reference (stored subset) and vs the oracle (every parameter); context drop"""
    from diffulab_amd import DDT
    from diffulab_amd.networks.embedders import PrecomputedEmbedder

    raw = golden("ddt_joint")
    none = set(str(n) for n in raw["none"])
    g = {k: torch.as_tensor(v) for k, v in raw.items() if k != "none"}
    Lc, Cd, B, H = 64, 96, 4, 16
    null = synth.normal("dj.null", (1, Lc, Cd)) * 0.5
    m = DDT(simple_ddt=False, context_embedder=PrecomputedEmbedder(null, null_embedding_seq_len=7), **JKW)
    cfg = oddt.DDTJointConfig(context_dim=Cd, **JKW)
    shapes = oddt.joint_param_shapes(cfg)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == shapes
    P = synth.dit_params(shapes, seed=101)
    m.load_state_dict(P)
    m = m.to(DEV)
    x, t = synth.normal("dj.x", (B, 4, H, H)), synth.uniform("dj.t", (B,), lo=0.05, hi=0.95)
    ctx = synth.normal("dj.ctx", (B, Lc, Cd))
    keep = torch.arange(Lc)[None, :] < torch.tensor([64, 20, 41, 5])[:, None]
    dy = synth.normal("dj.dy", (B, 4, H, H))
    ic = {"embeddings": ctx.to(DEV), "attn_mask": keep.to(DEV)}
    m.train()
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), initial_context=ic, p=0.0)["x"]
    assert rel(pred, g["pred"]) < 1.5e-2
    (pred * dy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    (oddt.ddt_joint_forward(Pr, x, t, ctx, keep, cfg) * dy).sum().backward()
    bad = []
    for n, p in m.named_parameters():
        if n in none:
            assert float(p.grad.abs().max()) == 0.0, n
            continue
        tol = 8e-2 if p.dim() == 1 else 4e-2
        if "g_" + n in g and rel(p.grad, g["g_" + n]) > tol:
            bad.append((n, "fixture", rel(p.grad, g["g_" + n])))
        if rel(p.grad, Pr[n].grad) > tol:
            bad.append((n, "oracle", rel(p.grad, Pr[n].grad)))
    assert not bad, bad
    m.context_embedder._draw_drop = lambda batch_size, p, device: g["b_u"].to(device) < p
    with torch.no_grad():
        assert rel(m(x=x.to(DEV), timesteps=t.to(DEV), initial_context=ic, p=0.5)["x"], g["b_pred"]) < 1.5e-2


@pytest.mark.parametrize("family", ["ddt", "sprint"])
def test_in_place_qk_backward_path_equals_the_two_kernel_path(family, monkeypatch):
    """From 32768 token rows per launch the DDT / SPRINT engines take the token-major attention backward + in-place QK-norm backward
    (ops.qk_inplace_ok) instead of head-major dQ / dK + qk_norm_rope_bwd.  The fixtures above run 1024 rows, i.e. the two-kernel
    path; here both paths run the same step at B = 128 x 256 tokens and every gradient agrees to bf16 rounding of dQ / dK."""
    from diffulab_amd import DDT, SprintDiT

    B, H = 128, 32
    x, t, y = synth.normal("ip.x", (B, 4, H, H)), synth.uniform("ip.t", (B,), lo=0.05, hi=0.95), synth.integers("ip.y", (B,), 10)
    dy = synth.normal("ip.dy", (B, 4, H, H))
    grads = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("DL_QK_INPLACE", flag)
        if family == "ddt":
            m = DDT(simple_ddt=True, **KW)
        else:
            m = SprintDiT(simple_dit=True, input_channels=4, output_channels=4, inner_dim=128, embedding_dim=128, num_heads=2, mlp_ratio=4,
                          patch_size=2, encoder_depth=1, deep_layers_depth=1, decoder_depth=1, n_classes=10, classifier_free=True,
                          drop_rate=0.75)
            scores = synth.normal("ip.scores", (B, 256))
            m._draw_scores = lambda Bn, S, device: scores.to(device)
        # (adaLN-zero would leave the attention branch without gradient: every parameter gets seeded noise, norm weights around 1)
        m.load_state_dict({k: synth.normal("ip.p." + k, tuple(v.shape)) * (v.shape[-1] ** -0.5 if v.dim() > 1 else 0.1)
                           + (1.0 if k.endswith(("norm_1.weight", "norm_2.weight", "scale")) else 0.0) for k, v in m.state_dict().items()})
        m = m.to(DEV)
        m.train()
        pred = m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV), p=0.0)["x"]
        (pred * dy.to(DEV)).sum().backward()
        torch.cuda.synchronize()
        assert (m.engine.ws.get("qk_part") is not None) == (flag == "1")
        grads[flag] = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    bad = [(n, rel(g, grads["0"][n])) for n, g in grads["1"].items() if rel(g, grads["0"][n]) > 4e-3]
    assert not bad, bad


def test_ddt_training_step_at_tiled_weight_gradient_dims_against_oracle():
    """The fixture dims (128 wide, 256 token rows) are below every tiled weight-gradient form.  256 wide / 4 heads at B = 8 x 256 tokens
    (2048 rows) is the smallest DDT the engine issues like the 512-wide configurations: the blocks' four weight gradients as one
    atomics-free launch on 256 x 256 tiles (ops.WgradGroups) and the stacked per-token adaLN weight gradient [2048, 3584]^T [2048, 256]
    through the same kernel (ws["tmod_slab"]); prediction and every parameter gradient against the fp32 oracle."""
    from diffulab_amd import DDT

    kw = dict(KW, inner_dim=256, num_heads=4)
    cfg = oddt.DDTConfig(**kw)
    shapes = oddt.param_shapes(cfg)
    m = DDT(simple_ddt=True, **kw)
    m.load_state_dict(synth.dit_params(shapes, seed=93))
    m = m.to(DEV)
    B, H = 8, 32
    x, t, y = synth.normal("dt.x", (B, 4, H, H)), synth.uniform("dt.t", (B,), lo=0.05, hi=0.95), synth.integers("dt.y", (B,), 10)
    dy = synth.normal("dt.dy", (B, 4, H, H))
    m.train()
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV), p=0.0)["x"]
    (pred * dy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    ws = m.engine.ws
    assert ws.get("tn_slab") is not None and ws.get("tmod_slab") is not None  # the tiled forms were the ones that ran
    Pr = {k: v.requires_grad_(True) for k, v in synth.dit_params(shapes, seed=93).items()}
    ref = oddt.ddt_forward(Pr, x, t, y, cfg)
    (ref * dy).sum().backward()
    assert rel(pred, ref) < 1.5e-2
    bad = []
    for n, p in m.named_parameters():
        tol = 8e-2 if n.endswith(("bias", "scale")) or "norm" in n else 4e-2
        if rel(p.grad, Pr[n].grad) > tol:
            bad.append((n, rel(p.grad, Pr[n].grad)))
    assert not bad, bad
