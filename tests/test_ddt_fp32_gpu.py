"""The fp32-class regime of DDT(simple_ddt=True) -- the precision `model=ddt` gets in the reference's class-conditional
configurations (they inherit trainer/default.yaml's precision_type "no"; configs/train_cifar10_ddt.yaml here) -- through the C ABI: the f32 decoder-conditioning kernels and the launch sequences of
ddt_engine_f32.py (per-token adaLN through the f32 LayerNorm kernels with one modulation row per token) against (1) outputs of the
reference module (tests/golden/ddt.npz) and (2) the CPU oracle.  Bar: per-tensor relative L2 <= 1e-5."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import ddt as oddt  # noqa: E402
from oracle import synth  # noqa: E402

DEV = "cuda"
TOL = 1e-5
KW = dict(input_channels=4, output_channels=4, inner_dim=128, num_heads=2, mlp_ratio=4, patch_size=2, encoder_depth=2, decoder_depth=2,
          n_classes=10, classifier_free=True)


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def test_f32_ddt_conditioning_and_per_token_layernorm_against_torch():
    from diffulab_amd import ops

    B, N, D = 3, 64, 128
    g = torch.Generator().manual_seed(7)
    enc2 = torch.randn(B * N, 2 * D, generator=g)  # (a strided view: the encoder output of another engine lives in a wider buffer)
    enc, temb = enc2[:, D:], torch.randn(B, D, generator=g)
    er, tr = enc.clone().requires_grad_(True), temb.clone().requires_grad_(True)
    sz = F.silu(F.silu(er.view(B, N, D) + tr[:, None, :])).view(B * N, D)
    out = torch.empty(B * N, D, device=DEV)
    ops.f32_ddt_cond_fwd(enc2.to(DEV)[:, D:], temb.to(DEV), B, N, out)
    assert rel(out, sz) < 1e-6
    d = torch.randn(B * N, D, generator=g)
    sz.backward(d)
    denc, dtemb = torch.empty(B * N, D, device=DEV), torch.full((B, D), 7.0, device=DEV)
    ops.f32_ddt_cond_bwd(d.to(DEV), enc2.to(DEV)[:, D:], temb.to(DEV), B, N, denc, dtemb)
    assert rel(denc, er.grad) < 2e-6 and rel(dtemb, tr.grad) < 2e-6  # (dtemb is written, not accumulated)

    # LayerNorm + per-token modulation (rows_per_mod = 1) with the gated residual of the previous sub-layer, forward and backward
    M = B * N
    x, t = torch.randn(M, D, generator=g), torch.randn(M, D, generator=g)
    w_, b_ = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    tm = 0.3 * torch.randn(M, 3 * D, generator=g)  # [scale | shift | gate] rows per token
    xr, tr_, wr, br, tmr = (v.clone().requires_grad_(True) for v in (x, t, w_, b_, tm))
    x_in = xr + tmr[:, 2 * D :] * tr_
    y = F.layer_norm(x_in, (D,), wr, br, 1e-5) * (1 + tmr[:, :D]) + tmr[:, D : 2 * D]
    dy, dres = torch.randn(M, D, generator=g), torch.randn(M, D, generator=g)
    ((y * dy).sum() + (x_in * dres).sum()).backward()
    dev = lambda v: v.to(DEV)  # noqa: E731
    tmd = dev(tm)
    o, xo, mean, rstd = torch.empty(M, D, device=DEV), torch.empty(M, D, device=DEV), torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.f32_ln_modulate_fwd(dev(x), dev(w_), dev(b_), tmd[:, :D], tmd[:, D : 2 * D], 1, 1e-5, o, mean, rstd, t=dev(t), gate=tmd[:, 2 * D :],
                            x_out=xo)
    assert rel(o, y) < 2e-6 and rel(xo, x_in) < 1e-6
    dx, dt = torch.empty(M, D, device=DEV), torch.empty(M, D, device=DEV)
    dtm, dwb = torch.zeros(M, 3 * D, device=DEV), torch.empty(M, 2, D, device=DEV)
    ops.f32_ln_modulate_bwd(dev(dy), xo, dev(w_), dev(b_), tmd[:, :D], 1, mean, rstd, dev(dres), dx, dtm[:, :D], dtm[:, D : 2 * D], dwb,
                            gate_t=dev(t), gate=tmd[:, 2 * D :], dt=dt, dgate=dtm[:, 2 * D :])
    assert rel(dx, xr.grad) < 3e-6 and rel(dt, tr_.grad) < 3e-6 and rel(dtm, tmr.grad) < 3e-6
    assert rel(dwb.sum(0)[0], wr.grad) < 3e-6 and rel(dwb.sum(0)[1], br.grad) < 3e-6


def _model():
    from diffulab_amd import DDT

    cfg = oddt.DDTConfig(**KW)
    m = DDT(simple_ddt=True, **KW)
    m.load_state_dict(synth.dit_params(oddt.param_shapes(cfg), seed=91))
    m.set_precision("fp32")
    m = m.to(DEV)
    assert m.precision == "fp32" and type(m.engine).__name__ == "DDTEngineF32"
    return m


def test_ddt_fp32_training_step_against_reference_fixture_and_oracle(golden):
    g = {k: torch.as_tensor(v) for k, v in golden("ddt").items()}
    m = _model()
    B, H = 4, 16
    x, t, y = synth.normal("dd.x", (B, 4, H, H)), synth.uniform("dd.t", (B,), lo=0.05, hi=0.95), synth.integers("dd.y", (B,), 10)
    dy = synth.normal("dd.dy", (B, 4, H, H))
    m.train()
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV), p=0.0)["x"]
    assert rel(pred, g["pred"]) < TOL
    (pred * dy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    cfg = oddt.DDTConfig(**KW)
    Pr = {k: v.requires_grad_(True) for k, v in synth.dit_params(oddt.param_shapes(cfg), seed=91).items()}
    (oddt.ddt_forward(Pr, x, t, y, cfg) * dy).sum().backward()
    errs = {}
    for n, p in m.named_parameters():
        if "g_" + n in g:
            errs[n + " (fixture)"] = rel(p.grad, g["g_" + n])
        errs[n + " (oracle)"] = rel(p.grad, Pr[n].grad)
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print("DDT fp32 regime, largest per-tensor gradient errors:", top)
    assert top[0][1] < TOL, top
    g1 = m._flat_grad.clone()  # bit-reproducible
    m.zero_grad()
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV), p=0.0)["x"]
    (pred * dy.to(DEV)).sum().backward()
    assert torch.equal(g1, m._flat_grad)


def test_ddt_fp32_guided_sampling_against_reference_fixture_and_optimizer_steps(golden):
    from diffulab_amd import Diffuser
    from diffulab_amd.training import FusedAdamW

    g = {k: torch.as_tensor(v) for k, v in golden("ddt").items()}
    m = _model()
    y = synth.integers("dd.y", (4,), 10)
    m.eval()
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    out = d.generate({"x": synth.normal("dd.init", (4, 4, 16, 16)).to(DEV), "y": y.to(DEV)}, use_tqdm=False, guidance_scale=2.0)
    assert rel(out["x"], g["loop_x"]) < TOL
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
