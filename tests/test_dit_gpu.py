"""GPU parity of the whole denoising hot path (DiT forward/backward under the flow / DDPM heads, sampler loops)
through the reference-shaped plugin API (`diffulab_amd.Diffuser` / `MMDiT`), i.e. through the C ABI.

Expected values: (1) committed reference outputs (tests/golden/*.npz, produced by importing the real reference),
(2) the CPU oracle on seeded inputs.  The HIP path computes in bf16 with f32 accumulation, so tolerances are the
bf16 ones of SURVEY.md §8c: activations / gradients relative-L2 <= 2e-2 per tensor, loss <= 2e-3 relative.
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import diffusion as od  # noqa: E402
from oracle import dit as odit  # noqa: E402
from oracle import synth  # noqa: E402

DEV = "cuda"
SMALL = dict(input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4,
             patch_size=2, depth=2, n_classes=10, classifier_free=True)


def rel(a, b):
    a = a.detach().double().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().double().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b)).double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def build(cfg_kwargs, seed):
    from diffulab_amd import MMDiT

    m = MMDiT(simple_dit=True, **cfg_kwargs)
    P = synth.dit_params(odit.param_shapes(odit.DiTConfig(**cfg_kwargs)), seed=seed)
    m.load_state_dict(P)
    return m.to(DEV), P


def test_state_dict_contract_and_init():
    """key names/shapes == reference Appendix A; adaLN-zero init makes every block the identity (mmdit.py:737-745)."""
    from diffulab_amd import MMDiT

    m = MMDiT(simple_dit=True, **SMALL)
    shapes = odit.param_shapes(odit.DiTConfig(**SMALL))
    sd = m.state_dict()
    assert set(sd) == set(shapes) and all(tuple(sd[k].shape) == shapes[k] for k in shapes)
    assert float(m.layers[0].modulation.lin.weight.detach().abs().sum()) == 0.0
    assert float(m.last_layer.adaLN_modulation[1].weight.abs().sum()) == 0.0
    m = m.to(DEV)
    x = synth.normal("init.x", (4, 4, 16, 16)).to(DEV)
    t = synth.uniform("init.t", (4,)).to(DEV)
    y = synth.integers("init.y", (4,), 10).to(DEV)
    with torch.no_grad():
        out = m(x=x, timesteps=t, y=y)["x"]
    P = {k: v.detach().cpu().float() for k, v in m.state_dict().items()}
    ref = odit.dit_forward(P, x.cpu(), t.cpu(), y.cpu(), odit.DiTConfig(**SMALL))
    assert rel(out, ref) < 1e-2
    # after .to() the parameters are views of one flat arena and survive a state_dict round trip
    assert m._is_flat()
    m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})
    assert m._is_flat()


def test_small_model_against_reference_fixture(golden):
    """flow loss / prediction / every parameter gradient vs the REFERENCE's outputs (dit_small16.npz).  The per-tensor bound is 2e-2,
    not SURVEY 8(c)'s 1e-2: the REFERENCE ITSELF under bf16 autocast is 9.5e-3 (median) ... 2e-2 off its own fp32 gradients on these
    inputs (`dit_autocast.npz`, keys s16_*; tests/test_parity_bf16_gpu.py bounds the HIP error by 1.5 x those per tensor)."""
    from diffulab_amd import Diffuser

    g = golden("dit_small16")
    m, _ = build(SMALL, seed=5)
    B, H = 4, 16
    x0 = synth.normal("s16.x0", (B, 4, H, H))
    noise = synth.normal("s16.noise", (B, 4, H, H))
    t = synth.uniform("s16.t", (B,), lo=0.02, hi=0.98)
    y = synth.integers("s16.y", (B,), 10)
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=50)
    inputs = {"x": x0.to(DEV), "y": y.to(DEV), "p": 0.0}
    losses = d.compute_loss(inputs, timesteps=t, noise=noise.to(DEV))
    assert rel(inputs["x"], od.flow_add_noise(x0, t, noise)) < 1e-6  # model_inputs["x"] is overwritten with z_t
    loss = losses["loss"]
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) / float(g["loss"]) < 1e-3
    with torch.no_grad():
        pred = m(x=inputs["x"], timesteps=t.to(DEV), y=y.to(DEV))["x"]
    assert rel(pred, g["pred"]) < 1e-2
    worst = 0.0
    for name, p in m.named_parameters():
        r = rel(p.grad, g["g_" + name])
        worst = max(worst, r)
        assert r < 2e-2, (name, r)
    print("worst grad rel-L2", worst)
    # accumulation semantics: a second backward doubles the gradient; zero_grad() clears the arena
    g1 = m.conv_proj.weight.grad.clone()
    inputs = {"x": x0.to(DEV), "y": y.to(DEV), "p": 0.0}
    d.compute_loss(inputs, timesteps=t, noise=noise.to(DEV))["loss"].backward()
    assert rel(m.conv_proj.weight.grad, 2 * g1) < 1e-3
    m.zero_grad()
    assert float(m._flat_grad.abs().sum()) == 0.0


def test_ddpm_head_and_label_drop_against_oracle():
    from diffulab_amd import Diffuser

    m, P = build(SMALL, seed=5)
    cfg = odit.DiTConfig(**SMALL)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    B, H = 4, 16
    x0 = synth.normal("dd.x0", (B, 4, H, H))
    noise = synth.normal("dd.noise", (B, 4, H, H))
    y = synth.integers("dd.y", (B,), 10)
    ti = torch.tensor([3, 500, 999, 0], dtype=torch.int32)
    d = Diffuser(m, sampling_method="ddpm", model_type="gaussian_diffusion", n_steps=1000)
    loss = d.compute_loss({"x": x0.to(DEV), "y": y.to(DEV), "p": 0.0}, timesteps=ti, noise=noise.to(DEV))["loss"]
    loss.backward()
    T = od.GaussianTables(1000)
    ref = od.mse_loss(odit.dit_forward(Pr, od.ddpm_add_noise(T, x0, ti, noise), ti, y, cfg), noise)
    ref.backward()
    assert abs(loss.item() - ref.item()) / ref.item() < 1e-3
    for name, p in m.named_parameters():
        assert rel(p.grad, Pr[name].grad) < 2.5e-2, name
    # p = 1 drops every label (Appendix C.12): equals the oracle fed the null class
    with torch.no_grad():
        z = synth.normal("dd.z", (B, 4, H, H))
        tf = synth.uniform("dd.t", (B,))
        out = m(x=z.to(DEV), timesteps=tf.to(DEV), y=y.to(DEV), p=1.0)["x"]
        ref_un = odit.dit_forward(P, z, tf, torch.full_like(y, 10), cfg)
    assert rel(out, ref_un) < 1e-2


def test_sampler_loops_against_reference_fixture(golden, monkeypatch):
    from diffulab_amd import Diffuser

    g = golden("dit_small16")
    m, _ = build(SMALL, seed=5)
    m.eval()
    B, H = 4, 16
    y = synth.integers("s16.y", (B,), 10).to(DEV)
    x_init = synth.normal("s16.init", (B, 4, H, H))
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    out = d.generate({"x": x_init.to(DEV), "y": y}, use_tqdm=False, guidance_scale=2.0, return_intermediates=True)
    assert out["xt"].shape == (B, 5, 4, H, H)
    assert rel(out["x"], g["loop_euler_x"]) < 2e-2
    assert rel(out["estimated_x0"], g["loop_euler_x0"]) < 2e-2
    # DDPM, respaced to 5 steps, CFG 1.5, clamp: replay the reference's CPU generator stream
    # (per step: rand(B) for the p=1 label drop, then randn_like -- make_golden.gen_small16)
    torch.manual_seed(23)
    noises = []
    for _ in range(5):
        torch.rand(B)
        noises.append(torch.randn(B, 4, H, H))
    it = iter(noises)
    monkeypatch.setattr(torch, "randn_like", lambda ref, **kw: next(it).to(ref.device))
    gd = Diffuser(m, sampling_method="ddpm", model_type="gaussian_diffusion", n_steps=1000)
    gd.set_steps(5)
    with torch.no_grad():
        out = gd.generate({"x": x_init.to(DEV), "y": y}, use_tqdm=False, guidance_scale=1.5, clamp_x=True)
    assert rel(out["x"], g["loop_ddpm_x"]) < 3e-2


@pytest.mark.parametrize("family", ["dit", "ddt", "unet"])
def test_guided_step_as_one_paired_forward_equals_the_two_forwards(family, monkeypatch):
    """a guided sampler step makes two denoiser forwards (p = 0, then p = 1: flow.py:256-259); the class-conditional
    denoisers run them as ONE forward over [x ; x] with the label rows [y ; dropped] (DL_CFG_PAIR, default on).  Same values -- every
    kernel of the path is per row / per sample -- and the same device RNG stream (the p = 1 forward's rand(B) is still drawn), checked
    on a stochastic sampler: Euler-Maruyama for the flow, DDPM for the UNet."""
    import diffulab_amd as da
    from diffulab_amd import Diffuser

    torch.manual_seed(0)
    small = dict(input_channels=4, output_channels=4, inner_dim=128, num_heads=2, mlp_ratio=4, patch_size=2)
    if family == "dit":
        m, H = da.MMDiT(simple_dit=True, embedding_dim=64, depth=2, n_classes=10, classifier_free=True, **small), 16
    elif family == "ddt":
        m, H = da.DDT(simple_ddt=True, encoder_depth=2, decoder_depth=1, n_classes=10, classifier_free=True, **small), 16
    else:
        m, H = da.UNetModel(image_size=[16, 16], in_channels=4, model_channels=32, out_channels=4, num_res_blocks=1, attention_resolutions=[2],
                            channel_mult="1, 2", num_heads=2, use_scale_shift_norm=True, resblock_updown=True, n_classes=10,
                            classifier_free=True), 16
    with torch.no_grad():
        for q in m.parameters():  # (zero-initialised output layers would make every prediction 0)
            if float(q.abs().sum()) == 0:
                q.normal_(0, 0.05)
    m = m.to(DEV).eval()
    B = 5
    y = torch.randint(0, 10, (B,), device=DEV)
    x_init = torch.randn(B, 4, H, H, device=DEV)
    if family == "unet":
        d = Diffuser(m, sampling_method="ddpm", model_type="gaussian_diffusion", n_steps=1000)
        d.set_steps(4)
    else:
        d = Diffuser(m, sampling_method="euler_maruyama", model_type="rectified_flow", n_steps=4)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DL_CFG_PAIR", mode)
        torch.manual_seed(11)
        outs[mode] = d.generate({"x": x_init.clone(), "y": y}, use_tqdm=False, guidance_scale=2.5)["x"]
        after = torch.rand(3, device=DEV)
        outs[mode + "rng"] = after
    assert torch.isfinite(outs["1"]).all() and float(outs["1"].abs().mean()) > 1e-3
    assert torch.equal(outs["1rng"], outs["0rng"])  # both modes consumed the same number of device draws
    # (bf16 kernels: a launch of 2 B rows may take another tile / split-K variant than one of B rows -- another f32 summation order
    # before the same bf16 rounding; the UNet's low-resolution convolutions do, and four stochastic steps carry the ulps along)
    tol = 2e-2 if family == "unet" else 2e-3
    assert rel(outs["1"], outs["0"]) < tol, rel(outs["1"], outs["0"])
    # and the pair itself, directly
    t = torch.full((B,), 0.4 if family != "unet" else 7, device=DEV, dtype=torch.float32 if family != "unet" else torch.int32)
    monkeypatch.setenv("DL_CFG_PAIR", "1")
    with torch.no_grad():
        pair = m.forward_cfg_pair(t, x=x_init, y=y)
        sep = (m(x=x_init, timesteps=t, y=y, p=0)["x"], m(x=x_init, timesteps=t, y=y, p=1)["x"])
    assert pair is not None and rel(pair[0], sep[0]) < tol and rel(pair[1], sep[1]) < tol and rel(pair[0], pair[1]) > 3e-2
    monkeypatch.setenv("DL_CFG_PAIR", "0")
    assert m.forward_cfg_pair(t, x=x_init, y=y) is None


@pytest.mark.timeout(900)
def test_dit_s2_against_reference_fixture(golden):
    """BASELINE config dims (DiT-S/2, 4x32x32 latents, 256 tokens): loss, prediction and gradient norms vs reference.  Per-tensor
    bounds 2e-2 (prediction) / 3e-2 (gradients) instead of SURVEY 8(c)'s 1e-2: on these inputs the REFERENCE under bf16 autocast loses
    1.56e-2 (median) / 2.45e-2 (max) against its own fp32 gradients, 149 of 154 tensors above 1e-2 (`dit_autocast.npz`, keys s2_*);
    tests/test_parity_bf16_gpu.py::test_hip_error_within_the_reference_s_own_autocast_error is the per-tensor statement."""
    from diffulab_amd import Diffuser

    g = golden("dit_s2")
    cfgk = dict(input_channels=4, output_channels=4, inner_dim=384, embedding_dim=384, num_heads=6, mlp_ratio=4,
                patch_size=2, depth=12, n_classes=1000, classifier_free=True)
    m, _ = build(cfgk, seed=7)
    B = 2
    x0 = synth.normal("s2.x0", (B, 4, 32, 32))
    noise = synth.normal("s2.noise", (B, 4, 32, 32))
    t = synth.uniform("s2.t", (B,), lo=0.05, hi=0.95)
    y = synth.integers("s2.y", (B,), 1000)
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=50)
    inputs = {"x": x0.to(DEV), "y": y.to(DEV), "p": 0.0}
    loss = d.compute_loss(inputs, timesteps=t, noise=noise.to(DEV))["loss"]
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) / float(g["loss"]) < 1e-3
    with torch.no_grad():
        pred = m(x=inputs["x"], timesteps=t.to(DEV), y=y.to(DEV))["x"]
    assert rel(pred, g["pred"]) < 2e-2
    norms = dict(zip(g["grad_names"].tolist(), g["grad_norms"].tolist()))
    params = dict(m.named_parameters())
    bad = []
    for n, ref in norms.items():
        got = params[n].grad.double().norm().item()
        if abs(got - ref) > 3e-2 * max(ref, 1e-12):
            bad.append((n, got, ref))
    assert not bad, bad[:8]
    errs = {}
    for k in g:
        if k.startswith("g_"):
            errs[k] = rel(params[k[2:]].grad, g[k])
        elif k.startswith("gs_"):
            # strided 512-entry sample of a big tensor: measure the error against the tensor's RMS scale
            # (sample entries can all be tiny: layers.10.mlp_input.2 samples 4 near-dead hidden units)
            gr = params[k[3:]].grad
            smp = gr.flatten()[:: max(1, gr.numel() // 512)][:512].double().cpu()
            ref_s = torch.from_numpy(g[k]).double()
            scale = norms[k[3:]] * (ref_s.numel() / gr.numel()) ** 0.5
            errs[k] = ((smp - ref_s).norm() / max(scale, ref_s.norm().item())).item()
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:12]
    print("largest per-tensor gradient errors:", top)
    assert top[0][1] < 3e-2, top


@pytest.mark.timeout(900)
def test_loss_curve_against_reference(golden):
    """20 AdamW steps of DiT-S/2 on fixed synthetic data: the loss curve of the HIP bf16 regime vs the
    REFERENCE's own fp32 curve, bounded by what the REFERENCE ITSELF loses on the same loop under ``torch.autocast("cpu",
    bfloat16)`` (`loss_curve_autocast.npz`, generated by make_golden.py curve_autocast from the imported reference: max 1.84e-3,
    mean 4.7e-4 against its fp32 curve): the maximum within 1.5 x the reference's maximum, the mean within 1.5 x the reference's mean,
    and per step ``max(1e-3, 1.5 x`` the reference's largest autocast error within two steps of that step ``)``.  (Point by point the
    two error sequences are rounding noise of one scale and different sign patterns -- measured: step 1 HIP 1.14e-3 / reference
    1.84e-3, step 3 HIP 1.53e-3 / reference 8.7e-4 -- so the per-step bound uses the reference's local envelope, not its value at
    the same index; HIP max 1.5e-3 / mean 4.0e-4 against the reference's 1.84e-3 / 4.7e-4.)  north_star's 1e-4 is an fp32 target -- tests/test_fp32_gpu.py holds the fp32 regime
    to it.  bench.py reports the same figures as config.loss_curve_rel_err(.reference_under_autocast)."""
    from diffulab_amd import Diffuser
    from diffulab_amd.training import FusedAdamW

    g = golden("loss_curve")
    cfgk = dict(input_channels=4, output_channels=4, inner_dim=384, embedding_dim=384, num_heads=6, mlp_ratio=4,
                patch_size=2, depth=12, n_classes=1000, classifier_free=True)
    m, _ = build(cfgk, seed=7)
    opt = FusedAdamW(m.parameters(), lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8)
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=50)
    B = 4
    x0 = synth.normal("curve.x0", (B, 4, 32, 32)).to(DEV)
    y = synth.integers("curve.y", (B,), 1000).to(DEV)
    got = []
    for s in range(len(g["losses"])):
        noise = synth.normal(f"curve.noise{s}", (B, 4, 32, 32)).to(DEV)
        t = synth.uniform(f"curve.t{s}", (B,), lo=0.02, hi=0.98)
        opt.zero_grad()
        loss = d.compute_loss({"x": x0, "y": y, "p": 0.0}, timesteps=t, noise=noise)["loss"]
        loss.backward()
        opt.step()
        got.append(loss.item())
    got, ref = np.array(got), g["losses"]
    err = np.abs(got - ref) / ref
    ra = golden("loss_curve_autocast")["rel_err_vs_fp32"]
    env = np.array([ra[max(0, s - 2) : s + 3].max() for s in range(len(ra))])
    bound = np.maximum(1e-3, 1.5 * env)
    print("loss curve rel err per step:", err, "\nreference under autocast:", ra, "\nworst err/bound:", (err / bound).max())
    assert len(ref) == 20 and len(ra) == 20
    assert err.max() <= 1.5 * ra.max(), (err.max(), ra.max())
    assert err.mean() <= 1.5 * ra.mean(), (err.mean(), ra.mean())
    assert (err <= bound).all(), (err, bound)


def test_cifar_dit_dims_against_oracle():
    """configs/model/dit.yaml dims (BASELINE.json configs[1]: RGB 32x32, patch 2 -> 12 features per patch, inner 512, 8 heads,
    not classifier-free) at depth 2: loss and every gradient against the CPU oracle -- covers the ragged patch width (12 -> K
    padded to 64) and D = 512."""
    from diffulab_amd import Diffuser

    kw = dict(input_channels=3, output_channels=3, inner_dim=512, embedding_dim=512, num_heads=8, mlp_ratio=4, patch_size=2,
              depth=2, n_classes=10, classifier_free=False)
    m, P = build(kw, seed=11)
    cfg = odit.DiTConfig(**kw)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    B = 4
    x0, noise = synth.normal("cf.x0", (B, 3, 32, 32)), synth.normal("cf.noise", (B, 3, 32, 32))
    y, t = synth.integers("cf.y", (B,), 10), synth.uniform("cf.t", (B,), lo=0.05, hi=0.95)
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=100, extra_args={"logits_normal": True})
    loss = d.compute_loss({"x": x0.to(DEV), "y": y.to(DEV), "p": 0.0}, timesteps=t, noise=noise.to(DEV))["loss"]
    loss.backward()
    ref = od.flow_loss(odit.dit_forward(Pr, od.flow_add_noise(x0, t, noise), t, y, cfg), x0, noise)
    ref.backward()
    assert abs(loss.item() - ref.item()) / ref.item() < 1e-3
    for name, p in m.named_parameters():
        assert rel(p.grad, Pr[name].grad) < 2.5e-2, name


def test_hipgraph_inference_matches_eager_and_tracks_parameter_updates(monkeypatch):
    """the inference forward is replayed from a captured hipGraph: same bits as the eager launch sequence, and a parameter
    update between two replays is seen (the bf16 weight shadows are refreshed outside the graph)"""
    m, _ = build(SMALL, seed=5)
    m.eval()
    x = synth.normal("hg.x", (4, 4, 16, 16)).to(DEV)
    t = synth.uniform("hg.t", (4,)).to(DEV)
    y = synth.integers("hg.y", (4,), 10).to(DEV)
    with torch.no_grad():
        a = m(x=x, timesteps=t, y=y)["x"]  # eager run + capture
        b = m(x=x, timesteps=t, y=y)["x"]  # replay
        assert m._graphs and all(v is not False for v in m._graphs.values()), "capture failed"
        monkeypatch.setenv("DL_HIPGRAPH", "0")
        c = m(x=x, timesteps=t, y=y)["x"]
        assert torch.equal(a, b) and torch.equal(a, c)
        for p in m.parameters():
            p.mul_(1.02)
        e = m(x=x, timesteps=t, y=y)["x"]  # eager, new weights
        monkeypatch.delenv("DL_HIPGRAPH")
        d = m(x=x, timesteps=t, y=y)["x"]  # replay, new weights
        assert torch.equal(d, e) and not torch.equal(a, d)
        # other inputs through the same graph
        x2 = synth.normal("hg.x2", (4, 4, 16, 16)).to(DEV)
        f = m(x=x2, timesteps=t, y=y)["x"]
        monkeypatch.setenv("DL_HIPGRAPH", "0")
        assert torch.equal(f, m(x=x2, timesteps=t, y=y)["x"])


def test_pipelined_attention_inside_the_captured_inference_forward(monkeypatch):
    """round 6: at sampler batch sizes (>= 3 (sample, head) items per CU) the inference forward runs the persistent, pipelined
    attention forward (attn_fwd_qkn_pipe_k, its inference form: no q / k / rrms outputs).  DiT-S/2 dims at depth 2, B = 160:
    eager, captured-graph replay and the chain form of the kernel (lab switch) give the same bits."""
    from diffulab_amd import ops

    cfgk = dict(input_channels=4, output_channels=4, inner_dim=384, embedding_dim=384, num_heads=6, mlp_ratio=4, patch_size=2, depth=2,
                n_classes=1000, classifier_free=True)
    m, _ = build(cfgk, seed=3)
    m.eval()
    B = 160
    x = synth.normal("pg.x", (B, 4, 32, 32)).to(DEV)
    t = synth.uniform("pg.t", (B,)).to(DEV)
    y = synth.integers("pg.y", (B,), 1000).to(DEV)
    with torch.no_grad():
        a = m(x=x, timesteps=t, y=y)["x"]  # eager run + capture
        b = m(x=x, timesteps=t, y=y)["x"]  # replay
        assert m._graphs and all(v is not False for v in m._graphs.values()), "capture failed"
        monkeypatch.setenv("DL_HIPGRAPH", "0")
        ops.lib().cdll.dl_lab_set_attn_pipe(0)
        try:
            c = m(x=x, timesteps=t, y=y)["x"]  # eager, chain form of the attention
        finally:
            ops.lib().cdll.dl_lab_set_attn_pipe(1)
        assert torch.equal(a, b) and torch.equal(a, c) and bool(torch.isfinite(a).all())


def test_dit_on_1024_tokens_against_oracle():
    """a 64x64 latent grid at patch 2 = 1024 tokens per image (e.g. 512-pixel images through an f8 VAE): the chunked attention
    kernels inside the full forward / backward, loss and gradients against the CPU oracle"""
    from diffulab_amd import Diffuser

    m, P = build(SMALL, seed=5)
    cfg = odit.DiTConfig(**SMALL)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    B, H = 2, 64
    x0, noise = synth.normal("lg.x0", (B, 4, H, H)), synth.normal("lg.noise", (B, 4, H, H))
    y, t = synth.integers("lg.y", (B,), 10), synth.uniform("lg.t", (B,), lo=0.05, hi=0.95)
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    loss = d.compute_loss({"x": x0.to(DEV), "y": y.to(DEV), "p": 0.0}, timesteps=t, noise=noise.to(DEV))["loss"]
    loss.backward()
    ref = od.flow_loss(odit.dit_forward(Pr, od.flow_add_noise(x0, t, noise), t, y, cfg), x0, noise)
    ref.backward()
    assert abs(loss.item() - ref.item()) / ref.item() < 1e-3
    for name, p in m.named_parameters():
        assert rel(p.grad, Pr[name].grad) < 2.5e-2, name


@pytest.mark.parametrize("B,H", [(1, 16), (3, 16), (5, 32), (7, 16)])
def test_ragged_batch_sizes_against_oracle(B, H):
    """batch sizes that are not multiples of anything (a last partial batch, single-image sampling): forward and every gradient"""
    from diffulab_amd import MMDiT

    kw = dict(input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4, patch_size=2, depth=2,
              n_classes=10, classifier_free=True)
    cfg = odit.DiTConfig(**kw)
    P = synth.dit_params(odit.param_shapes(cfg), seed=5)
    m = MMDiT(simple_dit=True, **kw)
    m.load_state_dict(P)
    m = m.to(DEV)
    x, t, y = synth.normal(f"e.x{B}", (B, 4, H, H)), synth.uniform(f"e.t{B}", (B,), lo=0.1, hi=0.9), synth.integers(f"e.y{B}", (B,), 10)
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV))["x"]
    pred.square().mean().backward()
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    po = odit.dit_forward(Pr, x, t, y, cfg)
    po.square().mean().backward()
    assert rel(pred, po) < 1.5e-2
    assert max(rel(p.grad, Pr[n].grad) for n, p in m.named_parameters()) < 4e-2


def test_non_square_latents_and_x_context_against_oracle():
    """16 x 32 latents (8 x 16 token grid: the row / column RoPE axes differ) with an extra conditioning image concatenated along
    the channels (`x_context`, mmdit.py:918-919): forward and every gradient"""
    from diffulab_amd import MMDiT

    kw = dict(input_channels=6, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4, patch_size=2, depth=2,
              n_classes=10, classifier_free=True)
    cfg = odit.DiTConfig(**kw)
    P = synth.dit_params(odit.param_shapes(cfg), seed=9)
    m = MMDiT(simple_dit=True, **kw)
    m.load_state_dict(P)
    m = m.to(DEV)
    B, H, W = 4, 16, 32
    x, xc = synth.normal("ns.x", (B, 4, H, W)), synth.normal("ns.xc", (B, 2, H, W))
    t, y = synth.uniform("ns.t", (B,), lo=0.1, hi=0.9), synth.integers("ns.y", (B,), 10)
    dy = synth.normal("ns.dy", (B, 4, H, W))
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV), x_context=xc.to(DEV))["x"]
    assert pred.shape == (B, 4, H, W)
    (pred * dy.to(DEV)).sum().backward()
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    po = odit.dit_forward(Pr, torch.cat((x, xc), dim=1), t, y, cfg)
    (po * dy).sum().backward()
    assert rel(pred, po) < 1.5e-2
    assert max(rel(p.grad, Pr[n].grad) for n, p in m.named_parameters()) < 4e-2


def test_native_block_driver_equals_the_python_issued_sequence(monkeypatch):
    """dl_dit_block_fwd / dl_dit_block_bwd (csrc/block.hip: one C call per block and direction) issue the same kernels in the same
    order as the Python launch sequence: predictions are bit-identical (train and inference, incl. the hipGraph replay), gradients
    agree up to the order of the f32 atomics of the split-R weight-gradient kernels; RePA-style feature gradients enter too."""
    from diffulab_amd import Diffuser

    B, H = 4, 16
    x0, noise = synth.normal("nb.x0", (B, 4, H, H)).to(DEV), synth.normal("nb.noise", (B, 4, H, H)).to(DEV)
    t, y = synth.uniform("nb.t", (B,), lo=0.05, hi=0.95), synth.integers("nb.y", (B,), 10).to(DEV)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DL_NATIVE_BLOCK", mode)
        m, _ = build(SMALL, seed=5)
        d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
        loss = d.compute_loss({"x": x0.clone(), "y": y, "p": 0.0}, timesteps=t, noise=noise)["loss"]
        # a feature gradient on block 0's output, as RepaLoss injects it (engine.backward dfeats)
        eng = m.engine
        feat = eng.feature(0)
        loss.backward()
        g_plain = m._flat_grad.clone()
        m.zero_grad()
        eng.forward(od.flow_add_noise(x0.cpu(), t, noise.cpu()).to(DEV), t.to(DEV), y, train=True)
        eng.backward(torch.ones(B, 4, H, H, device=DEV) * 1e-3, {0: torch.full_like(feat, 1e-3)})
        torch.cuda.synchronize()
        m.eval()
        with torch.no_grad():
            p1 = m(x=x0, timesteps=t.to(DEV), y=y)["x"].clone()
            p2 = m(x=x0, timesteps=t.to(DEV), y=y)["x"].clone()  # hipGraph replay
        res[mode] = (loss.item(), g_plain, m._flat_grad.clone(), p1, p2)
    assert res["0"][0] == res["1"][0]
    assert torch.equal(res["0"][3], res["1"][3]) and torch.equal(res["1"][3], res["1"][4])
    assert rel(res["1"][1], res["0"][1]) < 1e-4 and rel(res["1"][2], res["0"][2]) < 1e-4


@pytest.mark.timeout(900)
def test_qk_norm_on_load_path_native_equals_python_and_matches_the_row_kernel_path(monkeypatch):
    """round 4 (dl_gemm_nt_ssq + dl_attn_fwd_qkn, the shipped default on the row-complete path from 22 samples of 256 tokens): the
    native block driver and the Python-issued sequence run the same kernels (bit-identical loss and predictions), and against the
    qk_norm_rope_fwd row-kernel path (DL_QKN_ON_LOAD=0) loss, prediction and gradients agree to what a bf16 ulp on a few q / k
    elements per thousand allows; against the oracle the new path holds the same bounds as the old one"""
    from diffulab_amd import Diffuser

    kw = dict(input_channels=4, output_channels=4, inner_dim=384, embedding_dim=384, num_heads=6, mlp_ratio=4, patch_size=2, depth=2,
              n_classes=10, classifier_free=True)
    B = 24
    x0, noise = synth.normal("qo.x0", (B, 4, 32, 32)), synth.normal("qo.noise", (B, 4, 32, 32))
    t, y = synth.uniform("qo.t", (B,), lo=0.05, hi=0.95), synth.integers("qo.y", (B,), 10)
    res = {}
    for native, qkn in (("1", "1"), ("0", "1"), ("1", "0")):
        monkeypatch.setenv("DL_NATIVE_BLOCK", native)
        monkeypatch.setenv("DL_QKN_ON_LOAD", qkn)
        m, P = build(kw, seed=3)
        d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
        loss = d.compute_loss({"x": x0.to(DEV), "y": y.to(DEV), "p": 0.0}, timesteps=t, noise=noise.to(DEV))["loss"]
        loss.backward()
        assert ("ssq_all" in m.engine.ws) == (qkn == "1")
        m.eval()
        with torch.no_grad():
            pred = m(x=x0.to(DEV), timesteps=t.to(DEV), y=y.to(DEV))["x"].clone()
        res[(native, qkn)] = (loss.item(), m._flat_grad.clone(), pred)
    a, b, c = res[("1", "1")], res[("0", "1")], res[("1", "0")]
    assert a[0] == b[0] and torch.equal(a[2], b[2]) and rel(a[1], b[1]) < 1e-4
    assert abs(a[0] - c[0]) / c[0] < 1e-3 and rel(a[2], c[2]) < 5e-3 and rel(a[1], c[1]) < 1e-2
    cfg = odit.DiTConfig(**kw)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = od.flow_loss(odit.dit_forward(Pr, od.flow_add_noise(x0, t, noise), t, y, cfg), x0, noise)
    ref.backward()
    assert abs(a[0] - ref.item()) / ref.item() < 1e-3
    lay = m.engine.layout
    for name in Pr:
        assert rel(lay.view(a[1], name), Pr[name].grad) < 2.5e-2, name


def test_a_replaced_parameter_object_is_picked_up_by_the_next_forward():
    """The per-step flat-arena check walks a cached owner list (round 5: `named_parameters()` over the module tree cost milliseconds per
    step on the UNet).  A parameter OBJECT that is replaced in its module -- `load_state_dict(assign=True)`, a re-registered
    nn.Parameter -- is no longer a view of the arena: the check must see it (`is not`), re-flatten and compute with the new values;
    an in-place write through the old object is seen through the version counters as before."""
    m, P = build(SMALL, seed=5)
    cfg = odit.DiTConfig(**SMALL)
    B = 2
    x, t, y = synth.normal("rp.x", (B, 4, 16, 16)), synth.uniform("rp.t", (B,), lo=0.05, hi=0.95), synth.integers("rp.y", (B,), SMALL["n_classes"])
    m.eval()
    with torch.no_grad():
        out0 = m(x=x.cuda(), timesteps=t.cuda(), y=y.cuda(), p=0.0)["x"].clone()
        assert rel(out0, odit.dit_forward(P, x, t, y, cfg)) < 1.5e-2
        name = "layers.0.attention.proj_out.weight"
        new = synth.normal("rp.w", tuple(P[name].shape), std=P[name].shape[1] ** -0.5)
        m.layers[0].attention.proj_out.weight = torch.nn.Parameter(new.cuda())  # a NEW object in the module
        out1 = m(x=x.cuda(), timesteps=t.cuda(), y=y.cuda(), p=0.0)["x"].clone()
        P1 = dict(P, **{name: new})
        assert rel(out1, odit.dit_forward(P1, x, t, y, cfg)) < 1.5e-2 and rel(out1, out0) > 1e-2
        assert m._is_flat()  # re-flattened: the new object is a view of the arena again
        m.layers[0].attention.proj_out.weight.mul_(0.5)  # in place, through the (new) object
        out2 = m(x=x.cuda(), timesteps=t.cuda(), y=y.cuda(), p=0.0)["x"].clone()
        P2 = dict(P1, **{name: 0.5 * new})
        assert rel(out2, odit.dit_forward(P2, x, t, y, cfg)) < 1.5e-2
