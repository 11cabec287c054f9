"""Pin the CPU oracle against every golden vector generated from the real reference
(tests/golden/make_golden.py).  CPU only; this is what lets the GPU parity tests trust the oracle.

Tolerances: integer / index data bit-exact; fp64 tables bit-exact (same torch ops in the same order);
fp32 tensors <= 2e-6 relative-L2 (different-but-equivalent op order, e.g. explicit softmax vs SDPA).
"""

import os

import numpy as np
import pytest
import torch

from oracle import diffusion as od
from oracle import dit as odit
from oracle import synth

SMALL = odit.DiTConfig(input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2,
                       mlp_ratio=4, patch_size=2, depth=2, n_classes=10, classifier_free=True)
S2 = odit.DiTConfig()


def rel(a, b):
    a = (a.detach() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))).double()
    b = (b.detach() if isinstance(b, torch.Tensor) else torch.as_tensor(np.asarray(b))).double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def t2n(x):
    return x.detach().numpy()


# ------------------------------------------------------------------ schedules / draws (bit-exact)
def test_flow_timesteps(golden):
    g = golden("schedules")
    for n in (4, 50, 100):
        assert np.array_equal(np.array(od.flow_timesteps(n)), g[f"flow_ts_n{n}"])
        for sh in (4.63, 6.93):
            assert np.array_equal(np.array(od.flow_timesteps(n, sh)), g[f"flow_ts_n{n}_shift{sh}"])
            # constructor-time shift never reaches the sampling grid in the reference (flow.py:72-81 vs :118)
            assert np.array_equal(np.array(od.flow_timesteps(n)), g[f"flow_ts_ctor_n{n}_shift{sh}"])


def test_known_answers_survey_appendix_b():
    assert od.flow_timesteps(4) == [1.0, 0.75, 0.5, 0.25, 0.0]
    assert od.flow_timesteps(4, 4.63) == [1.0, 0.9328408327736736, 0.822380106571936, 0.6068152031454783, 0.0]
    torch.manual_seed(0)
    assert od.flow_draw_timesteps(4).tolist() == [0.49625658988952637, 0.7682217955589294, 0.08847743272781372,
                                                  0.13203048706054688]
    torch.manual_seed(0)
    assert od.ddpm_draw_timesteps(4, 1000).tolist() == [44, 239, 933, 760]
    T = od.GaussianTables()
    assert T.alphas_bar[-1].item() == 4.0358297653756754e-05
    assert T.posterior_variance[1].item() == 5.4531876613021935e-05
    assert T.posterior_log_variance_clipped[0].item() == -9.81672513529567
    assert sorted(od.space_timesteps(1000, 10))[:3] == [0, 111, 222]
    e = odit.timestep_embedding(torch.tensor([0.5]), 8)[0]
    assert np.allclose(e.numpy(), [0.87758255, 0.99875027, 0.9999875, 0.99999988, 0.47942555, 0.04997917,
                                   0.00499998, 0.0005], atol=1e-7)


def test_draw_timesteps_bit_exact(golden):
    g = golden("schedules")
    for seed in (0, 1, 2):
        for B in (4, 64):
            for tag, kw in (("uniform", {}), ("logit", {"logits_normal": True}),
                            ("logit_shift", {"logits_normal": True, "shift": 4.63}), ("xpred", {"x_prediction": True})):
                torch.manual_seed(seed)
                assert np.array_equal(t2n(od.flow_draw_timesteps(B, **kw)), g[f"draw_flow_{tag}_s{seed}_b{B}"]), tag
            torch.manual_seed(seed)
            got = od.ddpm_draw_timesteps(B, 1000)
            assert got.dtype == torch.int32 and np.array_equal(t2n(got), g[f"draw_ddpm_s{seed}_b{B}"])


def test_gaussian_tables_bit_exact(golden):
    g = golden("schedules")
    for sched in ("linear", "cosine"):
        T = od.GaussianTables(1000, schedule=sched)
        for nm in ("betas", "alphas_bar", "sqrt_alphas_bar", "alphas_bar_prev", "posterior_variance",
                   "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
            assert np.array_equal(t2n(getattr(T, nm)), g[f"gd_{sched}_{nm}"]), (sched, nm)


def test_respacing_bit_exact(golden):
    g = golden("schedules")
    for n in (50, 100, 250):
        T = od.GaussianTables(1000, n_steps=n)
        assert np.array_equal(np.array(T.timestep_map), g[f"respace_{n}_map"])
        assert np.array_equal(t2n(T.betas), g[f"respace_{n}_betas"])
        assert np.array_equal(t2n(T.posterior_variance), g[f"respace_{n}_postvar"])
    T = od.GaussianTables(1000, n_steps=30, section_counts="10,10,10")
    assert np.array_equal(np.array(T.timestep_map), g["respace_sections_map"])
    assert np.array_equal(np.array(sorted(od.space_timesteps(1000, 10))), g["space_1000_10"])
    assert int(g["space_ddim_raises"]) == 1
    with pytest.raises(ValueError):
        od.space_timesteps(1000, 10, ddim=True)
    assert np.array_equal(np.array(sorted(od.space_timesteps(1000, 1000, ddim=True))), g["space_ddim_full"])


# ------------------------------------------------------------------ primitives
def test_embeddings_and_rope(golden):
    g = golden("prims")
    t = synth.uniform("prims.t", (8,), lo=0.0, hi=1.0)
    ti = torch.tensor([0, 1, 17, 500, 999], dtype=torch.int32)
    assert np.array_equal(t2n(odit.timestep_embedding(t, 256)), g["temb_f"])
    assert np.array_equal(t2n(odit.timestep_embedding(ti, 128)), g["temb_i"])
    assert np.array_equal(t2n(odit.timestep_embedding(t, 9)), g["temb_odd"])
    c, s = odit.rope_tables(16, 16, [32, 32], 10_000)
    assert np.array_equal(t2n(c), g["rope_cos_16x16"]) and np.array_equal(t2n(s), g["rope_sin_16x16"])
    c, s = odit.rope_tables(3, 5, [8, 24], 2000)
    assert np.array_equal(t2n(c), g["rope_cos_3x5"]) and np.array_equal(t2n(s), g["rope_sin_3x5"])


# ------------------------------------------------------------------ DiT block fwd+bwd
def test_dit_block_fwd_bwd(golden):
    g = golden("dit_block")
    cfg = SMALL
    P = {k: v.requires_grad_(True) for k, v in synth.dit_params(odit.param_shapes(cfg), seed=3).items()}
    B, gh, gw = 2, 4, 4
    x = synth.normal("blk.x", (B, gh * gw, cfg.inner_dim)).requires_grad_(True)
    emb = synth.normal("blk.emb", (B, cfg.embedding_dim)).requires_grad_(True)
    cos, sin = odit.rope_tables(gh, gw, cfg.rope_axes_dim, cfg.rope_base)
    taps = {}
    y = odit.dit_block(P, "layers.1.", x, emb, cos, sin, cfg, taps)
    (y * synth.normal("blk.dy", tuple(y.shape))).sum().backward()
    assert rel(y, g["y"]) < 2e-6
    assert rel(taps["attn_proj"], g["tap_attn_proj"]) < 2e-6
    assert rel(taps["mlp_hidden"], g["tap_mlp_hidden"]) < 2e-6
    assert rel(x.grad, g["dx"]) < 2e-6 and rel(emb.grad, g["demb"]) < 2e-6
    for k in g:
        if k.startswith("g_"):
            assert rel(P["layers.1." + k[2:]].grad, g[k]) < 3e-6, k


# ------------------------------------------------------------------ full models
def _flow_loss(P, cfg, x0, t, y, noise):
    z = od.flow_add_noise(x0, t, noise)
    pred = odit.dit_forward(P, z, t, y, cfg)
    return pred, od.flow_loss(pred, x0, noise)


def test_small_model_flow_and_ddpm(golden):
    g = golden("dit_small")
    cfg = SMALL
    P = {k: v.requires_grad_(True) for k, v in synth.dit_params(odit.param_shapes(cfg), seed=5).items()}
    B, H = 3, 8
    x0 = synth.normal("small.x0", (B, 4, H, H))
    noise = synth.normal("small.noise", (B, 4, H, H))
    t = synth.uniform("small.t", (B,), lo=0.02, hi=0.98)
    y = synth.integers("small.y", (B,), cfg.n_classes)
    pred, loss = _flow_loss(P, cfg, x0, t, y, noise)
    loss.backward()
    assert rel(pred, g["pred"]) < 2e-6
    assert abs(loss.item() - float(g["loss"])) / float(g["loss"]) < 1e-6
    for k in g:
        if k.startswith("g_"):
            assert rel(P[k[2:]].grad, g[k]) < 5e-6, k
    # all labels dropped == p=1 in the reference (Appendix C.12)
    y_un = odit.drop_labels(y, 1.0, cfg.n_classes)
    assert (y_un == cfg.n_classes).all()
    with torch.no_grad():
        pu = odit.dit_forward(P, od.flow_add_noise(x0, t, noise), t, y_un, cfg)
    assert rel(pu, g["pred_uncond"]) < 2e-6
    # DDPM head: int32 indices fed unscaled to the timestep embedding
    for p in P.values():
        p.grad = None
    T = od.GaussianTables(1000)
    ti = torch.tensor([3, 500, 999], dtype=torch.int32)
    xt = od.ddpm_add_noise(T, x0, ti, noise)
    assert rel(xt, g["ddpm_xt"]) < 1e-7
    l2 = od.mse_loss(odit.dit_forward(P, xt, ti, y, cfg), noise)
    l2.backward()
    assert abs(l2.item() - float(g["ddpm_loss"])) / float(g["ddpm_loss"]) < 1e-6
    assert rel(P["conv_proj.weight"].grad, g["ddpm_g_conv_proj.weight"]) < 5e-6


@pytest.mark.timeout(900)
@pytest.mark.parametrize("tag,cfg,seed,B,H,lo", [("s16", SMALL, 5, 4, 16, 0.02), ("s2", S2, 7, 2, 32, 0.05)])
def test_bf16_autocast_leg_against_the_reference_under_autocast(golden, tag, cfg, seed, B, H, lo):
    """the bf16 YARDSTICK is pinned to the reference: `dit_autocast.npz` holds what the imported reference loses under
    ``torch.autocast("cpu", bfloat16)`` against its own fp32 run -- per parameter, on the inputs of the dit_small16 / dit_s2
    fixtures (s2: median 1.6e-2, 149 of 154 tensors above SURVEY's 1e-2).  ``odit.bf16_autocast()`` -- the leg the bounds of
    tests/test_parity_bf16_gpu.py were relative to until round 5 -- is the same regime but not the same rounding sequence (explicit
    softmax instead of SDPA, f32 modulation arithmetic): measured here, it loses 0.57 ... 1.0x of what the reference loses per
    tensor (median 0.78 / 0.84), i.e. it was the STRICTER yardstick, never a looser one.  Asserted: per tensor within [0.45, 1.15]
    of the reference's error, median within [0.65, 1.05], loss shift <= max(2x the reference's, 3e-4)."""
    g = golden("dit_autocast")
    x0, noise = synth.normal(f"{tag}.x0", (B, 4, H, H)), synth.normal(f"{tag}.noise", (B, 4, H, H))
    t = synth.uniform(f"{tag}.t", (B,), lo=lo, hi=1.0 - lo)
    y = synth.integers(f"{tag}.y", (B,), cfg.n_classes)
    legs = {}
    for leg in ("fp32", "bf16"):
        P = {k: v.requires_grad_(True) for k, v in synth.dit_params(odit.param_shapes(cfg), seed=seed).items()}
        if leg == "bf16":
            with odit.bf16_autocast():
                _, loss = _flow_loss(P, cfg, x0, t, y, noise)
        else:
            _, loss = _flow_loss(P, cfg, x0, t, y, noise)
        loss.backward()
        legs[leg] = (loss.item(), {k: v.grad for k, v in P.items()})
    l32, lref = float(g[f"{tag}_loss_fp32"]), float(g[f"{tag}_loss_autocast"])
    assert abs(legs["fp32"][0] - l32) < 1e-6 * l32
    assert abs(legs["bf16"][0] - l32) <= max(2.0 * abs(lref - l32), 3e-4 * l32)  # (a scalar: either leg can cancel by luck)
    ref_err = dict(zip(g[f"{tag}_names"].tolist(), g[f"{tag}_err"].tolist()))
    assert set(ref_err) == set(legs["fp32"][1])
    ratios = [rel(legs["bf16"][1][n], legs["fp32"][1][n]) / e_ref for n, e_ref in ref_err.items()]
    assert 0.45 < min(ratios) and max(ratios) < 1.15, (min(ratios), max(ratios))
    assert 0.65 < float(np.median(ratios)) < 1.05


@pytest.mark.timeout(600)
def test_dit_s2_fwd_bwd(golden):
    g = golden("dit_s2")
    cfg = S2
    P = {k: v.requires_grad_(True) for k, v in synth.dit_params(odit.param_shapes(cfg), seed=7).items()}
    assert sum(v.numel() for v in P.values()) == 39_922_576
    B = 2
    x0 = synth.normal("s2.x0", (B, 4, 32, 32))
    noise = synth.normal("s2.noise", (B, 4, 32, 32))
    t = synth.uniform("s2.t", (B,), lo=0.05, hi=0.95)
    y = synth.integers("s2.y", (B,), 1000)
    pred, loss = _flow_loss(P, cfg, x0, t, y, noise)
    loss.backward()
    assert rel(pred, g["pred"]) < 5e-6
    assert abs(loss.item() - float(g["loss"])) / float(g["loss"]) < 1e-6
    norms = dict(zip(g["grad_names"].tolist(), g["grad_norms"].tolist()))
    for n, ref in norms.items():
        got = P[n].grad.double().norm().item()
        assert abs(got - ref) <= 2e-5 * max(ref, 1e-12), (n, got, ref)
    for k in g:
        if k.startswith("g_"):
            assert rel(P[k[2:]].grad, g[k]) < 2e-5, k
        elif k.startswith("gs_"):
            gr = P[k[3:]].grad
            assert rel(gr.flatten()[:: max(1, gr.numel() // 512)][:512], g[k]) < 2e-5, k


# ------------------------------------------------------------------ samplers
def test_sampler_steps(golden):
    g = golden("samplers")
    shp = (3, 4, 8, 8)
    xt, v, nz = synth.normal("smp.xt", shp), synth.normal("smp.v", shp), synth.normal("smp.noise", shp)
    r = od.euler_step(xt, v, 0.75, 0.5)
    assert np.array_equal(t2n(r["x_prev"]), g["euler_x_prev"]) and np.array_equal(t2n(r["estimated_x0"]), g["euler_x0"])
    tmax = od.flow_timesteps(10)[1]
    r = od.euler_maruyama_step(xt, v, 0.6, 0.5, tmax, 0.7, noise=torch.from_numpy(g["em_noise"]))
    for k in ("x_prev", "x_prev_mean", "x_prev_std", "estimated_x0", "logprob"):
        assert rel(r[k], g["em_" + k]) < 1e-6, k
    r = od.euler_maruyama_step(xt, v, 1.0, 0.9, tmax, 0.7, x_prev=nz)
    assert rel(r["logprob"], g["em2_logprob"]) < 1e-6 and rel(r["x_prev_mean"], g["em2_mean"]) < 1e-6
    T = od.GaussianTables(1000)
    tt = torch.tensor([0, 7, 999], dtype=torch.int32)
    dn = torch.from_numpy(g["ddpm_noise"])
    for mt in ("epsilon", "xstart", "xprev"):
        for vt in ("fixed_small", "fixed_large"):
            for clamp in (False, True):
                r = od.ddpm_step(T, v, tt, xt, dn, mt, vt, clamp)
                tag = f"ddpm_{mt}_{vt}_{int(clamp)}_"
                for k in ("x_prev", "estimated_x0", "x_prev_mean", "x_prev_std", "logprob"):
                    assert rel(r[k], g[tag + k]) < 1e-6, tag + k
    for eta in (0.0, 0.5):
        r = od.ddim_step(T, v, tt, xt, torch.from_numpy(g["ddim_noise"]), eta)
        keys = [k[len(f"ddim_eta{eta}_"):] for k in g if k.startswith(f"ddim_eta{eta}_")]
        assert set(keys) == set(r.keys())
        for k in keys:
            a, b = r[k], g[f"ddim_eta{eta}_{k}"]
            if k == "logprob":  # contains -inf/nan rows at t==0 where sigma==0 in the reference too
                a, b = torch.nan_to_num(a, 0, 0, 0), np.nan_to_num(b, nan=0, posinf=0, neginf=0)
            assert rel(a, b) < 1e-6, (eta, k)


def test_sampler_loops(golden):
    g = golden("samplers")
    cfg = SMALL
    P = synth.dit_params(odit.param_shapes(cfg), seed=5)
    y = synth.integers("smp.y", (2,), cfg.n_classes)
    x = synth.normal("smp.init", (2, 4, 8, 8))
    y_un = torch.full_like(y, cfg.n_classes)

    def vel(x, t, yy):
        tt = torch.full((x.shape[0],), t, dtype=torch.float32)
        return odit.dit_forward(P, x, tt, yy, cfg)

    with torch.no_grad():
        ts = od.flow_timesteps(4)
        xs, x0s, xc = [x], [], x
        for tc, tp in zip(ts[:-1], ts[1:]):
            vv = od.cfg_combine(vel(xc, tc, y), vel(xc, tc, y_un), 2.0)
            r = od.euler_step(xc, vv, tc, tp)
            xc = r["x_prev"]
            xs.append(xc)
            x0s.append(r["estimated_x0"])
        assert rel(xc, g["loop_euler_x"]) < 5e-6
        assert rel(torch.stack(xs, 1), g["loop_euler_xt"]) < 5e-6
        assert rel(torch.stack(x0s, 1), g["loop_euler_x0"]) < 5e-6
        ts = od.flow_timesteps(3, 4.63)
        xc = x
        for tc, tp in zip(ts[:-1], ts[1:]):
            xc = od.euler_step(xc, vel(xc, tc, y), tc, tp)["x_prev"]
        assert rel(xc, g["loop_euler_shift_x"]) < 5e-6
        # respaced DDPM, CFG 1.5, clamp
        T = od.GaussianTables(1000, n_steps=5)
        tmap = torch.tensor(T.timestep_map, dtype=torch.int32)
        nz = torch.from_numpy(g["loop_ddpm_noise"])
        xc = x
        for i, t in enumerate(reversed(range(5))):
            ti = torch.full((2,), t, dtype=torch.int32)
            tm = tmap[ti.long()]
            e = od.cfg_combine(odit.dit_forward(P, xc, tm, y, cfg), odit.dit_forward(P, xc, tm, y_un, cfg), 1.5)
            xc = od.ddpm_step(T, e, ti, xc, nz[i], clamp_x=True)["x_prev"]
        assert rel(xc, g["loop_ddpm_x"]) < 1e-5


# ------------------------------------------------------------------ loss curve (AdamW, DiT-S/2)
@pytest.mark.timeout(900)
def test_loss_curve_first_steps(golden):
    g = golden("loss_curve")
    cfg = S2
    P = {k: v.requires_grad_(True) for k, v in synth.dit_params(odit.param_shapes(cfg), seed=7).items()}
    opt = torch.optim.AdamW(list(P.values()), lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8)
    B = 4
    x0 = synth.normal("curve.x0", (B, 4, 32, 32))
    y = synth.integers("curve.y", (B,), 1000)
    for s in range(3):  # 3 of the 12 stored steps keep the CPU suite short; the GPU test walks all of them
        noise = synth.normal(f"curve.noise{s}", (B, 4, 32, 32))
        t = synth.uniform(f"curve.t{s}", (B,), lo=0.02, hi=0.98)
        opt.zero_grad()
        _, loss = _flow_loss(P, cfg, x0, t, y, noise)
        loss.backward()
        opt.step()
        assert abs(loss.item() - g["losses"][s]) / g["losses"][s] < 2e-5, s


# ------------------------------------------------------------------ UNet (config 1 family)
def test_unet_blocks_and_small_model(golden):
    from oracle import unet as ounet

    g = golden("unet")
    te = 128
    cfg_ss = ounet.UNetConfig(use_scale_shift_norm=True)
    for tag, kw, cin, cout in (("plain", {}, 64, 96), ("up", {"up": True}, 64, 64), ("down", {"down": True}, 64, 64)):
        b = ounet.Block("res", "", cin, cout, **kw)
        shapes = {"in_layers.0.weight": (cin,), "in_layers.0.bias": (cin,), "in_layers.2.weight": (cout, cin, 3, 3),
                  "in_layers.2.bias": (cout,), "emb_layers.1.weight": (2 * cout, te), "emb_layers.1.bias": (2 * cout,),
                  "out_layers.0.weight": (cout,), "out_layers.0.bias": (cout,), "out_layers.3.weight": (cout, cout, 3, 3),
                  "out_layers.3.bias": (cout,)}
        if cin != cout:
            shapes["skip_connection.weight"], shapes["skip_connection.bias"] = (cout, cin, 1, 1), (cout,)
        P = {n: synth.generic_params({f"rb_{tag}." + n: s}, seed=21)[f"rb_{tag}." + n].requires_grad_(True) for n, s in shapes.items()}
        x = synth.normal(f"rb_{tag}.x", (4, cin, 8, 8)).requires_grad_(True)
        emb = synth.normal(f"rb_{tag}.emb", (4, te)).requires_grad_(True)
        y = ounet.res_block(P, b, x, emb, cfg_ss)
        (y * synth.normal(f"rb_{tag}.dy", tuple(y.shape))).sum().backward()
        assert rel(y, g[f"rb_{tag}_y"]) < 2e-6, tag
        assert rel(x.grad, g[f"rb_{tag}_dx"]) < 5e-6 and rel(emb.grad, g[f"rb_{tag}_demb"]) < 5e-6, tag
        for n in shapes:
            assert rel(P[n].grad, g[f"rb_{tag}_g_{n}"]) < 1e-5, (tag, n)
    # attention block
    c = 128
    shapes = {"norm_x.weight": (c,), "norm_x.bias": (c,), "norm_context.weight": (c,), "norm_context.bias": (c,),
              "to_q.weight": (c, c, 1), "to_q.bias": (c,), "to_kv.weight": (2 * c, c, 1), "to_kv.bias": (2 * c,),
              "to_out.0.weight": (c, c, 1), "to_out.0.bias": (c,)}
    P = {n: synth.generic_params({"ab." + n: s}, seed=22)["ab." + n].requires_grad_(True) for n, s in shapes.items()}
    x = synth.normal("ab.x", (4, c, 8, 8)).requires_grad_(True)
    y = ounet.attention_block(P, ounet.Block("attn", "", c, c), x, ounet.UNetConfig(num_heads=2))
    (y * synth.normal("ab.dy", tuple(y.shape))).sum().backward()
    assert rel(y, g["ab_y"]) < 2e-6 and rel(x.grad, g["ab_dx"]) < 5e-6
    for n in shapes:
        assert rel(P[n].grad, g["ab_g_" + n]) < 1e-5, n
    # small UNet under the DDPM head
    cfg = ounet.UNetConfig(image_size=(16, 16), in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1,
                           attention_resolutions=(2,), channel_mult=(1, 2), num_heads=2, use_scale_shift_norm=True,
                           resblock_updown=True, n_classes=10, classifier_free=True)
    P = {k: v.requires_grad_(True) for k, v in synth.generic_params(ounet.param_shapes(cfg), seed=23).items()}
    B = 4
    x0 = synth.normal("un.x0", (B, 1, 16, 16))
    noise = synth.normal("un.noise", (B, 1, 16, 16))
    yl = synth.integers("un.y", (B,), 10)
    ti = torch.tensor([3, 500, 999, 0], dtype=torch.int32)
    T = od.GaussianTables(1000)
    pred = ounet.unet_forward(P, od.ddpm_add_noise(T, x0, ti, noise), ti, yl, cfg)
    assert rel(pred, g["un_pred"]) < 5e-6
    loss = od.mse_loss(pred, noise)
    loss.backward()
    assert abs(loss.item() - float(g["un_loss"])) / float(g["un_loss"]) < 1e-6
    norms = dict(zip(g["un_grad_names"].tolist(), g["un_grad_norms"].tolist()))
    floor = 1e-6 * max(norms.values())  # conv biases in front of a GroupNorm have an exactly-zero gradient: pure noise
    for n, ref in norms.items():
        got = P[n].grad.double().norm().item()
        assert (ref <= floor and got <= 10 * floor) or abs(got - ref) <= 2e-5 * ref, n
    for k in g:
        if k.startswith("un_g_") and norms[k[5:]] > floor:
            assert rel(P[k[5:]].grad, g[k]) < 2e-5, k


UNET_VARIANTS = {  # tests/golden/make_golden.py::UNET_VARIANTS
    "dflt": dict(image_size=(16, 16), in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1, attention_resolutions=(4,),
                 channel_mult=(1, 2, 2), num_heads=2, use_scale_shift_norm=False, resblock_updown=False, conv_resample=True,
                 n_classes=10, classifier_free=True),
    "pool": dict(image_size=(16, 16), in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1, attention_resolutions=(4,),
                 channel_mult=(1, 2, 2), num_heads=2, use_scale_shift_norm=True, resblock_updown=False, conv_resample=False,
                 n_classes=None, classifier_free=False),
}


@pytest.mark.parametrize("tag", ["dflt", "pool"])
def test_unet_constructor_default_and_pooling_variants(golden, tag):
    """the UNetModel constructor defaults (additive conditioning, Downsample / Upsample with their 3x3 convs) and the conv-free
    resampling variant, DDPM loss fwd + bwd, against the reference run (tests/golden/unet_variants.npz)"""
    from oracle import unet as ounet

    g = golden("unet_variants")
    cfg = ounet.UNetConfig(**UNET_VARIANTS[tag])
    P = {k: v.requires_grad_(True) for k, v in synth.generic_params(ounet.param_shapes(cfg), seed=29).items()}
    B = 4
    x0, noise = synth.normal(f"uv.{tag}.x0", (B, 1, 16, 16)), synth.normal(f"uv.{tag}.noise", (B, 1, 16, 16))
    yl = synth.integers(f"uv.{tag}.y", (B,), 10) if cfg.n_classes else None
    ti = torch.tensor([7, 250, 999, 0], dtype=torch.int32)
    pred = ounet.unet_forward(P, od.ddpm_add_noise(od.GaussianTables(1000), x0, ti, noise), ti, yl, cfg)
    assert rel(pred, g[f"{tag}_pred"]) < 5e-6
    loss = od.mse_loss(pred, noise)
    loss.backward()
    assert abs(loss.item() - float(g[f"{tag}_loss"])) / float(g[f"{tag}_loss"]) < 1e-6
    norms = dict(zip(g[f"{tag}_grad_names"].tolist(), g[f"{tag}_grad_norms"].tolist()))
    assert set(norms) == set(P)
    floor = 1e-6 * max(norms.values())
    for n, ref in norms.items():
        got = P[n].grad.double().norm().item()
        assert (ref <= floor and got <= 10 * floor) or abs(got - ref) <= 2e-5 * ref, n
    pre = f"{tag}_g_"
    for k in g:
        if k.startswith(pre) and norms[k[len(pre):]] > floor:
            assert rel(P[k[len(pre):]].grad, g[k]) < 2e-5, k


def test_unet_at_config1_dims_against_the_reference(golden):
    """configs/model/unet.yaml dims (276.7 M parameters, the 1024 / 1536 / 2048-channel stages and the 12-channels-per-group GroupNorms
    the 32-channel fixtures never reach), B = 2, DDPM loss fwd + bwd: the oracle against the reference's own fp32 run
    (tests/golden/unet_full.npz: prediction, loss, every gradient norm, the small gradient tensors in full)"""
    from oracle import unet as ounet

    g = golden("unet_full")
    cfg = ounet.UNetConfig()
    P = {k: v.requires_grad_(True) for k, v in synth.generic_params(ounet.param_shapes(cfg), seed=41).items()}
    B = 2
    x0, noise = synth.normal("fd.x0", (B, 1, 32, 32)), synth.normal("fd.noise", (B, 1, 32, 32))
    y = synth.integers("fd.y", (B,), 10)
    ti = torch.tensor([17, 940], dtype=torch.int32)
    pred = ounet.unet_forward(P, od.ddpm_add_noise(od.GaussianTables(1000), x0, ti, noise), ti, y, cfg)
    assert rel(pred, g["pred"]) < 1e-5
    loss = od.mse_loss(pred, noise)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) / float(g["loss"]) < 2e-6
    norms = dict(zip(g["names"].tolist(), g["grad_norms"].tolist()))
    assert set(norms) == set(P)
    floor = 1e-6 * max(norms.values())
    for n, ref in norms.items():
        got = P[n].grad.double().norm().item()
        assert (ref <= floor and got <= 10 * floor) or abs(got - ref) <= 5e-5 * ref, (n, got, ref)
    for k in g:
        if k.startswith("g_") and norms[k[2:]] > floor:
            assert rel(P[k[2:]].grad, g[k]) < 5e-5, k


@pytest.mark.parametrize("tag", ["ddt", "sprint", "ddt_txt", "sprint_txt"])
def test_ddt_and_sprint_at_yaml_dims_against_the_reference(golden, tag):
    """configs/model/{ddt,sprint,ddt_txt,sprint_txt}.yaml dims (512 / 8, 640 / 10, 768 / 12 heads; 1024 image + 128 text tokens for the
    txt forms): the oracle's prediction, every gradient norm and the small gradient tensors against the reference's own fp32 run
    (tests/golden/yaml_dims.npz), token scores as the reference drew them"""
    import yaml

    from oracle import ddt as oddt
    from oracle import sprint as osprint

    g = golden("yaml_dims")
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "configs", "model", tag + ".yaml")) as f:
        kw = {k: v for k, v in yaml.safe_load(f).items() if k not in ("_target_", "simple_dit", "simple_ddt", "use_checkpoint")}
    Lc, Dc = 128, 1024
    if tag.endswith("_txt"):
        pre, B = ("ft", 2) if tag == "ddt_txt" else ("f5", 2)
        x, dy = synth.normal(f"{pre}.x", (B, 128, 32, 32)), synth.normal(f"{pre}.dy", (B, 128, 32, 32))
        t = torch.tensor([0.27, 0.88] if tag == "ddt_txt" else [0.31, 0.83])
        keep = torch.arange(Lc)[None, :] < torch.tensor([51, Lc] if tag == "ddt_txt" else [Lc, 37])[:, None]
        ctx = synth.normal(f"{pre}.ctx", (B, Lc, Dc)) * 0.5
    else:
        B = 4
        x, dy = synth.normal("fc.x", (B, 3, 32, 32)), synth.normal("fc.dy", (B, 3, 32, 32))
        t, y = synth.uniform("fc.t", (B,), lo=0.05, hi=0.95), synth.integers("fc.y", (B,), 10)
    if tag == "ddt":
        cfg = oddt.DDTConfig(**kw)
        P = synth.dit_params(oddt.param_shapes(cfg), seed=111)
        run = lambda Q: oddt.ddt_forward(Q, x, t, y, cfg)  # noqa: E731
    elif tag == "ddt_txt":
        cfg = oddt.DDTJointConfig(context_dim=Dc, **kw)
        P = synth.dit_params(oddt.joint_param_shapes(cfg), seed=117)
        run = lambda Q: oddt.ddt_joint_forward(Q, x, t, ctx, keep, cfg)  # noqa: E731
    else:
        joint = tag == "sprint_txt"
        cfg = osprint.SprintJointConfig(context_dim=Dc, **kw) if joint else osprint.SprintConfig(**kw)
        shapes = osprint.joint_param_shapes(cfg) if joint else osprint.param_shapes(cfg)
        P = synth.dit_params({k: v for k, v in shapes.items() if k != "mask_token"}, seed=91 if joint else 113)
        P["mask_token"] = synth.normal("f5.mask" if joint else "fc.mask", shapes["mask_token"]) * 0.5
        scores = torch.as_tensor(g[f"{tag}_scores"])
        kept = osprint.kept_indices(scores, osprint.n_kept(scores.shape[1], cfg.drop_rate))
        if joint:
            run = lambda Q: osprint.sprint_mmdit_forward(Q, x, t, ctx, keep, cfg, kept=kept)  # noqa: E731
        else:
            run = lambda Q: osprint.sprint_forward(Q, x, t, y, cfg, kept=kept)  # noqa: E731
    P = {k: v.requires_grad_(True) for k, v in P.items()}
    pred = run(P)
    assert rel(pred, g[f"{tag}_pred"]) < 1e-5
    (pred * dy).sum().backward()
    norms = dict(zip(g[f"{tag}_names"].tolist(), g[f"{tag}_grad_norms"].tolist()))
    assert set(norms) == {n for n, v in P.items() if v.grad is not None}
    for n, ref in norms.items():
        assert abs(P[n].grad.double().norm().item() - ref) <= 5e-5 * ref, (n, ref)
    pre = f"{tag}_g_"
    for k in g:
        if k.startswith(pre):
            assert rel(P[k[len(pre):]].grad, g[k]) < 5e-5, k


def test_repa_loss_hooked_into_small_dit(golden):
    """(ix) REPA alignment loss on the output of block 0 of the small DiT next to the flow loss: both losses, the projector
    gradients and every denoiser gradient (the feature gradient re-enters the residual stream) vs the reference"""
    from oracle import repa as orepa

    g = golden("repa")
    cfg = SMALL
    P = {k: v.requires_grad_(True) for k, v in synth.dit_params(odit.param_shapes(cfg), seed=5).items()}
    R = {k: v.requires_grad_(True) for k, v in synth.generic_params(orepa.param_shapes(cfg.inner_dim, 128, 64), seed=41).items()}
    B, H = 4, 16
    x0, noise = synth.normal("rp.x0", (B, 4, H, H)), synth.normal("rp.noise", (B, 4, H, H))
    y, t = synth.integers("rp.y", (B,), 10), synth.uniform("rp.t", (B,), lo=0.05, hi=0.95)
    dst = synth.normal("rp.dst", (B, (H // cfg.patch_size) ** 2, 64))
    taps: dict = {}
    pred = odit.dit_forward(P, od.flow_add_noise(x0, t, noise), t, y, cfg, taps=taps)
    loss = od.flow_loss(pred, x0, noise)
    repa = orepa.repa_loss(R, taps["layer0"], dst, coeff=0.5)
    (loss + repa).backward()
    assert abs(loss.item() - float(g["loss"])) / float(g["loss"]) < 1e-6
    assert abs(repa.item() - float(g["repa"])) / float(g["repa"]) < 1e-6
    for n, v in R.items():
        assert rel(v.grad, g["g_" + n]) < 1e-5, n
    for n, v in P.items():
        assert rel(v.grad, g["gd_" + n]) < 2e-5, n


def test_perceiver_resampler(golden):
    """(x) the REPA config's Perceiver resampler: output, input gradient and parameter gradients vs the reference module"""
    from oracle import repa as orepa

    g = golden("resampler")
    kw = dict(dim=128, depth=2, head_dim=64, num_heads=2, ff_mult=4, num_latents=256)
    P = {k: v.requires_grad_(True) for k, v in synth.generic_params(orepa.resampler_param_shapes(**kw), seed=51).items()}
    x = synth.normal("rs.x", (3, 64, 128)).requires_grad_(True)
    y = orepa.perceiver_resampler(P, x, depth=2, head_dim=64, num_heads=2)
    (y * synth.normal("rs.dy", tuple(y.shape))).sum().backward()
    assert rel(y, g["y"]) < 2e-6 and rel(x.grad, g["dx"]) < 1e-5
    for n, v in P.items():
        assert rel(v.grad, g["g_" + n]) < 2e-5, n


def test_sprint_dit(golden):
    """(xi) SprintDiT(simple_dit=True): training forward/backward with the recorded token-drop scores, label / path drops,
    eval forward with and without the deep path, and a guided 4-step Euler sampling loop"""
    from oracle import sprint as osprint

    g = {k: torch.as_tensor(v) for k, v in golden("sprint").items()}
    kw = dict(input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4, patch_size=2,
              encoder_depth=1, deep_layers_depth=2, decoder_depth=1, n_classes=10, classifier_free=True, drop_rate=0.75)
    cfg = osprint.SprintConfig(**kw)
    shapes = osprint.param_shapes(cfg)
    P = synth.dit_params({k: v for k, v in shapes.items() if k != "mask_token"}, seed=61)
    P["mask_token"] = synth.normal("sp.mask", shapes["mask_token"]) * 0.5
    P = {k: v.requires_grad_(True) for k, v in P.items()}
    B, H = 4, 32
    x, t, y = synth.normal("sp.x", (B, 4, H, H)), synth.uniform("sp.t", (B,), lo=0.05, hi=0.95), synth.integers("sp.y", (B,), 10)
    dy = synth.normal("sp.dy", (B, 4, H, H))
    k = osprint.n_kept(256, 0.75)
    assert k == 64
    # (a)
    pred = osprint.sprint_forward(P, x, t, y, cfg, kept=osprint.kept_indices(g["a_scores"], k))
    assert rel(pred, g["a_pred"]) < 2e-6
    (pred * dy).sum().backward()
    checked = 0
    for n, v in P.items():
        if "a_g_" + n in g:
            assert rel(v.grad, g["a_g_" + n]) < 2e-5, n
            checked += 1
        v.grad = None
    assert checked > 40
    # (b)
    y_eff = torch.where(g["b_label_u"] < 0.5, torch.full_like(y, 10), y)
    pred = osprint.sprint_forward(P, x, t, y_eff, cfg, kept=osprint.kept_indices(g["b_scores"], k), path_drop=g["b_path_u"] < 0.5)
    assert rel(pred, g["b_pred"]) < 2e-6
    (pred * dy).sum().backward()
    for n in ("mask_token", "fuse.weight", "layers.0.attention.qkv.weight", "deep_layers.1.mlp_input.2.weight",
              "decoder_layers.0.modulation.lin.weight", "label_embed.embedding.weight"):
        assert rel(P[n].grad, g["b_g_" + n]) < 2e-5, n
    # (c), (d)
    with torch.no_grad():
        assert rel(osprint.sprint_forward(P, x, t, y, cfg), g["c_pred"]) < 2e-6
        assert rel(osprint.sprint_forward(P, x, t, torch.full_like(y, 10), cfg, skip_deep=True), g["d_pred"]) < 2e-6
        # (e) flow.py:410-524 + euler.py:22-41: v = v_uncond + s (v_cond - v_uncond), x -= v dt
        xs = synth.normal("sp.init", (B, 4, H, H))
        ts = [1.0, 0.75, 0.5, 0.25, 0.0]
        for a, b in zip(ts[:-1], ts[1:]):
            tt = torch.full((B,), a)
            vc = osprint.sprint_forward(P, xs, tt, y, cfg)
            vu = osprint.sprint_forward(P, xs, tt, torch.full_like(y, 10), cfg, skip_deep=True)
            xs = xs - (vu + 2.0 * (vc - vu)) * (a - b)
        assert rel(xs, g["e_loop_x"]) < 1e-5


def test_mmdit_joint_blocks(golden):
    """(xii) MMDiT(simple_dit=False) with a precomputed context: joint text-image attention with a ragged key-padding mask,
    two modulation / MLP streams, 3-axis RoPE; context drop to the null embedding; guided sampling loop"""
    from oracle import mmdit as ommdit

    g = {k: torch.as_tensor(v) for k, v in golden("mmdit_joint").items()}
    kw = dict(input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4, patch_size=2, depth=2,
              rope_axes_dim=[16, 24, 24], rope_base=2000, classifier_free=True)
    cfg = ommdit.JointConfig(context_dim=96, **kw)
    P = {k: v.requires_grad_(True) for k, v in synth.dit_params(ommdit.param_shapes(cfg), seed=71).items()}
    Lc, Cd, B, H = 64, 96, 4, 16
    null = (synth.normal("mj.null", (1, Lc, Cd)) * 0.5)[0]
    null_keep = torch.arange(Lc) < 7
    x, t = synth.normal("mj.x", (B, 4, H, H)), synth.uniform("mj.t", (B,), lo=0.05, hi=0.95)
    ctx = synth.normal("mj.ctx", (B, Lc, Cd))
    keep = torch.arange(Lc)[None, :] < torch.tensor([64, 20, 41, 5])[:, None]
    dy = synth.normal("mj.dy", (B, 4, H, H))
    pred = ommdit.mmdit_forward(P, x, t, ctx, keep, cfg)
    assert rel(pred, g["a_pred"]) < 2e-6
    (pred * dy).sum().backward()
    for n, v in P.items():
        if "a_g_" + n in g:
            assert rel(v.grad, g["a_g_" + n]) < 2e-5, n
        else:  # context branch of the last block feeds nothing: no gradient in the reference, none here
            assert n.startswith("layers.1.") and "context" in n and v.grad is None, n
    with torch.no_grad():
        c2, k2 = ommdit.drop_context(ctx, keep, null, null_keep, g["b_u"] < 0.5)
        assert rel(ommdit.mmdit_forward(P, x, t, c2, k2, cfg), g["b_pred"]) < 2e-6
        xs = synth.normal("mj.init", (B, 4, H, H))
        cu, ku = ommdit.drop_context(ctx, keep, null, null_keep, torch.ones(B, dtype=torch.bool))
        ts = [1.0, 0.75, 0.5, 0.25, 0.0]
        for a, b in zip(ts[:-1], ts[1:]):
            tt = torch.full((B,), a)
            vc, vu = ommdit.mmdit_forward(P, xs, tt, ctx, keep, cfg), ommdit.mmdit_forward(P, xs, tt, cu, ku, cfg)
            xs = xs - (vu + 2.0 * (vc - vu)) * (a - b)
        assert rel(xs, g["e_loop_x"]) < 1e-5


def test_sprint_dit_joint_form(golden):
    """(xiii) SprintDiT(simple_dit=False): joint encoder / decoder blocks, single-stream deep blocks on the kept tokens,
    fuse + fuse_context; recorded token scores, context drop and path drop; eval with and without the deep path"""
    from oracle import mmdit as ommdit
    from oracle import sprint as osprint

    raw = golden("sprint_joint")
    no_grad = set(str(n) for n in raw["a_none"])
    g = {k: torch.as_tensor(v) for k, v in raw.items() if k != "a_none"}
    kw = dict(input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4, patch_size=1,
              encoder_depth=1, deep_layers_depth=3, n_single_stream_blocks=2, decoder_depth=2, rope_axes_dim=[16, 24, 24],
              rope_base=2000, classifier_free=True, drop_rate=0.75)
    cfg = osprint.SprintJointConfig(context_dim=96, **kw)
    shapes = osprint.joint_param_shapes(cfg)
    P = synth.dit_params({k: v for k, v in shapes.items() if k != "mask_token"}, seed=81)
    P["mask_token"] = synth.normal("sj.mask", shapes["mask_token"]) * 0.5
    P = {k: v.requires_grad_(True) for k, v in P.items()}
    Lc, Cd, B, H = 64, 96, 4, 16
    null, null_keep = (synth.normal("sj.null", (1, Lc, Cd)) * 0.5)[0], torch.arange(Lc) < 7
    x, t = synth.normal("sj.x", (B, 4, H, H)), synth.uniform("sj.t", (B,), lo=0.05, hi=0.95)
    ctx = synth.normal("sj.ctx", (B, Lc, Cd))
    keep = torch.arange(Lc)[None, :] < torch.tensor([64, 20, 41, 5])[:, None]
    dy = synth.normal("sj.dy", (B, 4, H, H))
    pred = osprint.sprint_mmdit_forward(P, x, t, ctx, keep, cfg, kept=osprint.kept_indices(g["a_scores"], 64))
    assert rel(pred, g["a_pred"]) < 2e-6
    (pred * dy).sum().backward()
    checked = 0
    for n, v in P.items():
        if "a_g_" + n in g:
            assert rel(v.grad, g["a_g_" + n]) < 2e-5, n
            checked += 1
        assert (v.grad is None) == (n in no_grad), n  # context branch of the last decoder block feeds nothing
    assert checked > 80 and all(n.startswith("decoder_layers.1.") and "context" in n for n in no_grad)
    with torch.no_grad():
        c2, k2 = ommdit.drop_context(ctx, keep, null, null_keep, g["b_ctx_u"] < 0.5)
        pred = osprint.sprint_mmdit_forward(P, x, t, c2, k2, cfg, kept=osprint.kept_indices(g["b_scores"], 64),
                                            path_drop=g["b_path_u"] < 0.5)
        assert rel(pred, g["b_pred"]) < 2e-6
        assert rel(osprint.sprint_mmdit_forward(P, x, t, ctx, keep, cfg), g["c_pred"]) < 2e-6
        cu, ku = ommdit.drop_context(ctx, keep, null, null_keep, torch.ones(B, dtype=torch.bool))
        assert rel(osprint.sprint_mmdit_forward(P, x, t, cu, ku, cfg, skip_deep=True), g["d_pred"]) < 2e-6


def test_ddt_simple(golden):
    """(xiv) DDT(simple_ddt=True): DiT encoder, decoder DiT blocks with per-token adaLN conditioning silu(enc + t_emb), guided
    sampling loop"""
    from oracle import ddt as oddt

    g = {k: torch.as_tensor(v) for k, v in golden("ddt").items()}
    kw = dict(input_channels=4, output_channels=4, inner_dim=128, num_heads=2, mlp_ratio=4, patch_size=2, encoder_depth=2,
              decoder_depth=2, n_classes=10, classifier_free=True)
    cfg = oddt.DDTConfig(**kw)
    P = {k: v.requires_grad_(True) for k, v in synth.dit_params(oddt.param_shapes(cfg), seed=91).items()}
    B, H = 4, 16
    x, t, y = synth.normal("dd.x", (B, 4, H, H)), synth.uniform("dd.t", (B,), lo=0.05, hi=0.95), synth.integers("dd.y", (B,), 10)
    dy = synth.normal("dd.dy", (B, 4, H, H))
    pred = oddt.ddt_forward(P, x, t, y, cfg)
    assert rel(pred, g["pred"]) < 2e-6
    (pred * dy).sum().backward()
    checked = 0
    for n, v in P.items():
        if "g_" + n in g:
            assert rel(v.grad, g["g_" + n]) < 2e-5, n
            checked += 1
    assert checked > 40
    with torch.no_grad():
        xs = synth.normal("dd.init", (B, 4, H, H))
        ts = [1.0, 0.75, 0.5, 0.25, 0.0]
        for a, b in zip(ts[:-1], ts[1:]):
            tt = torch.full((B,), a)
            vc, vu = oddt.ddt_forward(P, xs, tt, y, cfg), oddt.ddt_forward(P, xs, tt, torch.full_like(y, 10), cfg)
            xs = xs - (vu + 2.0 * (vc - vu)) * (a - b)
        assert rel(xs, g["loop_x"]) < 1e-5


def test_ddt_joint_encoder(golden):
    """(xv) DDT(simple_ddt=False): joint text-image encoder blocks, per-token-conditioned decoder on the 3-axis image RoPE rows"""
    from oracle import ddt as oddt
    from oracle import mmdit as ommdit

    raw = golden("ddt_joint")
    none = set(str(n) for n in raw["none"])
    g = {k: torch.as_tensor(v) for k, v in raw.items() if k != "none"}
    kw = dict(input_channels=4, output_channels=4, inner_dim=128, num_heads=2, mlp_ratio=4, patch_size=2, encoder_depth=2,
              decoder_depth=2, rope_axes_dim=[16, 24, 24], rope_base=1000, classifier_free=True)
    cfg = oddt.DDTJointConfig(context_dim=96, **kw)
    P = {k: v.requires_grad_(True) for k, v in synth.dit_params(oddt.joint_param_shapes(cfg), seed=101).items()}
    Lc, Cd, B, H = 64, 96, 4, 16
    null, null_keep = (synth.normal("dj.null", (1, Lc, Cd)) * 0.5)[0], torch.arange(Lc) < 7
    x, t = synth.normal("dj.x", (B, 4, H, H)), synth.uniform("dj.t", (B,), lo=0.05, hi=0.95)
    ctx = synth.normal("dj.ctx", (B, Lc, Cd))
    keep = torch.arange(Lc)[None, :] < torch.tensor([64, 20, 41, 5])[:, None]
    dy = synth.normal("dj.dy", (B, 4, H, H))
    pred = oddt.ddt_joint_forward(P, x, t, ctx, keep, cfg)
    assert rel(pred, g["pred"]) < 2e-6
    (pred * dy).sum().backward()
    checked = 0
    for n, v in P.items():
        if "g_" + n in g:
            assert rel(v.grad, g["g_" + n]) < 2e-5, n
            checked += 1
        assert (v.grad is None) == (n in none), n
    assert checked > 50
    with torch.no_grad():
        c2, k2 = ommdit.drop_context(ctx, keep, null, null_keep, g["b_u"] < 0.5)
        assert rel(oddt.ddt_joint_forward(P, x, t, c2, k2, cfg), g["b_pred"]) < 2e-6


def test_mmdit_with_single_stream_blocks(golden):
    """(xvi) MMDiT(simple_dit=False, n_single_stream_blocks=2): one joint block, then two single-stream blocks on [context ; image]"""
    from oracle import mmdit as ommdit

    raw = golden("mmdit_single")
    none = set(str(n) for n in raw["none"])
    g = {k: torch.as_tensor(v) for k, v in raw.items() if k != "none"}
    kw = dict(input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4, patch_size=2, depth=3,
              rope_axes_dim=[16, 24, 24], rope_base=2000, classifier_free=True, n_single_stream_blocks=2)
    cfg = ommdit.JointConfig(context_dim=96, **kw)
    P = {k: v.requires_grad_(True) for k, v in synth.dit_params(ommdit.param_shapes(cfg), seed=111).items()}
    Lc, Cd, B, H = 64, 96, 4, 16
    x, t = synth.normal("ms.x", (B, 4, H, H)), synth.uniform("ms.t", (B,), lo=0.05, hi=0.95)
    ctx = synth.normal("ms.ctx", (B, Lc, Cd))
    keep = torch.arange(Lc)[None, :] < torch.tensor([64, 20, 41, 5])[:, None]
    dy = synth.normal("ms.dy", (B, 4, H, H))
    pred = ommdit.mmdit_forward(P, x, t, ctx, keep, cfg)
    assert rel(pred, g["pred"]) < 2e-6
    (pred * dy).sum().backward()
    checked = 0
    for n, v in P.items():
        if "g_" + n in g:
            assert rel(v.grad, g["g_" + n]) < 2e-5, n
            checked += 1
        assert (v.grad is None) == (n in none), n
    assert checked > 30
