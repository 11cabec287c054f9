"""Structural fuzz of the denoiser constructors (round 5): seeded random UNetModel / MMDiT / DDT / SprintDiT configurations -- channel widths whose
GroupNorm groups have 3 / 5 / 6 / 10 / 12 / 20 channels, non-square and odd-sized maps (1, 9, 36, 144, 576 ... attention tokens), every
head count, additive and FiLM conditioning, all three resampling forms, batch 1-5 -- in BOTH precision regimes against the CPU oracle:
prediction and every parameter gradient.  The fixtures and the configuration yamls exercise a handful of shapes; this is what found
that the UNet attention refused (or crashed on) every map above 64 tokens and that the f32 softmax refused 1 / 9 / 25-token rows.  A
configuration the engine cannot run must refuse with NotImplementedError naming the way out -- the only such case left is the bf16
DiT on a token grid that is not a multiple of 64 (the fp32 regime takes any grid)."""
import random
import traceback

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dit as odit  # noqa: E402
from oracle import synth  # noqa: E402
from oracle import unet as ounet  # noqa: E402

DEV = "cuda"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _verdict(prec, pred_err, live, Pr, loose=False):
    """bf16 regime: prediction 3e-2, matrices 6e-2 (attention projections 9e-2), vectors (cancelling sums over pixels / tokens) 1.5e-1 (DDT, whose reference loses
    3.5e-2 / 8e-2 under autocast itself: twice that); fp32 regime: 2e-5 / 1e-4"""
    k = 2.0 if loose else 1.0
    # (a ResBlock's emb_layers gradient is the per-sample pixel sum of the block's activation gradient times emb: it rounds like a bias)
    # (the query / key-value projections of an AttentionBlock see their gradient through the softmax of a 1-4-head attention over a
    #  few tokens: the reference's own autocast error on them is 3-4e-2 at configuration dims; 6.2e-2 observed here on one draw)
    tol = lambda n, dim: (k * (1.5e-1 if dim == 1 or "emb_layers" in n else 9e-2 if (".to_q." in n or ".to_kv." in n) else 6e-2)) if prec == "bf16" else 1e-4  # noqa: E731
    errs = [(rel(p.grad, Pr[n].grad) / tol(n, p.dim()), rel(p.grad, Pr[n].grad), n) for n, p in live]
    worst = max(errs)
    bad = pred_err > (3e-2 * k if prec == "bf16" else 2e-5) or worst[0] > 1.0
    return ("**BAD** " if bad else "") + f"{prec}: pred {pred_err:.1e} worst grad {worst[1]:.1e} ({worst[0]:.2f} of its bound) {worst[2]}"


def run_unet(i, rng):
    from diffulab_amd import UNetModel

    mc = rng.choice([32, 64, 96, 160])
    mult = rng.choice([(1, 2), (1, 2, 2), (1, 2, 3), (1, 1, 2, 2), (2, 1)])
    H, W = rng.choice([(16, 16), (32, 32), (16, 32), (8, 8), (24, 24)])
    down = 2 ** (len(mult) - 1)
    att = tuple(sorted(rng.sample([1, 2, 4, 8], rng.randint(0, 2))))
    kw = dict(image_size=(H, W), in_channels=rng.choice([1, 3, 4]), model_channels=mc, out_channels=rng.choice([1, 3, 4, 6]),
              num_res_blocks=rng.randint(1, 3), attention_resolutions=att, channel_mult=mult, num_heads=rng.choice([1, 2, 4]),
              use_scale_shift_norm=rng.random() < 0.5, resblock_updown=rng.random() < 0.5, conv_resample=rng.random() < 0.7,
              n_classes=rng.choice([None, 10]), classifier_free=rng.random() < 0.5)
    if kw["n_classes"] is None:
        kw["classifier_free"] = False
    B = rng.choice([1, 2, 3, 5])
    tag = f"unet#{i} {kw} B={B}"
    if H % down or W % down:
        return tag, "skip (size)"
    cfg = ounet.UNetConfig(**kw)
    P = synth.generic_params(ounet.param_shapes(cfg), seed=1000 + i)
    mk = dict(kw, image_size=list(kw["image_size"]), attention_resolutions=list(att), channel_mult=", ".join(map(str, mult)))
    x = synth.normal(f"fz.x{i}", (B, kw["in_channels"], H, W))
    t = synth.integers(f"fz.t{i}", (B,), 1000).float()
    y = synth.integers(f"fz.y{i}", (B,), 10) if kw["n_classes"] else None
    dy = synth.normal(f"fz.dy{i}", (B, kw["out_channels"], H, W))
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = ounet.unet_forward(Pr, x, t, y, cfg)
    (ref * dy).sum().backward()
    res = []
    for prec in ("fp32", "bf16"):
        try:
            m = UNetModel(**mk)
            m.load_state_dict(P)
            m = m.set_precision(prec).to(DEV).train()
            cond = {"y": y.to(DEV), "p": 0.0} if y is not None else {}
            pred = m(x=x.to(DEV), timesteps=t.to(DEV), **cond)["x"]
            (pred * dy.to(DEV)).sum().backward()
            torch.cuda.synchronize()
        except NotImplementedError as e:
            res.append(f"{prec}: refused ({str(e)[:160]})")
            continue
        gmax = max(v.grad.norm().item() for v in Pr.values() if v.grad is not None)
        live = [(n, p) for n, p in m.named_parameters() if Pr[n].grad is not None and Pr[n].grad.norm().item() > 1e-5 * gmax]
        res.append(_verdict(prec, rel(pred, ref), live, Pr))
    return tag, " | ".join(res)


def run_tokens(i, rng, family="dit", grid=None, batch=None):
    """class-conditional token models: MMDiT(simple_dit), DDT(simple_ddt), SprintDiT(simple_dit)"""
    from oracle import ddt as oddt
    from oracle import sprint as osprint

    from diffulab_amd import DDT, MMDiT, SprintDiT

    H_ = rng.choice([1, 2, 3, 4, 6])
    D = H_ * 64
    patch = rng.choice([1, 2, 4])
    gh, gw = rng.choice([(8, 8), (16, 16), (8, 16), (4, 4), (16, 8), (12, 12), (6, 10)])
    C = rng.choice([3, 4, 16])
    kw = dict(input_channels=C, output_channels=C, inner_dim=D, num_heads=H_, mlp_ratio=rng.choice([2, 4]), patch_size=patch,
              n_classes=rng.choice([None, 10]), classifier_free=rng.random() < 0.5)
    if kw["n_classes"] is None:
        kw["classifier_free"] = False
    B = rng.choice([1, 2, 3, 5])
    if grid is not None:
        (gh, gw), B = grid, batch
    x = synth.normal(f"fd.x{i}", (B, C, gh * patch, gw * patch))
    t = synth.uniform(f"fd.t{i}", (B,), lo=0.05, hi=0.95)
    y = synth.integers(f"fd.y{i}", (B,), 10) if kw["n_classes"] else None
    dy = synth.normal(f"fd.dy{i}", tuple(x.shape))
    scores = None
    if family == "dit":
        kw.update(embedding_dim=rng.choice([64, 128, D]), depth=rng.randint(1, 3))
        cfg = odit.DiTConfig(**kw)
        P = synth.dit_params(odit.param_shapes(cfg), seed=2000 + i)
        build = lambda: MMDiT(simple_dit=True, **kw)  # noqa: E731
        oracle = lambda Q: odit.dit_forward(Q, x, t, y, cfg)  # noqa: E731
    elif family == "ddt":
        kw.update(encoder_depth=rng.randint(1, 2), decoder_depth=rng.randint(1, 2))
        cfg = oddt.DDTConfig(**kw)
        P = synth.dit_params(oddt.param_shapes(cfg), seed=3000 + i)
        build = lambda: DDT(simple_ddt=True, **kw)  # noqa: E731
        oracle = lambda Q: oddt.ddt_forward(Q, x, t, y, cfg)  # noqa: E731
    else:
        kw.update(embedding_dim=rng.choice([64, D]), encoder_depth=1, deep_layers_depth=rng.randint(1, 2), decoder_depth=1,
                  drop_rate=rng.choice([0.75, 0.5]))
        cfg = osprint.SprintConfig(**kw)
        shapes = osprint.param_shapes(cfg)
        P = synth.dit_params({k: v for k, v in shapes.items() if k != "mask_token"}, seed=4000 + i)
        P["mask_token"] = synth.normal(f"fd.mask{i}", shapes["mask_token"]) * 0.5
        scores = synth.uniform(f"fd.sc{i}", (B, gh * gw))
        kept = osprint.kept_indices(scores, osprint.n_kept(gh * gw, cfg.drop_rate))
        build = lambda: SprintDiT(simple_dit=True, **kw)  # noqa: E731
        oracle = lambda Q: osprint.sprint_forward(Q, x, t, y, cfg, kept=kept)  # noqa: E731
    tag = f"{family}#{i} {kw} grid={gh}x{gw} B={B}"
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = oracle(Pr)
    (ref * dy).sum().backward()
    res = []
    for prec in ("fp32", "bf16"):
        try:
            m = build()
            m.load_state_dict(P)
            m = m.set_precision(prec).to(DEV).train()
            if scores is not None:
                m._draw_scores = lambda B_, S_, device: scores.to(device)
            cond = {"y": y.to(DEV), "p": 0.0} if y is not None else {}
            pred = m(x=x.to(DEV), timesteps=t.to(DEV), **cond)["x"]
            (pred * dy.to(DEV)).sum().backward()
            torch.cuda.synchronize()
        except NotImplementedError as e:
            res.append(f"{prec}: refused ({str(e)[:160]})")
            continue
        live = [(n, p) for n, p in m.named_parameters() if Pr[n].grad is not None]
        res.append(_verdict(prec, rel(pred, ref), live, Pr, loose=family == "ddt"))
    return tag, " | ".join(res)


def run_joint(i, rng, family="mmdit_joint", grid=None, batch=None):
    """text-image forms behind a precomputed-embedding context embedder (bf16 regime only: their reference configurations pin it):
    MMDiT(simple_dit=False) with 0-2 single-stream blocks, SprintDiT(simple_dit=False), DDT(simple_ddt=False); random context
    lengths (multiples of 64 or not), ragged masks, head counts, grids"""
    from oracle import ddt as oddt
    from oracle import mmdit as ommdit
    from oracle import sprint as osprint

    from diffulab_amd import DDT, MMDiT, SprintDiT
    from diffulab_amd.networks.embedders import PrecomputedEmbedder

    H_ = rng.choice([1, 2, 3, 4])
    D = H_ * 64
    patch = rng.choice([1, 2])
    gh, gw = rng.choice([(8, 8), (16, 16), (8, 16), (16, 8)])
    C = rng.choice([4, 16])
    Lc = rng.choice([64, 128, 77, 33, 96, 5])
    Dc = rng.choice([32, 96, 160])
    B = rng.choice([1, 2, 3, 4])
    if grid is not None:
        (gh, gw), B, patch = grid, batch, 1
    kw = dict(input_channels=C, output_channels=C, inner_dim=D, num_heads=H_, mlp_ratio=rng.choice([2, 4]), patch_size=patch,
              classifier_free=True, rope_axes_dim=[16, 24, 24], rope_base=rng.choice([1000, 2000, 10000]))
    x = synth.normal(f"fj.x{i}", (B, C, gh * patch, gw * patch))
    t = synth.uniform(f"fj.t{i}", (B,), lo=0.05, hi=0.95)
    ctx = synth.normal(f"fj.ctx{i}", (B, Lc, Dc)) * 0.5
    lens = [rng.randint(1, Lc) for _ in range(B)]
    lens[rng.randrange(B)] = Lc
    keep = torch.arange(Lc)[None, :] < torch.tensor(lens)[:, None]
    dy = synth.normal(f"fj.dy{i}", tuple(x.shape))
    emb = PrecomputedEmbedder(synth.normal(f"fj.null{i}", (1, Lc, Dc)) * 0.5, null_embedding_seq_len=min(7, Lc))
    scores = None
    if family == "mmdit_joint":
        kw.update(embedding_dim=rng.choice([64, D]), depth=rng.randint(1, 3))
        kw["n_single_stream_blocks"] = rng.randint(0, min(2, kw["depth"]))
        cfg = ommdit.JointConfig(context_dim=Dc, **kw)
        P = synth.dit_params(ommdit.param_shapes(cfg), seed=5000 + i)
        m = MMDiT(simple_dit=False, context_embedder=emb, **kw)
        oracle = lambda Q: ommdit.mmdit_forward(Q, x, t, ctx, keep, cfg)  # noqa: E731
    elif family == "ddt_joint":
        kw.update(encoder_depth=rng.randint(1, 2), decoder_depth=rng.randint(1, 2))
        cfg = oddt.DDTJointConfig(context_dim=Dc, **kw)
        P = synth.dit_params(oddt.joint_param_shapes(cfg), seed=6000 + i)
        m = DDT(simple_ddt=False, context_embedder=emb, **kw)
        oracle = lambda Q: oddt.ddt_joint_forward(Q, x, t, ctx, keep, cfg)  # noqa: E731
    else:
        kw.update(embedding_dim=rng.choice([64, D]), encoder_depth=1, deep_layers_depth=rng.randint(1, 3), decoder_depth=1,
                  drop_rate=rng.choice([0.75, 0.5]))
        kw["n_single_stream_blocks"] = rng.randint(0, kw["deep_layers_depth"])
        cfg = osprint.SprintJointConfig(context_dim=Dc, **kw)
        shapes = osprint.joint_param_shapes(cfg)
        P = synth.dit_params({k: v for k, v in shapes.items() if k != "mask_token"}, seed=7000 + i)
        P["mask_token"] = synth.normal(f"fj.mask{i}", shapes["mask_token"]) * 0.5
        scores = synth.uniform(f"fj.sc{i}", (B, gh * gw))
        kept = osprint.kept_indices(scores, osprint.n_kept(gh * gw, cfg.drop_rate))
        m = SprintDiT(simple_dit=False, context_embedder=emb, **kw)
        oracle = lambda Q: osprint.sprint_mmdit_forward(Q, x, t, ctx, keep, cfg, kept=kept)  # noqa: E731
    tag = f"{family}#{i} {kw} grid={gh}x{gw} Lc={Lc} lens={lens} Dc={Dc} B={B}"
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in P.items()}, tag
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = oracle(Pr)
    (ref * dy).sum().backward()
    try:
        m.load_state_dict(P)
        m = m.to(DEV).train()
        if scores is not None:
            m._draw_scores = lambda B_, S_, device: scores.to(device)
        pred = m(x=x.to(DEV), timesteps=t.to(DEV), initial_context={"embeddings": ctx.to(DEV), "attn_mask": keep.to(DEV)}, p=0.0)["x"]
        (pred * dy.to(DEV)).sum().backward()
        torch.cuda.synchronize()
    except NotImplementedError as e:
        return tag, f"bf16: refused ({str(e)[:120]})"
    live = [(n, p) for n, p in m.named_parameters() if Pr[n].grad is not None]
    for n, p in m.named_parameters():
        if Pr[n].grad is None:
            assert float(p.grad.abs().max()) == 0.0, (tag, n)
    return tag, _verdict("bf16", rel(pred, ref), live, Pr, loose=family == "ddt_joint")


@pytest.mark.timeout(900)
def test_class_conditional_dit_on_arbitrary_token_grids():
    """MMDiT(simple_dit=True) in the bf16 regime on token grids that are not a multiple of 64 (or of 256 above 256): q / k / v rows
    padded to a multiple of 256 per (sample, head), pad keys masked (engine.py `_alloc`: from round 5; it used to refuse and name the
    fp32 regime).  batch * tokens must still be a multiple of 64 (the weight-gradient GEMMs' contraction): a refusal that says so."""
    rng = random.Random("dit-grids")
    bad = []
    for i, (grid, batch) in enumerate((((12, 12), 4), ((28, 36), 4), ((6, 10), 16), ((10, 10), 16), ((20, 24), 2), ((24, 40), 2))):
        tag, out = run_tokens(300 + i, rng, "dit", grid=grid, batch=batch)
        print(tag, "\n    ->", out, flush=True)
        if out.count("pred ") != 2 or "BAD" in out:
            bad.append((tag, out))
    assert not bad, bad
    tag, out = run_tokens(310, rng, "dit", grid=(10, 10), batch=3)
    assert "bf16: refused" in out and "multiple of 64" in out and "fp32: pred" in out, out


@pytest.mark.timeout(900)
@pytest.mark.parametrize("family", ["mmdit_joint", "sprint_joint", "ddt_joint"])
def test_joint_forms_on_multi_aspect_ratio_bucket_grids(family):
    """`ImageNetmultiAR` (datasets/imagenet.py:89-175) batches one aspect-ratio bucket at a time: latent grids like 28 x 36 = 1008 or
    24 x 40 = 960 image tokens, neither a multiple of 256.  The joint attention runs on rows padded to a multiple of 256 with masked
    pad keys; the DDT decoder (image tokens only) does the same from round 5 on.  Prediction and every gradient against the oracle;
    the engines ask for batch * tokens % 64 == 0, which every even batch of these grids satisfies."""
    rng = random.Random(f"multiar-{family}")
    bad = []
    for i, (grid, batch) in enumerate((((28, 36), 4), ((24, 40), 2), ((12, 12), 4), ((20, 48), 2))):
        tag, out = run_joint(100 + i, rng, family, grid=grid, batch=batch)
        print(tag, "\n    ->", out, flush=True)
        if "pred " not in out or "BAD" in out:
            bad.append((tag, out))
    assert not bad, bad


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("family,seed,n", [("unet", 1, 6), ("unet", 7, 6), ("dit", 1, 8), ("ddt", 1, 6), ("sprint", 1, 6),
                                          ("mmdit_joint", 1, 7), ("sprint_joint", 1, 6), ("ddt_joint", 1, 6)])
def test_random_configurations_against_the_oracle(family, seed, n):
    rng = random.Random(f"{family}-{seed}")
    fn = run_unet if family == "unet" else (lambda i, r: run_joint(i, r, family)) if family.endswith("_joint") else (lambda i, r: run_tokens(i, r, family))
    bad, refused, ran = [], 0, 0
    for i in range(n):
        try:
            tag, out = fn(i, rng)
        except Exception as e:  # noqa: BLE001
            tag, out = f"{family}#{i}", "**CRASH** " + "".join(traceback.format_exception_only(type(e), e)).strip()[:300]
        print(tag, "\n    ->", out, flush=True)
        if "CRASH" in out or "BAD" in out:
            bad.append((tag, out))
        refused += out.count("refused")
        ran += out.count("pred ")
    assert not bad, bad
    if family == "unet":
        assert refused == 0  # every UNet configuration runs in both regimes
    elif family.endswith("_joint"):
        assert ran >= n // 2, (ran, refused)  # (bf16 only; token counts the tiled attention does not take refuse by name)
    else:
        assert "fp32: refused" not in out and ran >= n  # the fp32 regime takes every token grid
