"""GPU parity of the SPRINT row (simple_dit form = configs/model/sprint.yaml): the token routing kernels
through the C ABI against torch gather / scatter, and SprintDiT end to end against outputs of the reference's own SprintDiT
(tests/golden/sprint.npz, with its random draws recorded and injected) and the CPU oracle."""

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dit as odit  # noqa: E402
from oracle import sprint as osprint  # noqa: E402
from oracle import synth  # noqa: E402

DEV = "cuda"
KW = dict(input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4, patch_size=2,
          encoder_depth=1, deep_layers_depth=2, decoder_depth=1, n_classes=10, classifier_free=True, drop_rate=0.75)


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def bf(x):
    return x.to(torch.bfloat16)


# ------------------------------------------------------------------ kernels
def test_gather_scatter_restore_kernels():
    """integer index work is bit-exact: gather == torch.gather, scatter-add is its adjoint, restore == scatter into a mask canvas"""
    from diffulab_amd import ops

    B, N, k, D = 3, 256, 64, 128
    cat = bf(synth.normal("tk.cat", (B * N, 2 * D))).to(DEV)  # the kernels read / write column windows of a wider buffer
    scores = synth.uniform("tk.s", (B, N))
    idx = osprint.kept_indices(scores, k)
    idx32 = idx.to(torch.int32).to(DEV).contiguous()
    src = cat[:, D:]
    out = torch.empty(B * k, D, device=DEV, dtype=torch.bfloat16)
    ops.gather_tokens(src, idx32, out, B, N, k, D)
    want = torch.gather(src.view(B, N, D).cpu(), 1, idx[..., None].expand(-1, -1, D))
    assert torch.equal(out.cpu().view(B, k, D), want)
    keep = torch.tensor([1, 0, 1], dtype=torch.int32, device=DEV)
    ops.gather_tokens(src, idx32, out, B, N, k, D, keep=keep)
    assert torch.equal(out.cpu().view(B, k, D)[0], want[0]) and float(out.view(B, k, D)[1].abs().max()) == 0.0
    # restore: canvas of the mask token, kept rows scattered back, sample 1 entirely masked
    inv = torch.full((B, N), -1, dtype=torch.int32)
    inv.scatter_(1, idx, torch.arange(k, dtype=torch.int32).expand(B, k))
    inv[1] = -1
    mask = synth.normal("tk.mask", (D,)).to(DEV)
    xd = bf(synth.normal("tk.xd", (B * k, D))).to(DEV)
    canvas = torch.zeros(B * N, 2 * D, device=DEV, dtype=torch.bfloat16)
    ops.restore_tokens(xd, inv.to(DEV), mask, canvas[:, :D], B, N, k, D)
    ref = bf(mask).cpu().expand(B, N, D).clone()
    ref.scatter_(1, idx[..., None].expand(-1, -1, D), xd.cpu().view(B, k, D))
    ref[1] = bf(mask).cpu()
    assert torch.equal(canvas[:, :D].cpu().view(B, N, D), ref) and float(canvas[:, D:].abs().max()) == 0.0
    # scatter-add (adjoint of the gather) and the mask-token gradient
    dst = bf(synth.normal("tk.dst", (B * N, D))).to(DEV)
    before = dst.clone()
    ops.scatter_tokens_add(xd, idx32, dst, B, N, k, D)
    ref2 = before.float().cpu().view(B, N, D).clone()
    ref2.scatter_add_(1, idx[..., None].expand(-1, -1, D), xd.float().cpu().view(B, k, D))
    assert torch.equal(dst.cpu().view(B, N, D), bf(ref2))
    g = torch.zeros(D, device=DEV)
    ops.masked_colsum(before, inv.to(DEV).view(-1), g, B * N, D)
    assert rel(g, before.float().cpu()[inv.view(-1) < 0].sum(0)) < 1e-5


def test_gated_residual_and_position_indexed_rope():
    from diffulab_amd import ops
    from diffulab_amd.engine import rope_grid_tables

    B, N, k, H, D = 2, 256, 64, 2, 128
    x, t = bf(synth.normal("gr.x", (B * k, D))), bf(synth.normal("gr.t", (B * k, D)))
    gate = bf(synth.normal("gr.g", (B, 3 * D)))
    out = torch.zeros(B * k, 2 * D, device=DEV, dtype=torch.bfloat16)
    ops.gated_residual_fwd(x.to(DEV), t.to(DEV), gate.to(DEV)[:, D : 2 * D], k, out[:, D:])
    want = x.float() + gate.float()[:, D : 2 * D].repeat_interleave(k, 0) * t.float()
    assert rel(out[:, D:].float(), want) < 3e-3 and float(out[:, :D].abs().max()) == 0.0
    # QK-norm + RoPE on a subset of the grid: the position index picks the table rows
    cos, sin = rope_grid_tables(16, 16, [32, 32], 10_000.0)
    idx = osprint.kept_indices(synth.uniform("gr.s", (B, N)), k)
    qkv = bf(synth.normal("gr.qkv", (B * k, 3 * D)))
    sq, sk = 1 + 0.1 * synth.normal("gr.sq", (D,)), 1 + 0.1 * synth.normal("gr.sk", (D,))
    q, kk, v = (torch.empty(B, H, k, 64, device=DEV, dtype=torch.bfloat16) for _ in range(3))
    rrms = torch.empty(B * k, 2, device=DEV)
    pos = idx.to(torch.int32).to(DEV).contiguous().view(-1)
    ops.qk_norm_rope_fwd(qkv.to(DEV), sq.to(DEV), sk.to(DEV), cos.to(DEV), sin.to(DEV), q, kk, v, rrms, B, k, H, 64, 64, pos=pos)
    qf, kf, vf = qkv.float().view(B, k, 3 * D).split(D, dim=-1)
    cd, sd = cos[idx], sin[idx]
    qr = odit.apply_rope(odit.rms_norm(qf, sq).view(B, k, H, 64), cd, sd).transpose(1, 2)
    kr = odit.apply_rope(odit.rms_norm(kf, sk).view(B, k, H, 64), cd, sd).transpose(1, 2)
    assert rel(q.float(), qr) < 4e-3 and rel(kk.float(), kr) < 4e-3
    assert torch.equal(v.cpu(), bf(vf).view(B, k, H, 64).transpose(1, 2))
    # backward against autograd of the same expression
    qkv_r = qkv.float().requires_grad_(True)
    sq_r, sk_r = sq.clone().requires_grad_(True), sk.clone().requires_grad_(True)
    q2, k2, v2 = qkv_r.view(B, k, 3 * D).split(D, dim=-1)
    qo = odit.apply_rope(odit.rms_norm(q2, sq_r).view(B, k, H, 64), cd, sd).transpose(1, 2)
    ko = odit.apply_rope(odit.rms_norm(k2, sk_r).view(B, k, H, 64), cd, sd).transpose(1, 2)
    vo = v2.reshape(B, k, H, 64).transpose(1, 2)
    dq, dk, dv = (bf(synth.normal(f"gr.d{n}", (B, H, k, 64))) for n in "qkv")
    (qo * dq.float()).sum().add((ko * dk.float()).sum()).add((vo * dv.float()).sum()).backward()
    dqkv = torch.empty(B * k, 3 * D, device=DEV, dtype=torch.bfloat16)
    dscale = torch.zeros(2, D, device=DEV)
    ops.qk_norm_rope_bwd(dq.to(DEV), dk.to(DEV), dv.to(DEV), qkv.to(DEV), sq.to(DEV), sk.to(DEV), cos.to(DEV), sin.to(DEV), rrms,
                         dqkv, dscale, B, k, H, 64, 64, pos=pos)
    assert rel(dqkv.float(), qkv_r.grad) < 6e-3
    assert rel(dscale[0], sq_r.grad) < 6e-3 and rel(dscale[1], sk_r.grad) < 6e-3


# ------------------------------------------------------------------ the module
def _model():
    from diffulab_amd import SprintDiT

    cfg = osprint.SprintConfig(**KW)
    shapes = osprint.param_shapes(cfg)
    m = SprintDiT(simple_dit=True, **KW)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == shapes
    P = synth.dit_params({k: v for k, v in shapes.items() if k != "mask_token"}, seed=61)
    P["mask_token"] = synth.normal("sp.mask", shapes["mask_token"]) * 0.5
    m.load_state_dict(P)
    return m.to(DEV), P, cfg


def _inputs():
    B, H = 4, 32
    return (synth.normal("sp.x", (B, 4, H, H)), synth.uniform("sp.t", (B,), lo=0.05, hi=0.95), synth.integers("sp.y", (B,), 10),
            synth.normal("sp.dy", (B, 4, H, H)))


def _inject(m, scores=None, label_u=None, path_u=None):
    if scores is not None:
        m._draw_scores = lambda B, S, device: scores.to(device)
    if label_u is not None:
        m._draw_label_drop = lambda y, p: torch.where(label_u.to(y.device) < p, m.n_classes, y)
    if path_u is not None:
        m._draw_path_drop = lambda B, p, device: path_u.to(device) < p


def test_sprint_training_step_against_reference_fixture(golden):
    """train mode, p = 0: 64 of 256 tokens kept by the recorded scores; prediction and EVERY parameter gradient vs the reference"""
    g = {k: torch.as_tensor(v) for k, v in golden("sprint").items()}
    m, _, _ = _model()
    x, t, y, dy = _inputs()
    m.train()
    _inject(m, scores=g["a_scores"])
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV), p=0.0)["x"]
    assert rel(pred, g["a_pred"]) < 1.5e-2
    (pred * dy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    # the fixture stores every vector and one matrix of each kind; the oracle (pinned by the same fixture) covers every parameter
    _, P0, cfg = _model()
    Pr = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
    (osprint.sprint_forward(Pr, x, t, y, cfg, kept=osprint.kept_indices(g["a_scores"], 64)) * dy).sum().backward()
    bad = []
    for n, p in m.named_parameters():
        tol = 8e-2 if n.endswith(("bias", "scale", "mask_token")) or "norm" in n else 4e-2
        if "a_g_" + n in g and rel(p.grad, g["a_g_" + n]) > tol:
            bad.append((n, "fixture", rel(p.grad, g["a_g_" + n])))
        if rel(p.grad, Pr[n].grad) > tol:
            bad.append((n, "oracle", rel(p.grad, Pr[n].grad)))
    assert not bad, bad


def test_sprint_label_and_path_drop_against_reference_fixture(golden):
    """train mode, p = 0.5: label drop, token drop and the per-sample drop of the deep path (mask-token canvas)"""
    g = {k: torch.as_tensor(v) for k, v in golden("sprint").items()}
    m, _, _ = _model()
    x, t, y, dy = _inputs()
    m.train()
    _inject(m, scores=g["b_scores"], label_u=g["b_label_u"], path_u=g["b_path_u"])
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV), p=0.5)["x"]
    assert rel(pred, g["b_pred"]) < 1.5e-2
    (pred * dy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    sd = dict(m.named_parameters())
    for n in ("mask_token", "fuse.weight", "layers.0.attention.qkv.weight", "deep_layers.1.mlp_input.2.weight",
              "decoder_layers.0.modulation.lin.weight", "label_embed.embedding.weight"):
        assert rel(sd[n].grad, g["b_g_" + n]) < 5e-2, n


def test_sprint_eval_paths_and_guided_sampling_against_reference_fixture(golden):
    """eval: every token goes through the deep blocks (p = 0) or none (p = 1, the unconditional branch); 4-step guided Euler loop"""
    from diffulab_amd import Diffuser

    g = {k: torch.as_tensor(v) for k, v in golden("sprint").items()}
    m, _, _ = _model()
    x, t, y, _ = _inputs()
    m.eval()
    with torch.no_grad():
        assert rel(m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV), p=0.0)["x"], g["c_pred"]) < 1.5e-2
        assert rel(m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV), p=1.0)["x"], g["d_pred"]) < 1.5e-2
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    out = d.generate({"x": synth.normal("sp.init", (4, 4, 32, 32)).to(DEV), "y": y.to(DEV)}, use_tqdm=False, guidance_scale=2.0)
    assert rel(out["x"], g["e_loop_x"]) < 3e-2


def test_sprint_flow_loss_step_with_adamw_learns():
    """the module inside the plugin API: Diffuser.compute_loss -> backward -> FusedAdamW for a few steps on fixed data"""
    from diffulab_amd import Diffuser, SprintDiT
    from diffulab_amd.training import FusedAdamW

    torch.manual_seed(0)
    m = SprintDiT(simple_dit=True, **KW).to(DEV)
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=8)
    opt = FusedAdamW(m.parameters(), lr=2e-3, weight_decay=0.0)
    x0 = synth.normal("sl.x0", (8, 4, 32, 32)).to(DEV)
    y = synth.integers("sl.y", (8,), 10).to(DEV)
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = d.compute_loss({"x": x0.clone(), "y": y, "p": 0.1}, timesteps=d.draw_timesteps(8))["loss"]
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(v == v for v in losses) and sum(losses[-5:]) < sum(losses[:5])
    assert float(m.mask_token.grad.abs().max()) > 0


# ------------------------------------------------------------------ joint text-image form (simple_dit=False)
JKW = dict(input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4, patch_size=1,
           encoder_depth=1, deep_layers_depth=3, n_single_stream_blocks=2, decoder_depth=2, rope_axes_dim=[16, 24, 24], rope_base=2000,
           classifier_free=True, drop_rate=0.75)


def _joint_model(**over):
    from diffulab_amd import SprintDiT
    from diffulab_amd.networks.embedders import PrecomputedEmbedder

    kw = {**JKW, **over}
    null = synth.normal("sj.null", (1, 64, 96)) * 0.5
    m = SprintDiT(simple_dit=False, context_embedder=PrecomputedEmbedder(null, null_embedding_seq_len=7), **kw)
    cfg = osprint.SprintJointConfig(context_dim=96, **kw)
    shapes = osprint.joint_param_shapes(cfg)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == shapes
    P = synth.dit_params({k: v for k, v in shapes.items() if k != "mask_token"}, seed=81)
    P["mask_token"] = synth.normal("sj.mask", shapes["mask_token"]) * 0.5
    m.load_state_dict(P)
    return m.to(DEV), P, cfg


def _joint_inputs():
    B, H, Lc = 4, 16, 64
    keep = torch.arange(Lc)[None, :] < torch.tensor([64, 20, 41, 5])[:, None]
    return (synth.normal("sj.x", (B, 4, H, H)), synth.uniform("sj.t", (B,), lo=0.05, hi=0.95), synth.normal("sj.ctx", (B, Lc, 96)), keep,
            synth.normal("sj.dy", (B, 4, H, H)))


def _check_grads(m, ref: dict, none: set):
    bad = []
    for n, p in m.named_parameters():
        if n in none:  # no gradient in the reference: exact zeros here
            assert float(p.grad.abs().max()) == 0.0, n
            continue
        if n not in ref:
            continue
        e = rel(p.grad, ref[n])
        if e > (8e-2 if p.dim() == 1 or n == "mask_token" else 4e-2):
            bad.append((n, e))
    assert not bad, bad


def test_sprint_joint_training_step_against_reference_fixture_and_oracle(golden):
    """joint encoder block -> 64 of 256 image tokens through one joint + two single-stream deep blocks -> fuse / fuse_context ->
    two joint decoder blocks: prediction vs the reference; gradients vs the reference (stored subset) and vs the oracle (all)"""
    raw = golden("sprint_joint")
    none = set(str(n) for n in raw["a_none"])
    g = {k: torch.as_tensor(v) for k, v in raw.items() if k != "a_none"}
    m, P, cfg = _joint_model()
    x, t, ctx, keep, dy = _joint_inputs()
    m.train()
    _inject(m, scores=g["a_scores"])
    ic = {"embeddings": ctx.to(DEV), "attn_mask": keep.to(DEV)}
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), initial_context=ic, p=0.0)["x"]
    assert rel(pred, g["a_pred"]) < 1.5e-2
    (pred * dy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    _check_grads(m, {n: g["a_g_" + n] for n, _ in m.named_parameters() if "a_g_" + n in g}, none)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    po = osprint.sprint_mmdit_forward(Pr, x, t, ctx, keep, cfg, kept=osprint.kept_indices(g["a_scores"], 64))
    (po * dy).sum().backward()
    _check_grads(m, {n: v.grad for n, v in Pr.items()}, none)


def test_sprint_joint_drops_and_eval_against_reference_fixture(golden):
    from oracle import mmdit as ommdit  # noqa: F401

    raw = golden("sprint_joint")
    g = {k: torch.as_tensor(v) for k, v in raw.items() if k != "a_none"}
    m, _, _ = _joint_model()
    x, t, ctx, keep, _ = _joint_inputs()
    ic = {"embeddings": ctx.to(DEV), "attn_mask": keep.to(DEV)}
    m.train()
    _inject(m, scores=g["b_scores"], path_u=g["b_path_u"])
    m.context_embedder._draw_drop = lambda batch_size, p, device: g["b_ctx_u"].to(device) < p
    with torch.no_grad():
        assert rel(m(x=x.to(DEV), timesteps=t.to(DEV), initial_context=ic, p=0.5)["x"], g["b_pred"]) < 1.5e-2
    del m.context_embedder._draw_drop
    m.eval()
    with torch.no_grad():
        assert rel(m(x=x.to(DEV), timesteps=t.to(DEV), initial_context=ic, p=0.0)["x"], g["c_pred"]) < 1.5e-2
        assert rel(m(x=x.to(DEV), timesteps=t.to(DEV), initial_context=ic, p=1.0)["x"], g["d_pred"]) < 1.5e-2


@pytest.mark.parametrize("ns", [0, 3])
def test_sprint_joint_other_deep_stage_mixes_against_oracle(ns):
    """deep stage of joint blocks only (n_single_stream_blocks = 0) and of single-stream blocks only (the txt-to-img config)"""
    m, P, cfg = _joint_model(n_single_stream_blocks=ns)
    x, t, ctx, keep, dy = _joint_inputs()
    scores = synth.uniform("sj.sc", (4, 256))
    path_u = torch.tensor([0.9, 0.1, 0.8, 0.7])
    m.train()
    _inject(m, scores=scores, path_u=path_u)
    m.context_embedder._draw_drop = lambda batch_size, p, device: torch.zeros(batch_size, dtype=torch.bool, device=device)
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), initial_context={"embeddings": ctx.to(DEV), "attn_mask": keep.to(DEV)}, p=0.5)["x"]
    (pred * dy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    po = osprint.sprint_mmdit_forward(Pr, x, t, ctx, keep, cfg, kept=osprint.kept_indices(scores, 64), path_drop=path_u < 0.5)
    assert rel(pred, po) < 1.5e-2
    (po * dy).sum().backward()
    none = {n for n, v in Pr.items() if v.grad is None}
    _check_grads(m, {n: v.grad for n, v in Pr.items()}, none)


def test_sprint_joint_ragged_context_length_against_oracle():
    """77 text tokens, batch 3, single-stream deep stage on 77 + 64 = 141 latent rows per sample"""
    from diffulab_amd import SprintDiT
    from diffulab_amd.networks.embedders import PrecomputedEmbedder

    Lr, Br = 77, 3
    m = SprintDiT(simple_dit=False, context_embedder=PrecomputedEmbedder(torch.zeros(1, Lr, 96), null_embedding_seq_len=5), **JKW)
    cfg = osprint.SprintJointConfig(context_dim=96, **JKW)
    shapes = osprint.joint_param_shapes(cfg)
    P = synth.dit_params({k: v for k, v in shapes.items() if k != "mask_token"}, seed=83)
    P["mask_token"] = synth.normal("sr.mask", shapes["mask_token"]) * 0.5
    m.load_state_dict(P)
    m = m.to(DEV)
    x, t = synth.normal("sr.x", (Br, 4, 16, 16)), synth.uniform("sr.t", (Br,), lo=0.05, hi=0.95)
    ctx, dy = synth.normal("sr.ctx", (Br, Lr, 96)), synth.normal("sr.dy", (Br, 4, 16, 16))
    keep = torch.arange(Lr)[None, :] < torch.tensor([77, 9, 40])[:, None]
    scores = synth.uniform("sr.sc", (Br, 256))
    m.train()
    _inject(m, scores=scores)
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), initial_context={"embeddings": ctx.to(DEV), "attn_mask": keep.to(DEV)}, p=0.0)["x"]
    (pred * dy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    po = osprint.sprint_mmdit_forward(Pr, x, t, ctx, keep, cfg, kept=osprint.kept_indices(scores, 64))
    assert rel(pred, po) < 1.5e-2
    (po * dy).sum().backward()
    _check_grads(m, {n: v.grad for n, v in Pr.items()}, {n for n, v in Pr.items() if v.grad is None})


def test_sprint_training_step_at_tiled_weight_gradient_dims_against_oracle():
    """256 wide / 4 heads at B = 32 x 256 tokens: the encoder / decoder stages run 8192 token rows and the deep stage 2048 (64 kept
    tokens per sample), so every block's four weight gradients go through ONE atomics-free launch on 256 x 256 tiles (ops.WgradGroups,
    the form the 512-wide configurations train with) -- the fixture dims above (128 wide) never reach it.  Prediction and every
    parameter gradient against the fp32 oracle."""
    from diffulab_amd import SprintDiT

    kw = dict(KW, inner_dim=256, embedding_dim=128, num_heads=4)
    cfg = osprint.SprintConfig(**kw)
    shapes = osprint.param_shapes(cfg)
    P0 = synth.dit_params({k: v for k, v in shapes.items() if k != "mask_token"}, seed=67)
    P0["mask_token"] = synth.normal("sp.mask2", shapes["mask_token"]) * 0.5
    m = SprintDiT(simple_dit=True, **kw)
    m.load_state_dict(P0)
    m = m.to(DEV)
    B, H = 32, 32
    x, t, y = synth.normal("sq.x", (B, 4, H, H)), synth.uniform("sq.t", (B,), lo=0.05, hi=0.95), synth.integers("sq.y", (B,), 10)
    dy = synth.normal("sq.dy", (B, 4, H, H))
    scores = synth.normal("sq.scores", (B, 256))
    m.train()
    _inject(m, scores=scores)
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV), p=0.0)["x"]
    (pred * dy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    assert m.engine.ws.get("tn_slab") is not None
    Pr = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
    ref = osprint.sprint_forward(Pr, x, t, y, cfg, kept=osprint.kept_indices(scores, 64))
    (ref * dy).sum().backward()
    assert rel(pred, ref) < 1.5e-2
    bad = []
    for n, p in m.named_parameters():
        tol = 8e-2 if n.endswith(("bias", "scale", "mask_token")) or "norm" in n else 4e-2
        if rel(p.grad, Pr[n].grad) > tol:
            bad.append((n, rel(p.grad, Pr[n].grad)))
    assert not bad, bad
