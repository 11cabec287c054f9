"""GPU parity of the joint text-image MMDiT row: MMDiT(simple_dit=False) behind a PrecomputedEmbedder on
the HIP path against outputs of the reference module (tests/golden/mmdit_joint.npz) and the CPU oracle."""

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dit as odit  # noqa: E402
from oracle import mmdit as ommdit  # noqa: E402
from oracle import synth  # noqa: E402

DEV = "cuda"
KW = dict(input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4, patch_size=2, depth=2,
          rope_axes_dim=[16, 24, 24], rope_base=2000, classifier_free=True)
Lc, Cd, B, H = 64, 96, 4, 16


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def bf(x):
    return x.to(torch.bfloat16)


def test_qk_norm_rope_row_window_and_copy_rows3d():
    """each stream writes its window of the joint q / k / v buffers; the joint attention output is sliced back per stream"""
    from diffulab_amd import ops
    from diffulab_amd.mmdit_engine import joint_rope_tables

    Bq, Hh, D, n_ctx, gh = 2, 2, 128, 64, 8
    N, Tp = gh * gh, 256
    cos, sin = joint_rope_tables(n_ctx, gh, gh, [16, 24, 24], 2000.0)
    co, so = ommdit.rope_tables_joint(n_ctx, gh, gh, [16, 24, 24], 2000.0)
    assert torch.equal(cos, co) and torch.equal(sin, so)
    q, k, v = (torch.zeros(Bq, Hh, Tp, 64, device=DEV, dtype=torch.bfloat16) for _ in range(3))
    want = {}
    for name, nt, off in (("c", n_ctx, 0), ("x", N, n_ctx)):
        qkv = bf(synth.normal("jw.qkv" + name, (Bq * nt, 3 * D)))
        sq, sk = 1 + 0.1 * synth.normal("jw.sq" + name, (D,)), 1 + 0.1 * synth.normal("jw.sk" + name, (D,))
        rr = torch.empty(Bq * nt, 2, device=DEV)
        ops.qk_norm_rope_fwd(qkv.to(DEV), sq.to(DEV), sk.to(DEV), cos.to(DEV)[off:], sin.to(DEV)[off:], q, k, v, rr, Bq, nt, Hh, 64, 64,
                             n_off=off)
        qf, kf, vf = qkv.float().view(Bq, nt, 3 * D).split(D, dim=-1)
        want[name] = (odit.apply_rope(odit.rms_norm(qf, sq).view(Bq, nt, Hh, 64), cos[off : off + nt], sin[off : off + nt]),
                      odit.apply_rope(odit.rms_norm(kf, sk).view(Bq, nt, Hh, 64), cos[off : off + nt], sin[off : off + nt]),
                      vf.reshape(Bq, nt, Hh, 64))
    for j, buf in enumerate((q, k, v)):
        ref = torch.cat((want["c"][j], want["x"][j]), 1).transpose(1, 2)
        assert rel(buf[:, :, : n_ctx + N].float(), ref) < 4e-3
        assert float(buf[:, :, n_ctx + N :].abs().max()) == 0.0
    # slice a [B, Tp, D] joint buffer into a contiguous per-stream buffer and back
    joint = bf(synth.normal("jw.j", (Bq * Tp, D))).to(DEV)
    img = torch.empty(Bq * N, D, device=DEV, dtype=torch.bfloat16)
    ops.copy_rows3d(joint[n_ctx:], Tp * D, D, img, N * D, D, Bq, N, D)
    assert torch.equal(img.view(Bq, N, D), joint.view(Bq, Tp, D)[:, n_ctx : n_ctx + N])
    back = torch.zeros_like(joint)
    ops.copy_rows3d(img, N * D, D, back[n_ctx:], Tp * D, D, Bq, N, D)
    assert torch.equal(back.view(Bq, Tp, D)[:, n_ctx : n_ctx + N], img.view(Bq, N, D))
    assert float(back.view(Bq, Tp, D)[:, :n_ctx].abs().max()) == 0.0 and float(back.view(Bq, Tp, D)[:, n_ctx + N :].abs().max()) == 0.0


def _model():
    from diffulab_amd import MMDiT
    from diffulab_amd.networks.embedders import PrecomputedEmbedder

    null = synth.normal("mj.null", (1, Lc, Cd)) * 0.5
    m = MMDiT(simple_dit=False, context_embedder=PrecomputedEmbedder(null, null_embedding_seq_len=7), **KW)
    cfg = ommdit.JointConfig(context_dim=Cd, **KW)
    shapes = ommdit.param_shapes(cfg)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == shapes
    m.load_state_dict(synth.dit_params(shapes, seed=71))
    return m.to(DEV)


def _inputs():
    keep = torch.arange(Lc)[None, :] < torch.tensor([64, 20, 41, 5])[:, None]
    return (synth.normal("mj.x", (B, 4, H, H)), synth.uniform("mj.t", (B,), lo=0.05, hi=0.95), synth.normal("mj.ctx", (B, Lc, Cd)), keep,
            synth.normal("mj.dy", (B, 4, H, H)))


def test_joint_mmdit_training_step_against_reference_fixture(golden):
    """ragged key-padding mask (64 / 20 / 41 / 5 valid text tokens): prediction and every parameter gradient vs the reference; the
    context branch of the last block has no gradient there and exact zeros here"""
    g = {k: torch.as_tensor(v) for k, v in golden("mmdit_joint").items()}
    m = _model()
    x, t, ctx, keep, dy = _inputs()
    m.train()
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), initial_context={"embeddings": ctx.to(DEV), "attn_mask": keep.to(DEV)}, p=0.0)["x"]
    assert rel(pred, g["a_pred"]) < 1.5e-2
    (pred * dy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    bad = []
    for n, p in m.named_parameters():
        if "a_g_" + n in g:
            e = rel(p.grad, g["a_g_" + n])
            if e > (8e-2 if p.dim() == 1 else 4e-2):
                bad.append((n, e))
        else:
            assert n.startswith("layers.1.") and "context" in n and float(p.grad.abs().max()) == 0.0, n
    assert not bad, bad


def test_joint_mmdit_context_drop_and_guided_sampling_against_reference_fixture(golden):
    from diffulab_amd import Diffuser

    g = {k: torch.as_tensor(v) for k, v in golden("mmdit_joint").items()}
    m = _model()
    x, t, ctx, keep, _ = _inputs()
    m.context_embedder._draw_drop = lambda batch_size, p, device: g["b_u"].to(device) < p
    m.train()
    with torch.no_grad():
        pred = m(x=x.to(DEV), timesteps=t.to(DEV), initial_context={"embeddings": ctx.to(DEV), "attn_mask": keep.to(DEV)}, p=0.5)["x"]
    assert rel(pred, g["b_pred"]) < 1.5e-2
    del m.context_embedder._draw_drop
    m.eval()
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    out = d.generate({"x": synth.normal("mj.init", (B, 4, H, H)).to(DEV),
                      "initial_context": {"embeddings": ctx.to(DEV), "attn_mask": keep.to(DEV)}}, use_tqdm=False, guidance_scale=2.0)
    assert rel(out["x"], g["e_loop_x"]) < 3e-2


def test_joint_mmdit_ragged_context_length_against_oracle():
    """77 text tokens, batch 3 (rows of the context stream are not a multiple of 64: zero-padded row buffers feed the weight-gradient
    GEMMs), no attention mask: prediction and every gradient against the oracle"""
    from diffulab_amd import MMDiT
    from diffulab_amd.networks.embedders import PrecomputedEmbedder

    Lr, Br = 77, 3
    m = MMDiT(simple_dit=False, context_embedder=PrecomputedEmbedder(torch.zeros(1, Lr, Cd), null_embedding_seq_len=5), **KW)
    cfg = ommdit.JointConfig(context_dim=Cd, **KW)
    P = synth.dit_params(ommdit.param_shapes(cfg), seed=73)
    m.load_state_dict(P)
    m = m.to(DEV)
    x, t = synth.normal("mr.x", (Br, 4, H, H)), synth.uniform("mr.t", (Br,), lo=0.05, hi=0.95)
    ctx, dy = synth.normal("mr.ctx", (Br, Lr, Cd)), synth.normal("mr.dy", (Br, 4, H, H))
    keep = torch.ones(Br, Lr, dtype=torch.bool)
    m.train()
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), initial_context={"embeddings": ctx.to(DEV), "attn_mask": keep.to(DEV)}, p=0.0)["x"]
    (pred * dy.to(DEV)).sum().backward()
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    po = ommdit.mmdit_forward(Pr, x, t, ctx, keep, cfg)
    assert rel(pred, po) < 1.5e-2
    (po * dy).sum().backward()
    bad = []
    for n, p in m.named_parameters():
        if Pr[n].grad is None:
            assert float(p.grad.abs().max()) == 0.0, n
        elif rel(p.grad, Pr[n].grad) > (8e-2 if p.dim() == 1 else 4e-2):
            bad.append((n, rel(p.grad, Pr[n].grad)))
    assert not bad, bad


@pytest.mark.parametrize("depth,ns", [(3, 2), (2, 2)])
def test_mmdit_with_single_stream_blocks(golden, depth, ns):
    """MMDiT(simple_dit=False, n_single_stream_blocks > 0): joint blocks then single-stream blocks on [context ; image] (also a
    stack of single-stream blocks only): prediction vs the reference fixture (depth 3), every gradient vs the oracle"""
    from diffulab_amd import MMDiT
    from diffulab_amd.networks.embedders import PrecomputedEmbedder

    kw = dict(KW, depth=depth, n_single_stream_blocks=ns)
    m = MMDiT(simple_dit=False, context_embedder=PrecomputedEmbedder(torch.zeros(1, Lc, Cd), null_embedding_seq_len=7), **kw)
    cfg = ommdit.JointConfig(context_dim=Cd, **kw)
    shapes = ommdit.param_shapes(cfg)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == shapes
    P = synth.dit_params(shapes, seed=111)
    m.load_state_dict(P)
    m = m.to(DEV)
    x, t = synth.normal("ms.x", (B, 4, H, H)), synth.uniform("ms.t", (B,), lo=0.05, hi=0.95)
    ctx, dy = synth.normal("ms.ctx", (B, Lc, Cd)), synth.normal("ms.dy", (B, 4, H, H))
    keep = torch.arange(Lc)[None, :] < torch.tensor([64, 20, 41, 5])[:, None]
    m.train()
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), initial_context={"embeddings": ctx.to(DEV), "attn_mask": keep.to(DEV)}, p=0.0)["x"]
    if depth == 3:
        g = {k: torch.as_tensor(v) for k, v in golden("mmdit_single").items() if k != "none"}
        assert rel(pred, g["pred"]) < 1.5e-2
    (pred * dy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    po = ommdit.mmdit_forward(Pr, x, t, ctx, keep, cfg)
    assert rel(pred, po) < 1.5e-2
    (po * dy).sum().backward()
    bad = []
    for n, p in m.named_parameters():
        if Pr[n].grad is None:
            assert float(p.grad.abs().max()) == 0.0, n
        elif rel(p.grad, Pr[n].grad) > (8e-2 if p.dim() == 1 else 4e-2):
            bad.append((n, rel(p.grad, Pr[n].grad)))
    assert not bad, bad
    m.eval()
    with torch.no_grad():  # eager and hipGraph-replayed inference agree
        a1 = m(x=x.to(DEV), timesteps=t.to(DEV), initial_context={"embeddings": ctx.to(DEV), "attn_mask": keep.to(DEV)}, p=0.0)["x"].clone()
        a2 = m(x=x.to(DEV), timesteps=t.to(DEV), initial_context={"embeddings": ctx.to(DEV), "attn_mask": keep.to(DEV)}, p=0.0)["x"]
    assert rel(a1, po) < 1.5e-2 and rel(a2, a1) < 1e-6
