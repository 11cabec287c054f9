"""The fp32-class regime of SprintDiT(simple_dit=True) -- the precision `model=sprint` gets in the reference's class-conditional
configurations (they inherit trainer/default.yaml's precision_type "no"; configs/train_cifar10_sprint.yaml here) -- through the C ABI: the f32 token-routing kernels of csrc/f32.hip and the launch
sequences of sprint_engine_f32.py against (1) outputs of the reference's own SprintDiT (tests/golden/sprint.npz, random draws recorded
and injected) and (2) the CPU oracle.  Bar: per-tensor relative L2 <= 1e-5."""

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sprint as osprint  # noqa: E402
from oracle import synth  # noqa: E402

DEV = "cuda"
TOL = 1e-5
KW = dict(input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4, patch_size=2,
          encoder_depth=1, deep_layers_depth=2, decoder_depth=1, n_classes=10, classifier_free=True, drop_rate=0.75)


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


# ------------------------------------------------------------------ kernels
def test_f32_token_routing_kernels_against_torch():
    from diffulab_amd import ops

    B, N, k, D = 3, 64, 16, 96
    g = torch.Generator().manual_seed(3)
    src = torch.randn(B * N, 2 * D, generator=g)  # the encoder output sits in the right half of a [M, 2D] buffer
    idx = torch.stack([torch.randperm(N, generator=g)[:k].sort().values for _ in range(B)]).to(torch.int32)
    keep = torch.tensor([1, 0, 1], dtype=torch.int32)
    sd, idd = src.to(DEV), idx.to(DEV)
    # gather (bytewise through dl_gather_tokens), with and without the per-sample keep flags
    out = torch.empty(B * k, D, device=DEV)
    ops.f32_gather_tokens(sd[:, D:], idd, out, B, N, k, D)
    ref = torch.gather(src[:, D:].reshape(B, N, D), 1, idx.long()[..., None].expand(B, k, D))
    assert torch.equal(out.cpu().view(B, k, D), ref)
    ops.f32_gather_tokens(sd[:, D:], idd, out, B, N, k, D, keep=keep.to(DEV))
    assert torch.equal(out.cpu().view(B, k, D), ref * keep.view(B, 1, 1))
    # scatter-add = adjoint of the gather
    dst = torch.randn(B * N, D, generator=g)
    upd = torch.randn(B * k, D, generator=g)
    dd = dst.to(DEV)
    ops.f32_scatter_tokens_add(upd.to(DEV), idd, dd, B, N, k, D)
    want = dst.view(B, N, D).clone()
    want.scatter_add_(1, idx.long()[..., None].expand(B, k, D), upd.view(B, k, D))
    assert torch.equal(dd.cpu().view(B, N, D), want)
    # restore into a mask-token canvas (strided destination) + the mask token's gradient
    inv = torch.full((B, N), -1, dtype=torch.int32)
    inv.scatter_(1, idx.long(), torch.arange(k, dtype=torch.int32).expand(B, k))
    inv[1] = -1  # a sample whose deep path was dropped
    mask = torch.randn(D, generator=g)
    canvas = torch.zeros(B * N, 2 * D, device=DEV)
    ops.f32_restore_tokens(upd.to(DEV), inv.to(DEV), mask.to(DEV), canvas[:, :D], B, N, k, D)
    want = torch.where(inv.view(B, N, 1) >= 0, torch.gather(upd.view(B, k, D), 1, inv.clamp_min(0).long()[..., None].expand(B, N, D)),
                       mask.view(1, 1, D))
    assert torch.equal(canvas[:, :D].cpu().view(B, N, D), want) and canvas[:, D:].abs().max().item() == 0
    x = torch.randn(B * N, D, generator=g)
    gm = torch.ones(D, device=DEV)
    scr = torch.empty(1 << 16, device=DEV)
    ops.f32_masked_colsum(x.to(DEV), inv.view(-1).to(DEV), gm, B * N, D, scr)
    assert rel(gm.cpu() - 1, (x * (inv.view(-1, 1) < 0)).sum(0)) < 2e-6
    g2 = torch.ones(D, device=DEV)
    ops.f32_masked_colsum(x.to(DEV), inv.view(-1).to(DEV), g2, B * N, D, scr)
    assert torch.equal(gm, g2)  # no atomics


def test_f32_gated_residual_and_position_indexed_rope_against_torch():
    from diffulab_amd import ops
    from diffulab_amd.engine import rope_grid_tables
    from oracle import dit as odit

    B, nt, D, H = 3, 16, 128, 2
    g = torch.Generator().manual_seed(4)
    x, t = torch.randn(B * nt, D, generator=g), torch.randn(B * nt, D, generator=g)
    gate = torch.randn(B, 3 * D, generator=g)[:, D : 2 * D]  # a column window of the modulation matrix
    xr, tr, gr = x.clone().requires_grad_(True), t.clone().requires_grad_(True), gate.clone().requires_grad_(True)
    yr = xr.view(B, nt, D) + gr.view(B, 1, D) * tr.view(B, nt, D)
    dy = torch.randn(B * nt, D, generator=g)
    (yr.reshape(-1, D) * dy).sum().backward()
    gd = torch.randn(B, 3 * D, generator=g).to(DEV)
    gd[:, D : 2 * D] = gate.to(DEV)
    out = torch.zeros(B * nt, 2 * D, device=DEV)
    ops.f32_gated_residual_fwd(x.to(DEV), t.to(DEV), gd[:, D : 2 * D], nt, out[:, D:])
    assert rel(out[:, D:], yr.reshape(-1, D)) < 1e-6 and out[:, :D].abs().max().item() == 0
    dt, dg = torch.empty(B * nt, D, device=DEV), torch.zeros(B, 3 * D, device=DEV)
    ops.f32_gate_bwd(dy.to(DEV), t.to(DEV), gd[:, D : 2 * D], nt, dt, dg[:, D : 2 * D])
    assert rel(dt, tr.grad) < 1e-6 and rel(dg[:, D : 2 * D], gr.grad) < 2e-6 and dg[:, :D].abs().max().item() == 0

    # QK-norm + RoPE with the rotary table row picked per token (the kept tokens of a 8x8 grid), forward and backward
    gh = gw = 8
    N = gh * gw
    cos, sin = rope_grid_tables(gh, gw, [32, 32], 10_000.0)
    pos = torch.stack([torch.randperm(N, generator=g)[:nt].sort().values for _ in range(B)]).to(torch.int32)
    qkv = torch.randn(B * nt, 3 * D, generator=g)
    sq, sk = 1 + 0.1 * torch.randn(D, generator=g), 1 + 0.1 * torch.randn(D, generator=g)
    qkv_r, sq_r, sk_r = qkv.clone().requires_grad_(True), sq.clone().requires_grad_(True), sk.clone().requires_grad_(True)
    cd, sd = cos[pos.long().view(-1)].view(B, nt, -1), sin[pos.long().view(-1)].view(B, nt, -1)
    q2, k2, _ = qkv_r.view(B, nt, 3 * D).split(D, dim=-1)
    qo = odit.apply_rope(odit.rms_norm(q2, sq_r).view(B, nt, H, 64), cd, sd)
    ko = odit.apply_rope(odit.rms_norm(k2, sk_r).view(B, nt, H, 64), cd, sd)
    dqk = torch.randn(B * nt, 2 * D, generator=g)
    ((qo.reshape(B * nt, D) * dqk[:, :D]).sum() + (ko.reshape(B * nt, D) * dqk[:, D:]).sum()).backward()
    qk, rr = torch.empty(B * nt, 2 * D, device=DEV), torch.empty(B * nt, 2, device=DEV)
    pd = pos.view(-1).to(DEV)
    ops.f32_qk_norm_rope_fwd(qkv.to(DEV), sq.to(DEV), sk.to(DEV), cos.to(DEV), sin.to(DEV), qk, rr, B, nt, H, 64, 64, pos=pd)
    assert rel(qk[:, :D], qo.reshape(B * nt, D)) < 2e-6 and rel(qk[:, D:], ko.reshape(B * nt, D)) < 2e-6
    dqkv, part = torch.zeros(B * nt, 3 * D, device=DEV), torch.empty(B, 2, D, device=DEV)
    ops.f32_qk_norm_rope_bwd(dqk.to(DEV), qkv.to(DEV), sq.to(DEV), sk.to(DEV), cos.to(DEV), sin.to(DEV), rr, dqkv, part, B, nt, H, 64, 64,
                             pos=pd)
    assert rel(dqkv[:, : 2 * D], qkv_r.grad[:, : 2 * D]) < 3e-6
    assert rel(part.sum(0)[0], sq_r.grad) < 3e-6 and rel(part.sum(0)[1], sk_r.grad) < 3e-6


# ------------------------------------------------------------------ the module
def _model():
    from diffulab_amd import SprintDiT

    cfg = osprint.SprintConfig(**KW)
    shapes = osprint.param_shapes(cfg)
    m = SprintDiT(simple_dit=True, **KW)
    P = synth.dit_params({k: v for k, v in shapes.items() if k != "mask_token"}, seed=61)
    P["mask_token"] = synth.normal("sp.mask", shapes["mask_token"]) * 0.5
    m.load_state_dict(P)
    m.set_precision("fp32")
    m = m.to(DEV)
    assert m.precision == "fp32" and type(m.engine).__name__ == "SprintEngineF32"
    return m, P, cfg


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


def test_sprint_fp32_training_step_against_reference_fixture_and_oracle(golden):
    """train mode, p = 0: 64 of 256 tokens kept by the recorded scores; prediction and EVERY parameter gradient"""
    g = {k: torch.as_tensor(v) for k, v in golden("sprint").items()}
    m, P0, cfg = _model()
    x, t, y, dy = _inputs()
    m.train()
    _inject(m, scores=g["a_scores"])
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV), p=0.0)["x"]
    assert rel(pred, g["a_pred"]) < TOL
    (pred * dy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    Pr = {k: v.clone().requires_grad_(True) for k, v in P0.items()}
    (osprint.sprint_forward(Pr, x, t, y, cfg, kept=osprint.kept_indices(g["a_scores"], 64)) * dy).sum().backward()
    errs = {}
    for n, p in m.named_parameters():
        if "a_g_" + n in g:
            errs[n + " (fixture)"] = rel(p.grad, g["a_g_" + n])
        errs[n + " (oracle)"] = rel(p.grad, Pr[n].grad)
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print("SPRINT fp32 regime, largest per-tensor gradient errors:", top)
    assert top[0][1] < TOL, top
    # bit-reproducible
    g1 = m._flat_grad.clone()
    m.zero_grad()
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV), p=0.0)["x"]
    (pred * dy.to(DEV)).sum().backward()
    assert torch.equal(g1, m._flat_grad)


def test_sprint_fp32_label_and_path_drop_against_reference_fixture(golden):
    """train mode, p = 0.5: label drop, token drop and the per-sample drop of the deep path (mask-token canvas)"""
    g = {k: torch.as_tensor(v) for k, v in golden("sprint").items()}
    m, _, _ = _model()
    x, t, y, dy = _inputs()
    m.train()
    _inject(m, scores=g["b_scores"], label_u=g["b_label_u"], path_u=g["b_path_u"])
    pred = m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV), p=0.5)["x"]
    assert rel(pred, g["b_pred"]) < TOL
    (pred * dy.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    sd = dict(m.named_parameters())
    for n in ("mask_token", "fuse.weight", "layers.0.attention.qkv.weight", "deep_layers.1.mlp_input.2.weight",
              "decoder_layers.0.modulation.lin.weight", "label_embed.embedding.weight"):
        assert rel(sd[n].grad, g["b_g_" + n]) < TOL, (n, rel(sd[n].grad, g["b_g_" + n]))


def test_sprint_fp32_eval_paths_and_guided_sampling_against_reference_fixture(golden):
    """eval: every token goes through the deep blocks (p = 0) or none (p = 1, the unconditional branch); 4-step guided Euler loop"""
    from diffulab_amd import Diffuser

    g = {k: torch.as_tensor(v) for k, v in golden("sprint").items()}
    m, _, _ = _model()
    x, t, y, _ = _inputs()
    m.eval()
    with torch.no_grad():
        assert rel(m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV), p=0.0)["x"], g["c_pred"]) < TOL
        assert rel(m(x=x.to(DEV), timesteps=t.to(DEV), y=y.to(DEV), p=1.0)["x"], g["d_pred"]) < TOL
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    out = d.generate({"x": synth.normal("sp.init", (4, 4, 32, 32)).to(DEV), "y": y.to(DEV)}, use_tqdm=False, guidance_scale=2.0)
    assert rel(out["x"], g["e_loop_x"]) < TOL


def test_sprint_fp32_trainer_default_precision_learns(tmp_path):
    """BaseTrainer with the reference's default precision_type ("no") picks the fp32 launch sequences of SprintDiT and the loss falls"""
    from diffulab_amd import Diffuser, SprintDiT
    from diffulab_amd.training import FusedAdamW

    torch.manual_seed(0)
    m = SprintDiT(simple_dit=True, **KW).to(DEV)
    m.set_precision("fp32")
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=10)
    opt = FusedAdamW(m.parameters(), lr=2e-3, weight_decay=0.0)
    x0, y = synth.normal("spl.x", (16, 4, 32, 32)).to(DEV), synth.integers("spl.y", (16,), 10).to(DEV)
    losses = []
    for _ in range(12):
        opt.zero_grad()
        loss = d.compute_loss({"x": x0, "y": y, "p": 0.1}, timesteps=d.draw_timesteps(16).to(DEV))["loss"]
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert type(m.engine).__name__ == "SprintEngineF32" and losses[-1] < 0.9 * losses[0], losses
