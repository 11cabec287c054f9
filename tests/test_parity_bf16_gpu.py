"""Parity of the HIP path against the fp32 oracle, HELD TO THE bf16 REGIME'S OWN ERROR.

The HIP denoisers compute in bf16 with f32 accumulation -- the precision of the reference under accelerate's bf16 mixed precision.
Every test here runs three legs on the same seeded inputs and weights:

    fp32  : the oracle in fp32 (pinned to the reference by tests/golden/*.npz)          -> the expected values
    bf16  : the SAME oracle under ``odit.bf16_autocast()`` (bf16 matmuls / residual stream, f32 norm statistics)
    hip   : the product path through the C ABI

(round 5: the bf16 leg is itself pinned -- tests/golden/dit_autocast.npz holds the per-tensor error of the IMPORTED REFERENCE under
``torch.autocast("cpu", bfloat16)``; tests/test_oracle_golden.py shows the oracle's leg loses 0.57 ... 1.0x of that per tensor, i.e. it is
the stricter yardstick, and ``test_hip_error_within_the_reference_s_own_autocast_error`` bounds the HIP path by the reference's own
numbers on the fixture inputs)

and asserts, per tensor, ``err(hip, fp32) <= max(SURVEY's bound, 1.5 * err(bf16, fp32))``: SURVEY §8c's 1e-2 (gradients /
activations) and 1e-3 (loss) hold wherever a bf16 pipeline can meet them at all, and where bf16 itself cannot (deep stacks: the
autocast leg alone is 1.4e-2 median on DiT-S/2 gradients) the HIP path may not be more than 1.5x worse than it.
"""

import time

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import diffusion as od  # noqa: E402
from oracle import dit as odit  # noqa: E402
from oracle import synth  # noqa: E402

DEV = "cuda"
S2 = dict(input_channels=4, output_channels=4, inner_dim=384, embedding_dim=384, num_heads=6, mlp_ratio=4, patch_size=2, depth=12,
          n_classes=1000, classifier_free=True)
SMALL = dict(input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4, patch_size=2, depth=2,
             n_classes=10, classifier_free=True)
GRAD_BOUND, LOSS_BOUND, FACTOR = 1e-2, 1e-3, 1.5


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def three_legs(cfgk, seed, B, H, tag, head="flow", legs=("fp32", "bf16"), chunk=None):
    """-> (loss, grads) of the fp32 oracle, the bf16-autocast oracle and the HIP path on identical inputs.  chunk: the oracle legs
    run the batch in pieces of `chunk` samples and accumulate (the loss is a mean over samples and every sample's path through the
    network is independent of its batch-mates, so loss and gradients are the batch's up to f32 summation order) -- at B = 256 the
    one-piece autograd graph is ~100 GB of host memory and runs 7x slower per image than pieces of 32"""
    from diffulab_amd import Diffuser, MMDiT

    cfg = odit.DiTConfig(**cfgk)
    P = synth.dit_params(odit.param_shapes(cfg), seed=seed)
    C = cfgk["input_channels"]
    x0, noise = synth.normal(f"{tag}.x0", (B, C, H, H)), synth.normal(f"{tag}.noise", (B, C, H, H))
    t = synth.uniform(f"{tag}.t", (B,), lo=0.05, hi=0.95)
    y = synth.integers(f"{tag}.y", (B,), cfgk["n_classes"])
    out = {}
    for leg in legs:
        Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        t0 = time.time()
        step = chunk or B
        assert B % step == 0
        total = 0.0
        for lo in range(0, B, step):
            sl = slice(lo, lo + step)
            xs, ns, ts, ys = x0[sl], noise[sl], t[sl], y[sl]
            if leg == "bf16":
                with odit.bf16_autocast():
                    loss = od.flow_loss(odit.dit_forward(Pr, od.flow_add_noise(xs, ts, ns), ts, ys, cfg), xs, ns)
            else:
                loss = od.flow_loss(odit.dit_forward(Pr, od.flow_add_noise(xs, ts, ns), ts, ys, cfg), xs, ns)
            loss = loss * (step / B)
            loss.backward()  # (gradients accumulate in Pr[k].grad)
            total += loss.item()

        class _L:  # (what the callers read of the loss)
            def __init__(self, v):
                self.v = v

            def item(self):
                return self.v

        loss = _L(total)
        out[leg] = (loss.item(), {k: v.grad for k, v in Pr.items()})
        print(f"{tag}: {leg} oracle leg {time.time() - t0:.1f} s")
    m = MMDiT(simple_dit=True, **cfgk)
    m.load_state_dict(P)
    m = m.to(DEV)
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=50)
    loss = d.compute_loss({"x": x0.to(DEV), "y": y.to(DEV), "p": 0.0}, timesteps=t, noise=noise.to(DEV))["loss"]
    loss.backward()
    torch.cuda.synchronize()
    out["hip"] = (loss.item(), {k: p.grad.detach().cpu() for k, p in m.named_parameters()})
    return out


def check(out, tag, bf16_fixture=None):
    """bf16_fixture: {"loss": the bf16 leg's loss, "err": {tensor: its relative L2 error against the fp32 leg}} recorded once for
    these seeded inputs (tests/golden/make_b256_autocast.py) instead of running the bf16 leg here"""
    l32, g32 = out["fp32"]
    l_bf = out["bf16"][0] if bf16_fixture is None else bf16_fixture["loss"]
    e_loss_bf, e_loss_hip = abs(l_bf - l32) / l32, abs(out["hip"][0] - l32) / l32
    assert e_loss_hip <= max(LOSS_BOUND, FACTOR * e_loss_bf), (tag, e_loss_hip, e_loss_bf)
    rows, relaxed = [], 0
    for k in g32:
        e_bf = rel(out["bf16"][1][k], g32[k]) if bf16_fixture is None else bf16_fixture["err"][k]
        e_hip = rel(out["hip"][1][k], g32[k])
        bound = max(GRAD_BOUND, FACTOR * e_bf)
        relaxed += bound > GRAD_BOUND
        rows.append((e_hip / bound, k, e_hip, e_bf))
    rows.sort(reverse=True)
    med = sorted(r[2] for r in rows)[len(rows) // 2], sorted(r[3] for r in rows)[len(rows) // 2]
    print(f"{tag}: loss err hip {e_loss_hip:.2e} / bf16-autocast {e_loss_bf:.2e};  gradient rel-L2 median hip {med[0]:.2e} / bf16-autocast "
          f"{med[1]:.2e};  {relaxed} of {len(rows)} tensors need more than {GRAD_BOUND:g} (bf16 itself misses it);  tightest: "
          + ", ".join(f"{k} hip {eh:.2e} bf16 {eb:.2e}" for _, k, eh, eb in rows[:4]))
    bad = [(k, eh, eb) for ratio, k, eh, eb in rows if ratio > 1.0]
    assert not bad, (tag, bad[:8])


def test_small_dit_hip_error_within_the_bf16_regime():
    check(three_legs(SMALL, seed=5, B=4, H=16, tag="pb.small"), "small DiT (2 blocks, 64 tokens)")


def test_dit_s2_small_batch_hip_error_within_the_bf16_regime():
    check(three_legs(S2, seed=7, B=4, H=32, tag="pb.s2"), "DiT-S/2 B=4")


@pytest.mark.parametrize("tag,cfgk,seed,B,H,lo", [("s16", SMALL, 5, 4, 16, 0.02), ("s2", S2, 7, 2, 32, 0.05)])
def test_hip_error_within_the_reference_s_own_autocast_error(tag, cfgk, seed, B, H, lo):
    """the yardstick pinned to the REFERENCE: tests/golden/dit_autocast.npz holds, per parameter, what the imported
    reference loses under ``torch.autocast("cpu", bfloat16)`` against its own fp32 run on the inputs of the dit_small16 / dit_s2
    fixtures.  The HIP path on the same inputs, against the fp32 oracle (pinned to the reference's fp32 run by those fixtures), may
    be at most max(SURVEY's 1e-2, 1.5 x THAT error) off per tensor -- no builder-authored leg on either side of the bound."""
    import os

    import numpy as np

    from diffulab_amd import Diffuser, MMDiT

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dit_autocast.npz"))
    ref_err = dict(zip(g[f"{tag}_names"].tolist(), g[f"{tag}_err"].tolist()))
    cfg = odit.DiTConfig(**cfgk)
    P = synth.dit_params(odit.param_shapes(cfg), seed=seed)
    x0, noise = synth.normal(f"{tag}.x0", (B, 4, H, H)), synth.normal(f"{tag}.noise", (B, 4, H, H))
    t = synth.uniform(f"{tag}.t", (B,), lo=lo, hi=1.0 - lo)
    y = synth.integers(f"{tag}.y", (B,), cfgk["n_classes"])
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    l32 = od.flow_loss(odit.dit_forward(Pr, od.flow_add_noise(x0, t, noise), t, y, cfg), x0, noise)
    l32.backward()
    assert abs(l32.item() - float(g[f"{tag}_loss_fp32"])) < 1e-6 * l32.item()  # same inputs as the fixture
    m = MMDiT(simple_dit=True, **cfgk)
    m.load_state_dict(P)
    m = m.to(DEV)
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=50)
    loss = d.compute_loss({"x": x0.to(DEV), "y": y.to(DEV), "p": 0.0}, timesteps=t, noise=noise.to(DEV))["loss"]
    loss.backward()
    torch.cuda.synchronize()
    e_loss_ref = abs(float(g[f"{tag}_loss_autocast"]) - l32.item()) / l32.item()
    assert abs(loss.item() - l32.item()) / l32.item() <= max(LOSS_BOUND, FACTOR * e_loss_ref)
    rows = []
    for k, p in m.named_parameters():
        e_hip, bound = rel(p.grad, Pr[k].grad), max(GRAD_BOUND, FACTOR * ref_err[k])
        rows.append((e_hip / bound, k, e_hip, ref_err[k]))
    rows.sort(reverse=True)
    med = sorted(r[2] for r in rows)[len(rows) // 2], sorted(r[3] for r in rows)[len(rows) // 2]
    print(f"{tag}: gradient rel-L2 median hip {med[0]:.2e} / reference under autocast {med[1]:.2e}; tightest: "
          + ", ".join(f"{k} hip {eh:.2e} ref-autocast {eb:.2e}" for _, k, eh, eb in rows[:4]))
    assert rows[0][0] <= 1.0, rows[:6]


@pytest.mark.timeout(1200)
def test_full_training_step_at_the_benched_shape_b256():
    """ONE full B=256 DiT-S/2 train step through Diffuser.compute_loss + backward -- the shape bench.py times: M = 65536 token rows,
    the 256x384 persistent NT tiles (plain and fused-SwiGLU epilogues), the 384x128 split-R weight-gradient kernel on the side
    stream with its 192-workgroup cap and f32 atomics, attention with V in place -- against the fp32 oracle on the same inputs,
    loss and EVERY parameter gradient (the benched kernels were never parity-checked at the benched shape)."""
    import os

    import numpy as np

    # round 6: the bf16-autocast leg of these seeded inputs is a committed fixture (tests/golden/make_b256_autocast.py ran the SAME
    # oracle leg once, on the GPU box's host: ~2 of this test's 3 minutes in every run of the suite before); the fp32 leg -- the
    # expected values, every gradient tensor in full -- still runs here, and its loss must reproduce the fixture's (same inputs)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dit_b256_autocast.npz"))
    fixture = {"loss": float(g["loss_bf16"]), "err": dict(zip(g["names"].tolist(), g["err"].tolist()))}
    out = three_legs(S2, seed=7, B=256, H=32, tag="pb.b256", legs=("fp32",), chunk=32)
    assert abs(out["fp32"][0] - float(g["loss_fp32"])) <= 1e-5 * abs(float(g["loss_fp32"]))
    check(out, "DiT-S/2 B=256 (benched shape)", bf16_fixture=fixture)
    # whole-arena figure as well: one number for the report
    flat = lambda g_: torch.cat([g_[k].flatten().double() for k in sorted(g_)])  # noqa: E731
    e = ((flat(out["hip"][1]) - flat(out["fp32"][1])).norm() / flat(out["fp32"][1]).norm()).item()
    e_bf = float(g["whole_grad_err"])
    print(f"B=256 whole-gradient rel-L2: hip {e:.3e}, bf16-autocast oracle {e_bf:.3e}")
    assert e <= max(GRAD_BOUND, FACTOR * e_bf)


def test_b256_step_is_bit_reproducible():
    """two runs of the benched B=256 DiT-S/2 step (same parameters, same inputs) must give
    BIT-IDENTICAL loss and gradients -- every weight gradient of the blocks comes from the grouped launch with its fixed-order fold
    (dl_gemm_tn_group), the per-sample LayerNorm / gate sums are complete inside one tile, the QK-norm scale gradients, the head and
    conditioning reductions go through partial buffers folded in a fixed order.  No f32 atomic is left on this path."""
    from diffulab_amd import Diffuser, MMDiT
    from oracle import dit as odit
    from oracle import synth

    kw = dict(S2)
    m = MMDiT(simple_dit=True, **kw)
    m.load_state_dict(synth.dit_params(odit.param_shapes(odit.DiTConfig(**kw)), seed=7))
    m = m.to("cuda")
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=50)
    B = 256
    x0, noise = synth.normal("rep.x0", (B, 4, 32, 32)).cuda(), synth.normal("rep.noise", (B, 4, 32, 32)).cuda()
    y, t = synth.integers("rep.y", (B,), 1000).cuda(), synth.uniform("rep.t", (B,), lo=0.02, hi=0.98)
    runs = []
    for _ in range(3):
        m.zero_grad()
        loss = d.compute_loss({"x": x0.clone(), "y": y, "p": 0.0}, timesteps=t, noise=noise)["loss"]
        loss.backward()
        torch.cuda.synchronize()
        runs.append((loss.detach().clone(), m._flat_grad.clone()))
    lay = m.engine.layout
    for i in (1, 2):
        assert torch.equal(runs[0][0], runs[i][0]), "loss differs between two runs"
        if not torch.equal(runs[0][1], runs[i][1]):
            bad = [n for n, (off, shape) in lay.entries.items()
                   if not torch.equal(lay.view(runs[0][1], n), lay.view(runs[i][1], n))]
            raise AssertionError(f"{len(bad)} gradient tensors differ between two runs of the same step: {bad[:24]}")
