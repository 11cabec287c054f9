"""GPU tests of the training-loop row: BaseTrainer.training_step / train on the HIP path, gradient
accumulation, fused EMA, checkpoints with the reference's file names and state_dict keys."""

import json

import pytest
import torch
from torch.utils.data import DataLoader

pytestmark = pytest.mark.gpu

from oracle import synth  # noqa: E402

DEV = "cuda"
SMALL = dict(input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4,
             patch_size=2, depth=2, n_classes=10, classifier_free=True)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def small_dit(seed=5):
    from diffulab_amd import MMDiT
    from oracle import dit as odit

    m = MMDiT(simple_dit=True, **SMALL)
    m.load_state_dict(synth.dit_params(odit.param_shapes(odit.DiTConfig(**SMALL)), seed=seed))
    return m.to(DEV)


def test_gradient_accumulation_equals_one_big_batch():
    """two micro-batches of 4 with loss/2 accumulate the same gradient as one batch of 8 (mean loss)"""
    from diffulab_amd import Diffuser

    B = 8
    x0 = synth.normal("ga.x0", (B, 4, 16, 16)).to(DEV)
    noise = synth.normal("ga.noise", (B, 4, 16, 16)).to(DEV)
    y = synth.integers("ga.y", (B,), 10).to(DEV)
    t = synth.uniform("ga.t", (B,), lo=0.05, hi=0.95)
    ma, mb = small_dit(), small_dit()
    da = Diffuser(ma, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    db = Diffuser(mb, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    da.compute_loss({"x": x0.clone(), "y": y, "p": 0.0}, timesteps=t, noise=noise)["loss"].backward()
    for lo in (0, 4):
        sl = slice(lo, lo + 4)
        loss = db.compute_loss({"x": x0[sl].clone(), "y": y[sl], "p": 0.0}, timesteps=t[sl], noise=noise[sl])["loss"]
        (loss / 2).backward()
    for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert rel(pb.grad, pa.grad) < 1e-2, n


def test_ema_fused_update_follows_ema_pytorch_rule():
    from diffulab_amd.training import EMA

    m = small_dit()
    ema = EMA(m, beta=0.99, update_after_step=2, update_every=2)
    ref = {k: v.detach().clone() for k, v in m.state_dict().items()}  # EMA starts as a copy
    assert all(torch.equal(ref[k], v) for k, v in ema.ema_model.state_dict().items())
    initted = False
    for step in range(9):
        with torch.no_grad():  # move the online weights (stands in for an optimizer step)
            for p in m.parameters():
                p.add_(0.01 * (step + 1))
        cur = {k: v.detach().clone() for k, v in m.state_dict().items()}
        # ema_pytorch.update restated
        if step % 2 == 0:
            if step <= 2:
                ref = cur
            else:
                if not initted:  # first update past update_after_step re-copies, then lerps (a no-op on equal tensors)
                    ref, initted = cur, True
                epoch = max((step + 1) - 2 - 1, 0)
                decay = 0.0 if epoch <= 0 else min(max(1 - (1 + epoch) ** (-2 / 3), 0.0), 0.99)
                ref = {k: ref[k] + (1 - decay) * (cur[k] - ref[k]) for k in ref}
        ema.update()
        got = ema.ema_model.state_dict()
        assert max(rel(got[k], ref[k]) for k in ref) < 1e-6, step
    # the EMA copy is a working denoiser with its own arena
    x = synth.normal("ema.x", (2, 4, 16, 16)).to(DEV)
    with torch.no_grad():
        out = ema.ema_model(x=x, timesteps=torch.tensor([0.3, 0.7], device=DEV), y=torch.tensor([1, 2], device=DEV))["x"]
    assert bool(torch.isfinite(out).all())


def test_base_trainer_train_loop_checkpoints_and_logs(tmp_path):
    from diffulab_amd import Diffuser, MMDiT
    from diffulab_amd.datasets import SyntheticDataset
    from diffulab_amd.training import BaseTrainer, FusedAdamW

    torch.manual_seed(0)
    m = MMDiT(simple_dit=True, **SMALL)
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=8, extra_args={"logits_normal": True})
    opt = FusedAdamW(m.parameters(), lr=2e-3, weight_decay=0.0)
    mk = lambda n, seed: DataLoader(SyntheticDataset(n, (4, 16, 16), 10, seed), batch_size=16, drop_last=True)  # noqa: E731
    tr = BaseTrainer(n_epoch=3, gradient_accumulation_step=2, save_path=tmp_path, project_name="t", use_ema=True,
                     ema_update_after_step=0, ema_update_every=1)
    tr.train(diffuser=d, optimizer=opt, train_dataloader=mk(128, 1), val_dataloader=mk(32, 2), val_steps=4,
             p_classifier_free_guidance=0.1)
    out = tmp_path / "t"
    rows = [json.loads(l) for l in (out / "metrics.jsonl").read_text().splitlines()]
    train = [r["train/loss"] for r in rows if "train/loss" in r]
    val = [r["val/loss"] for r in rows if "val/loss" in r]
    assert len(train) == 3 and len(val) == 3 and all(v == v and v < 10 for v in train + val)
    assert train[-1] < train[0]  # it learns the (fixed) synthetic distribution
    sd = torch.load(out / "denoiser.pt")
    assert set(sd) == set(m.state_dict()) and all(v.device.type == "cpu" for v in sd.values())
    ema_sd = torch.load(out / "ema.pt")
    assert set(ema_sd) == set(sd)
    osd = torch.load(out / "optimizer.pt", weights_only=False)
    assert "state" in osd and "param_groups" in osd
    imgs = torch.load(out / "val_images_epoch1.pt")
    assert imgs.shape == (16, 4, 16, 16) and float(imgs.min()) >= 0 and float(imgs.max()) <= 1
    # a fresh module restores from the checkpoint and reproduces the trained model's output
    m2 = MMDiT(simple_dit=True, **SMALL)
    m2.load_state_dict(sd)
    m2 = m2.to(DEV)
    x = synth.normal("ck.x", (2, 4, 16, 16)).to(DEV)
    kw = dict(timesteps=torch.tensor([0.3, 0.7], device=DEV), y=torch.tensor([1, 2], device=DEV))
    with torch.no_grad():
        assert rel(m2(x=x, **kw)["x"], m(x=x, **kw)["x"]) < 1e-6


def test_grad_reducer_on_rccl_single_rank():
    """the data-parallel reduction path (comm stream, events from the side-stream wgrads, async RCCL all-reduce of arena ranges,
    1/world folded into AdamW) executed for real on the GPU with a 1-rank RCCL group: gradients must equal the plain run"""
    import os

    import torch.distributed as dist

    from diffulab_amd import Diffuser
    from diffulab_amd.training.dp import GradReducer, broadcast_arena

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        B = 8
        x0 = synth.normal("rc.x0", (B, 4, 16, 16)).to(DEV)
        noise = synth.normal("rc.noise", (B, 4, 16, 16)).to(DEV)
        y = synth.integers("rc.y", (B,), 10).to(DEV)
        t = synth.uniform("rc.t", (B,), lo=0.05, hi=0.95)
        ma, mb = small_dit(), small_dit()
        broadcast_arena(mb.engine.params)
        red = GradReducer(mb._flat_grad, bucket_bytes=1 << 18)  # small buckets: several collectives per backward
        red.enabled = True  # world == 1 would switch it off
        red.comm_stream = torch.cuda.Stream()
        mb.engine.reducer = red
        for m in (ma, mb):
            d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
            d.compute_loss({"x": x0.clone(), "y": y, "p": 0.0}, timesteps=t, noise=noise)["loss"].backward()
        torch.cuda.synchronize()
        assert red.grad_scale == 1.0 and not red._works and not red._pending
        # (not bit-equal: split-R wgrads and the LayerNorm column sums meet through f32 atomics, whose order varies run to run)
        assert rel(mb._flat_grad, ma._flat_grad) < 1e-4
    finally:
        if created:
            dist.destroy_process_group()


class _RangeRecorder:
    """stands in for training.dp.GradReducer: records the gradient-arena ranges an engine's backward declares final"""

    def __init__(self):
        self.ranges, self.finished = [], False

    def ready(self, lo, hi, extra_events=()):
        assert not self.finished
        self.ranges.append((lo, hi))

    def finish(self):
        self.finished = True


@pytest.mark.parametrize("family", ["dit", "sprint", "ddt", "joint", "sprint_joint", "ddt_joint", "unet"])
def test_every_engine_hands_the_whole_gradient_arena_to_the_reducer_in_contiguous_descending_ranges(family):
    """the data-parallel reducer buckets the ranges of one backward into contiguous all-reduces (dp.py:_flush asserts it): every
    engine must announce [0, size) exactly once, high addresses first (blocks finish in reverse order), then finish()"""
    import diffulab_amd as da
    from diffulab_amd.networks.embedders import PrecomputedEmbedder

    torch.manual_seed(0)
    emb = PrecomputedEmbedder(torch.zeros(1, 64, 96), 7)
    small = dict(input_channels=4, output_channels=4, inner_dim=128, num_heads=2, mlp_ratio=4, patch_size=2)
    jk = dict(rope_axes_dim=[16, 24, 24], classifier_free=True)
    B, H = 4, 16
    inputs = {"y": torch.randint(0, 10, (B,), device=DEV)}
    ctx = {"initial_context": {"embeddings": torch.randn(B, 64, 96, device=DEV), "attn_mask": torch.ones(B, 64, dtype=torch.bool, device=DEV)}}
    if family == "dit":
        m = da.MMDiT(simple_dit=True, embedding_dim=64, depth=3, n_classes=10, **small)
    elif family == "sprint":
        m, H = da.SprintDiT(simple_dit=True, embedding_dim=64, encoder_depth=1, deep_layers_depth=2, decoder_depth=1, n_classes=10, **small), 32
    elif family == "ddt":
        m = da.DDT(simple_ddt=True, encoder_depth=2, decoder_depth=2, n_classes=10, **small)
    elif family == "joint":
        m, inputs = da.MMDiT(simple_dit=False, context_embedder=emb, embedding_dim=64, depth=2, **small, **jk), ctx
    elif family == "sprint_joint":
        m, inputs, H = da.SprintDiT(simple_dit=False, context_embedder=emb, embedding_dim=64, encoder_depth=1, deep_layers_depth=3,
                                    n_single_stream_blocks=2, decoder_depth=2, **small, **jk), ctx, 32
    elif family == "ddt_joint":
        m, inputs = da.DDT(simple_ddt=False, context_embedder=emb, encoder_depth=2, decoder_depth=2, **small, **jk), ctx
    else:
        m = da.UNetModel(image_size=[16, 16], in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1,
                         attention_resolutions=[2], channel_mult="1, 2", num_heads=2, use_scale_shift_norm=True, resblock_updown=True,
                         n_classes=10, classifier_free=True)
        small = dict(input_channels=1)
    m = m.to(DEV).train()
    rec = _RangeRecorder()
    m.engine.reducer = rec
    x = torch.randn(B, small["input_channels"], H, H, device=DEV)
    out = m(x=x, timesteps=torch.rand(B, device=DEV), **inputs)["x"]
    out.square().mean().backward()
    torch.cuda.synchronize()
    assert rec.finished and rec.ranges
    assert rec.ranges[0][1] == m._flat_grad.numel() and rec.ranges[-1][0] == 0
    for (lo, hi), (lo2, hi2) in zip(rec.ranges, rec.ranges[1:]):
        assert lo < hi and hi2 == lo, rec.ranges
