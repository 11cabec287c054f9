"""GPU tests of the training-loop row: BaseTrainer.training_step / train on the HIP path, gradient
accumulation, fused EMA, checkpoints with the reference's file names and state_dict keys."""

import json
import os

import numpy as np
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


def test_backward_accumulates_into_the_gradient_arena():
    """engine property (what textbook accumulation, DIFFULAB_TRUE_ACCUMULATION=1, relies on): two micro-batches of 4 with loss/2
    accumulate the same gradient as one batch of 8 (mean loss).  The TRAINER's default follows the reference instead: see
    test_training_step_applies_reference_accumulation below and tests/test_trainer_host.py (accelerate fixture)."""
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


def test_training_step_applies_reference_accumulation(tmp_path):
    """k = 2 on the HIP path, default mode: the gradient the optimizer sees on the synchronising micro-step is
    (1/2) * grad(second micro-batch) -- the first micro-batch's gradient was zeroed before the second backward, exactly what the
    reference does under Accelerate -- and nothing is applied on the first micro-step."""
    from diffulab_amd import Diffuser
    from diffulab_amd.training import BaseTrainer, FusedAdamW
    from diffulab_amd.training.utils import AverageMeter

    m, twin = small_dit(), small_dit()
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    dt = Diffuser(twin, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    opt = FusedAdamW(m.parameters(), lr=1e-3)
    tr = BaseTrainer(n_epoch=1, gradient_accumulation_step=2, save_path=tmp_path, project_name="acc")
    assert tr.reference_accumulation
    calls, seen = [], []
    orig_loss, orig_step = d.compute_loss, opt.step

    def spy_loss(model_inputs, timesteps=None, **kw):
        calls.append((torch.get_rng_state(), torch.cuda.get_rng_state(), {k: (v.clone() if torch.is_tensor(v) else v)
                                                                           for k, v in model_inputs.items()}, timesteps.clone()))
        return orig_loss(model_inputs=model_inputs, timesteps=timesteps, **kw)

    def spy_step(*a, **kw):
        seen.append(m._flat_grad.clone())
        return orig_step(*a, **kw)

    d.compute_loss, opt.step = spy_loss, spy_step
    w0 = m._flat.clone() if m._flat is not None else None
    torch.manual_seed(3)
    for i in range(2):
        batch = {"model_inputs": {"x": synth.normal(f"ra.x{i}", (4, 4, 16, 16)), "y": synth.integers(f"ra.y{i}", (4,), 10)}}
        tr.training_step(d, opt, batch, AverageMeter(), p_classifier_free_guidance=0.1)
        if i == 0:
            assert not seen, "no optimizer step on the non-synchronising micro-step"
            w0 = m._flat.clone()
    assert len(seen) == 1 and not torch.equal(m._flat, w0)
    cpu_rng, gpu_rng, inputs, ts = calls[1]
    torch.set_rng_state(cpu_rng)
    torch.cuda.set_rng_state(gpu_rng)
    (dt.compute_loss(model_inputs=inputs, timesteps=ts, extra_args={})["loss"] / 2).backward()
    assert rel(seen[0], twin._flat_grad) < 2e-3  # (not bit-equal: f32 atomics in the weight-gradient reductions)


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
    assert m.precision == "fp32"  # the trainer's default precision_type is the reference's "no": the whole loop ran the fp32 regime
    m2 = MMDiT(simple_dit=True, **SMALL)
    m2.load_state_dict(sd)
    m2 = m2.set_precision(m.precision).to(DEV)
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


def test_grad_reducer_through_the_comm_c_abi_single_rank():
    """the same backward with the exchange going through libdiffulab_comm.so (dl_comm_init / dl_reduce_scatter_allgather_async /
    dl_comm_after_event / dl_comm_wait on the library's own RCCL communicator and stream; 1 rank: the sum leaves the data as it is):
    gradients equal the plain run and the compute stream really waited for the collectives"""
    from diffulab_amd import Diffuser
    from diffulab_amd._comm import Communicator
    from diffulab_amd.training.dp import GradReducer

    B = 8
    x0 = synth.normal("ca.x0", (B, 4, 16, 16)).to(DEV)
    noise = synth.normal("ca.noise", (B, 4, 16, 16)).to(DEV)
    y = synth.integers("ca.y", (B,), 10).to(DEV)
    t = synth.uniform("ca.t", (B,), lo=0.05, hi=0.95)
    ma, mb = small_dit(), small_dit()
    comm = Communicator(rank=0, world=1, device=0)
    try:
        mb.engine  # flatten
        comm.broadcast_async(mb._flat.data_ptr(), mb._flat.numel(), 0, torch.cuda.current_stream().cuda_stream)
        comm.wait(torch.cuda.current_stream().cuda_stream)
        red = GradReducer(mb._flat_grad, bucket_bytes=1 << 18, backend="abi", comm=comm)
        red.enabled = True  # world == 1 would switch it off
        mb.engine.reducer = red
        for m in (ma, mb):
            d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
            d.compute_loss({"x": x0.clone(), "y": y, "p": 0.0}, timesteps=t, noise=noise)["loss"].backward()
        torch.cuda.synchronize()
        assert not red._pending and rel(mb._flat_grad, ma._flat_grad) < 1e-4 and torch.equal(mb._flat, ma._flat)
    finally:
        mb.engine.reducer = None
        comm.close()


class _RangeRecorder:
    """stands in for training.dp.GradReducer: records the gradient-arena ranges an engine's backward declares final"""

    def __init__(self):
        self.ranges, self.finished = [], False

    def ready(self, lo, hi, extra_events=(), flush=False):
        assert not self.finished
        if hi > lo:
            self.ranges.append((lo, hi))

    def finish(self):
        self.finished = True

    def rebind(self, flat_grad):
        self.rebound = flat_grad


def test_average_meter_reads_device_losses_back_lazily_and_identically():
    """the trainer hands the meter `loss.detach()` (no host synchronisation between forward and backward); the values are read back
    when the meter is looked at, in the order of the calls, as the floats `.item()` would have produced"""
    from diffulab_amd.training.utils import AverageMeter

    g = torch.Generator().manual_seed(3)
    vals = torch.rand(300, generator=g)
    lazy, eager = AverageMeter(), AverageMeter()
    for i, v in enumerate(vals):
        key = "train/loss" if i % 3 else "train/repa"
        lazy.update(v.to(DEV), key, n=1 + i % 2)
        eager.update(v.item(), key, n=1 + i % 2)
        if i == 10:
            assert lazy._pending and lazy.keys == eager.keys  # nothing read yet, the keys exist
    assert len(lazy._pending) < AverageMeter.MAX_PENDING  # (flushed once on the way: bounded)
    assert lazy.avg == eager.avg and lazy.sum == eager.sum and lazy.count == eager.count and not lazy._pending
    lazy.update(torch.tensor(2.0, device=DEV), "train/loss")
    lazy.reset()
    assert lazy.count["train/loss"] == 0 and not lazy._pending


def test_reducer_survives_a_precision_switch_after_prepare():
    """set_precision() after the trainer's prepare() drops the engine the reducer was attached to; the engine that
    replaces it (and a plain re-flattening) must carry the reducer over, re-pointed at the NEW gradient arena -- otherwise a
    data-parallel run silently stops exchanging gradients"""
    import diffulab_amd as da

    m = da.MMDiT(simple_dit=True, input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4,
                 patch_size=2, depth=2, n_classes=10).to(DEV).train()
    rec = _RangeRecorder()
    m.engine.reducer = rec
    old_engine, old_grad = m.engine, m._flat_grad
    m.set_precision("fp32")
    eng = m.engine
    assert eng is not old_engine and eng.reducer is rec and rec.rebound is m._flat_grad and m._flat_grad is not old_grad
    m.flatten_parameters()  # same engine, new arena
    assert m.engine.reducer is rec and rec.rebound is m._flat_grad
    x = torch.randn(4, 4, 16, 16, device=DEV)
    m(x=x, timesteps=torch.rand(4, device=DEV), y=torch.randint(0, 10, (4,), device=DEV))["x"].square().mean().backward()
    torch.cuda.synchronize()
    assert rec.finished and sorted(rec.ranges)[0][0] == 0 and max(hi for _, hi in rec.ranges) == m._flat_grad.numel()
    from diffulab_amd.training.dp import GradReducer

    red = GradReducer(torch.zeros(8, device=DEV))
    new = torch.zeros(8, device=DEV)
    red.rebind(new)
    assert red.flat is new
    with pytest.raises(RuntimeError):
        red.rebind(torch.zeros(9, device=DEV))


@pytest.mark.parametrize("family", ["dit", "sprint", "ddt", "joint", "sprint_joint", "ddt_joint", "unet", "dit:fp32", "sprint:fp32", "ddt:fp32",
                                    "unet:fp32"])
def test_every_engine_hands_the_whole_gradient_arena_to_the_reducer_exactly_once(family):
    """the data-parallel reducer sums the ranges an engine's backward declares final (one all-reduce per contiguous run of a
    flush, dp.py:_flush): every engine must announce every element of [0, size) exactly once -- block ranges high addresses first
    (blocks finish in reverse order; the DiT engine adds each block's slice of the stacked adaLN matrix) -- then finish()"""
    import diffulab_amd as da
    from diffulab_amd.networks.embedders import PrecomputedEmbedder

    family, _, precision = family.partition(":")  # (the fp32-class engines announce the whole arena once, at the end)
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
    if precision:
        m.set_precision(precision)
    m = m.to(DEV).train()
    assert getattr(m.engine, "precision", "bf16") == (precision or "bf16")
    rec = _RangeRecorder()
    m.engine.reducer = rec
    x = torch.randn(B, small["input_channels"], H, H, device=DEV)
    out = m(x=x, timesteps=torch.rand(B, device=DEV), **inputs)["x"]
    out.square().mean().backward()
    torch.cuda.synchronize()
    assert rec.finished and rec.ranges
    assert rec.ranges[0][1] == m._flat_grad.numel(), rec.ranges[:2]  # the last block's parameters come first
    covered = sorted(rec.ranges)
    assert covered[0][0] == 0 and covered[-1][1] == m._flat_grad.numel()
    for (lo, hi), (lo2, hi2) in zip(covered, covered[1:]):
        assert lo < hi and lo2 == hi, covered  # no gap, no overlap


def test_optimizer_checkpoint_resumes_step_and_moments(tmp_path):
    """save -> "new process" (fresh module, fresh optimizer, new arena address) -> resume: the fused AdamW picks its step count
    and moments up from optimizer.pt, also when load_state_dict runs before the model is on the GPU (the state used
    to be keyed on the arena's address and silently restarted from zero)."""
    from diffulab_amd import Diffuser, MMDiT
    from diffulab_amd.training import FusedAdamW

    x0 = synth.normal("rs.x0", (4, 4, 16, 16)).to(DEV)
    noise = synth.normal("rs.noise", (4, 4, 16, 16)).to(DEV)
    y = synth.integers("rs.y", (4,), 10).to(DEV)
    t = synth.uniform("rs.t", (4,), lo=0.05, hi=0.95)

    def one_step(m, opt):
        opt.zero_grad()
        d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
        d.compute_loss({"x": x0.clone(), "y": y, "p": 0.0}, timesteps=t, noise=noise)["loss"].backward()
        opt.step()

    m = small_dit()
    opt = FusedAdamW(m.parameters(), lr=1e-3)
    for _ in range(3):
        one_step(m, opt)
    torch.save(opt.state_dict(), tmp_path / "optimizer.pt")
    torch.save({k: v.cpu() for k, v in m.state_dict().items()}, tmp_path / "denoiser.pt")
    one_step(m, opt)  # the continuation the resumed run must reproduce
    keep = torch.zeros(1 << 20, device=DEV)  # shifts the allocator so the new arena cannot land on the old address by luck

    m2 = MMDiT(simple_dit=True, **SMALL)
    m2.load_state_dict(torch.load(tmp_path / "denoiser.pt"))
    opt2 = FusedAdamW(m2.parameters(), lr=1e-3)
    opt2.load_state_dict(torch.load(tmp_path / "optimizer.pt", weights_only=False))  # BEFORE the move to the GPU
    m2 = m2.to(DEV)
    one_step(m2, opt2)
    st = [v for v in opt2.state.values() if "m" in v and v["m"].numel() == m2._flat.numel()]
    assert len(st) == 1 and st[0]["step"] == 4 and st[0]["m"].is_cuda
    assert not any(isinstance(k, str) for k in opt2.state), "no orphaned address-keyed entry"
    assert rel(m2._flat, m._flat) < 1e-5
    del keep


def test_reference_adamw_checkpoint_resumes_in_the_fused_optimizer(tmp_path):
    """the reference saves torch.optim.AdamW's state (per-parameter exp_avg / exp_avg_sq / step, base_trainer.py:246-251).
    FusedAdamW packs such an optimizer.pt into its arena moments and continues the trajectory: 2 stock-AdamW steps + 1 fused step
    == 3 stock-AdamW steps (the two optimizers are the same update rule; the stock one runs on the same HIP gradients)."""
    from diffulab_amd import Diffuser
    from diffulab_amd.training import FusedAdamW

    x0 = synth.normal("ra.x0", (4, 4, 16, 16)).to(DEV)
    noise = synth.normal("ra.noise", (4, 4, 16, 16)).to(DEV)
    y = synth.integers("ra.y", (4,), 10).to(DEV)
    t = synth.uniform("ra.t", (4,), lo=0.05, hi=0.95)

    def one_step(m, opt):
        opt.zero_grad()
        d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
        d.compute_loss({"x": x0.clone(), "y": y, "p": 0.0}, timesteps=t, noise=noise)["loss"].backward()
        opt.step()

    kw = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    m = small_dit()
    stock = torch.optim.AdamW(m.parameters(), foreach=False, **kw)
    one_step(m, stock)
    one_step(m, stock)
    torch.save(stock.state_dict(), tmp_path / "optimizer.pt")
    sd = {k: v.cpu().clone() for k, v in m.state_dict().items()}
    one_step(m, stock)  # the continuation

    m2 = small_dit()
    m2.load_state_dict(sd)
    opt2 = FusedAdamW(m2.parameters(), **kw)
    opt2.load_state_dict(torch.load(tmp_path / "optimizer.pt", weights_only=False))
    one_step(m2, opt2)
    st = [v for v in opt2.state.values() if "m" in v and v["m"].numel() == m2._flat.numel()]
    assert len(st) == 1 and st[0]["step"] == 3 and "exp_avg" not in st[0]
    assert not any("exp_avg" in v for v in opt2.state.values()), "per-parameter entries of the arena parameters are consumed"
    assert rel(m2._flat, m._flat) < 2e-5


def test_reference_adamw_checkpoint_resumes_with_parameters_outside_the_arena(tmp_path):
    """examples/train_repa.py puts the REPA projector (and resampler) tensors into the SAME param group as the denoiser;
    they live outside the arena.  A stock torch.optim.AdamW optimizer.pt must resume for them as well (exp_avg / exp_avg_sq ->
    m / v): 2 stock steps + 1 fused step == 3 stock steps, for the arena AND for the projector."""
    from diffulab_amd import Diffuser
    from diffulab_amd.training import FusedAdamW

    x0 = synth.normal("rb.x0", (4, 4, 16, 16)).to(DEV)
    noise = synth.normal("rb.noise", (4, 4, 16, 16)).to(DEV)
    y = synth.integers("rb.y", (4,), 10).to(DEV)
    t = synth.uniform("rb.t", (4,), lo=0.05, hi=0.95)

    def make():
        torch.manual_seed(3)
        return small_dit(), torch.nn.Linear(16, 8).to(DEV)

    def one_step(m, proj, opt):
        opt.zero_grad()
        d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
        loss = d.compute_loss({"x": x0.clone(), "y": y, "p": 0.0}, timesteps=t, noise=noise)["loss"]
        (loss + proj(x0.reshape(4, -1, 16)).square().mean()).backward()
        opt.step()

    kw = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    m, proj = make()
    stock = torch.optim.AdamW(list(m.parameters()) + list(proj.parameters()), foreach=False, **kw)
    one_step(m, proj, stock)
    one_step(m, proj, stock)
    torch.save(stock.state_dict(), tmp_path / "optimizer.pt")
    sd = {k: v.cpu().clone() for k, v in m.state_dict().items()}
    sdp = {k: v.cpu().clone() for k, v in proj.state_dict().items()}
    one_step(m, proj, stock)

    m2, proj2 = make()
    m2.load_state_dict(sd)
    proj2.load_state_dict(sdp)
    opt2 = FusedAdamW(list(m2.parameters()) + list(proj2.parameters()), **kw)
    opt2.load_state_dict(torch.load(tmp_path / "optimizer.pt", weights_only=False))
    one_step(m2, proj2, opt2)
    assert not any("exp_avg" in v for v in opt2.state.values())
    assert all(opt2.state[q]["step"] == 3 for q in proj2.parameters())
    assert rel(m2._flat, m._flat) < 2e-5
    for a, b in zip(proj2.parameters(), proj.parameters()):
        assert rel(a, b) < 2e-6


def test_weights_written_through_parameters_reach_the_inference_shadows():
    """in-place writes through a parameter (load_state_dict on a flattened model, a stock optimizer) do not bump the
    arena's version counter; eval / no_grad / hipGraph-replay forwards must still see the new weights"""
    from oracle import dit as odit

    m = small_dit(seed=5).eval()
    x = synth.normal("sh.x", (2, 4, 16, 16)).to(DEV)
    kw = dict(timesteps=torch.tensor([0.3, 0.7], device=DEV), y=torch.tensor([1, 2], device=DEV))
    with torch.no_grad():
        a1 = m(x=x, **kw)["x"].clone()
        a2 = m(x=x, **kw)["x"].clone()   # second call: the captured graph replays
    assert torch.equal(a1, a2)
    m.load_state_dict(synth.dit_params(odit.param_shapes(odit.DiTConfig(**SMALL)), seed=6))  # writes through p.copy_
    fresh = small_dit(seed=6).eval()
    with torch.no_grad():
        got, want = m(x=x, **kw)["x"], fresh(x=x, **kw)["x"]
    assert rel(got, want) < 1e-6 and rel(got, a1) > 1e-2
    sgd = torch.optim.SGD(m.parameters(), lr=0.5)  # a stock optimizer writes p.data in place
    for p in m.parameters():
        p.grad = torch.ones_like(p) * 0.01
    sgd.step()
    fresh2 = small_dit(seed=6)
    with torch.no_grad():
        for p in fresh2.parameters():
            p.sub_(0.5 * 0.01)
        got, want = m(x=x, **kw)["x"], fresh2.eval()(x=x, **kw)["x"]
    assert rel(got, want) < 1e-6


def test_graph_replay_survives_workspace_eviction():
    """the engine keeps 8 workspaces and the module 8 graphs, with unlinked bounds; a replay of an old shape after its
    workspace left the cache must still be correct (graph entries now keep their workspace alive)"""
    m = small_dit().eval()
    kw = lambda b: dict(timesteps=torch.full((b,), 0.4, device=DEV), y=torch.zeros(b, dtype=torch.long, device=DEV))  # noqa: E731
    xs = {b: synth.normal(f"ev.x{b}", (b, 4, 16, 16)).to(DEV) for b in range(1, 10)}
    want = {}
    with torch.no_grad():
        m(x=xs[1], **kw(1))
        want[1] = m(x=xs[1], **kw(1))["x"].clone()  # replayed once: graph for B=1 is live
    # a training shape, then 8 further inference shapes: the B=1 workspace is evicted from the engine's cache
    from diffulab_amd import Diffuser
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    m.train()
    d.compute_loss({"x": xs[4].clone(), "y": kw(4)["y"], "p": 0.0}, timesteps=torch.full((4,), 0.5))["loss"].backward()
    m.eval()
    with torch.no_grad():
        for b in range(2, 9):
            m(x=xs[b], **kw(b))
        junk = [torch.randn(1 << 18, device=DEV) for _ in range(64)]  # reuse whatever memory was returned to the allocator
        got = m(x=xs[1], **kw(1))["x"]
    del junk
    assert rel(got, want[1]) < 1e-6


def test_backward_through_a_stale_forward_raises():
    from diffulab_amd import Diffuser

    m = small_dit()
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    mk = lambda tag: d.compute_loss({"x": synth.normal(tag, (2, 4, 16, 16)).to(DEV), "y": torch.zeros(2, dtype=torch.long, device=DEV),  # noqa: E731
                                     "p": 0.0}, timesteps=torch.tensor([0.3, 0.6]))["loss"]
    l1, l2 = mk("st.a"), mk("st.b")
    l2.backward()
    with pytest.raises(RuntimeError, match="no longer the module's latest"):
        l1.backward()


def _dp_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    import os

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # both ranks share the one GPU of the test box
    from diffulab_amd import Diffuser
    from diffulab_amd.training.dp import GradReducer, broadcast_arena

    m = small_dit(seed=5 + rank)  # different start per rank: the broadcast must equalise them
    eng = m.engine
    broadcast_arena(m._flat)
    red = GradReducer(m._flat_grad, bucket_bytes=1 << 18)
    eng.reducer = red
    B = 8
    n = B // world
    x0 = synth.normal("dp2.x0", (B, 4, 16, 16))[rank * n:(rank + 1) * n].to(DEV)
    noise = synth.normal("dp2.noise", (B, 4, 16, 16))[rank * n:(rank + 1) * n].to(DEV)
    y = synth.integers("dp2.y", (B,), 10)[rank * n:(rank + 1) * n].to(DEV)
    t = synth.uniform("dp2.t", (B,), lo=0.05, hi=0.95)[rank * n:(rank + 1) * n]
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    d.compute_loss({"x": x0.clone(), "y": y, "p": 0.0}, timesteps=t, noise=noise)["loss"].backward()
    torch.cuda.synchronize()
    torch.save({"grad": (m._flat_grad * red.grad_scale).cpu(), "param": m._flat.cpu()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_gradients_equal_the_concatenated_batch(tmp_path):
    """SURVEY §4 / VERDICT r1: the MODEL on 2 ranks.  Two processes (both on the test box's single GPU, gloo for the exchange) run
    the engine's backward on their half of a batch of 8 with the GradReducer attached; after the bucketed all-reduce and the
    1/world scale both ranks hold the gradient a single process computes on the whole batch, and the rank-0 broadcast made the
    parameters equal although the ranks were initialised differently."""
    import torch.multiprocessing as mp

    from diffulab_amd import Diffuser

    port = 29000 + (hash(str(tmp_path)) % 2000)
    mp.start_processes(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["param"], r1["param"]) and torch.equal(r0["grad"], r1["grad"])
    m = small_dit(seed=5)
    assert torch.equal(m.engine.params.cpu(), r0["param"])
    B = 8
    x0, noise = synth.normal("dp2.x0", (B, 4, 16, 16)).to(DEV), synth.normal("dp2.noise", (B, 4, 16, 16)).to(DEV)
    y, t = synth.integers("dp2.y", (B,), 10).to(DEV), synth.uniform("dp2.t", (B,), lo=0.05, hi=0.95)
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
    d.compute_loss({"x": x0.clone(), "y": y, "p": 0.0}, timesteps=t, noise=noise)["loss"].backward()
    assert rel(r0["grad"], m._flat_grad) < 2e-3  # (bf16 kernels on batch 4 vs 8: different tiles / atomics order, same math)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("launcher", ["torchrun", "self"])
def test_bench_data_parallel_branch_runs_on_two_ranks(tmp_path, launcher):
    """launcher "self" = the plain `python bench.py --gpus 2` form: bench.py spawns its own ranks.
    VERDICT r2 #8: bench.py's N > 1 branch (reducer attached to the engine, weak-scaling accounting, the `dp` object with the
    exposed all-reduce time) had never executed anywhere.  Its dry mode -- the driver's own launch line with `--dp-backend gloo`, two
    ranks sharing this box's GPU -- runs the same code path end to end and prints the same JSON line."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29000 + (hash(str(tmp_path)) % 2000)
    pre = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] if launcher == "torchrun" else [sys.executable]
    cmd = pre + [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "15", "--batch", "16",
                 "--dp-backend", "gloo", "--no-roofline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=800, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 3 and line["warmup"] == 15
    assert line["config"]["global_batch"] == 32 and line["config"]["per_gpu_batch"] == 16 and line["config"]["parallelism"] == "dp2"
    assert abs(line["value"] - 32 * 1e3 / line["ms_per_step"]) / line["value"] < 1e-2  # whole-job images/s over the max-rank time
    dp = line["dp"]
    assert dp["rccl_ranks"] == 2 and dp["backend"] == "gloo" and dp["grad_bytes_per_step"] > 100e6
    assert isinstance(dp["exposed_allreduce_ms_per_step"], float) and dp["exposed_allreduce_ms_per_step"] >= 0.0
    # the schedule of the exchange was decided by the warm-up measurement (training/dp.py: steps 2-7 overlapped, 8-13 after the backward)
    ex = dp["exchange"]
    assert ex["mode"] in ("overlapped", "after_backward") and ex["overlapped_ms_per_step"] > 0 and ex["after_backward_ms_per_step"] > 0
    assert np.isfinite(line["config"]["final_loss"]) and "DRY MODE" in line["data"]


def test_graphed_training_step_follows_the_eager_trajectory():
    """scripts/lab/graph_step.py (lab code): after three eager steps the whole step (zero_grad -> noise + loss -> forward -> backward with the
    side-stream wgrads -> FusedAdamW reading its scalars from the device) is captured once and replayed.  Same seeds, same data:
    the parameters after 8 steps agree with 8 eager steps (device RNG draws go through torch's graph-safe generator) and the step
    count / learning-rate changes reach the captured update."""
    from diffulab_amd import Diffuser
    from diffulab_amd.training import FusedAdamW
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "lab"))
    from graph_step import GraphedTrainStep  # scripts/lab (lab code, not part of the package)

    B = 8
    x0 = synth.normal("gs.x0", (B, 4, 16, 16)).to(DEV)
    y = synth.integers("gs.y", (B,), 10).to(DEV)
    ts = [synth.uniform(f"gs.t{i}", (B,), lo=0.05, hi=0.95).to(DEV) for i in range(8)]

    def run(graphed: bool):
        m = small_dit()
        d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=4)
        opt = FusedAdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
        gs = GraphedTrainStep(d, opt, warmup=3) if graphed else None
        torch.manual_seed(123)
        losses = []
        for i in range(8):
            if i == 5:
                opt.param_groups[0]["lr"] = 3e-4  # a scheduler step
            inputs = {"x": x0, "y": y, "p": 0.0}
            if graphed:
                out = gs(inputs, ts[i], {})
            else:
                opt.zero_grad()
                out = d.compute_loss(model_inputs=dict(inputs), timesteps=ts[i], extra_args={})
                sum(out.values()).backward()
                opt.step()
            losses.append(out["loss"].item())
        captured = graphed and any(v not in (None, False) for v in gs._graphs.values())
        st = [v for v in opt.state.values() if "m" in v and v["m"].numel() == m._flat.numel()][0]
        res = (m._flat.clone(), losses, int(st["step"]), captured)
        if graphed:  # the captured graphs die before the buffers their nodes point to, on an idle device
            torch.cuda.synchronize()
            gs._graphs.clear()
            del gs
            import gc

            gc.collect()
            torch.cuda.synchronize()
        return res

    pe, le, se, _ = run(False)
    pg, lg, sg, captured = run(True)
    assert captured, "the step was not captured"
    assert se == sg == 8
    print("eager losses ", [f"{v:.5f}" for v in le])
    print("graph losses ", [f"{v:.5f}" for v in lg])
    assert max(abs(a - b) / abs(a) for a, b in zip(le, lg)) < 2e-3
    assert rel(pg, pe) < 2e-3
