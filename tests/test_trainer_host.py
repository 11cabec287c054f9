"""Host-side logic of the trainer mirror (no GPU): config composition, ``_target_`` resolution, EMA decay schedule of
ema_pytorch (restated), batch sharding with ``split_batches=True`` semantics, accumulation boundaries."""

import copy
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_composition_and_instantiate():
    from diffulab_amd.config import instantiate, load_config

    c = load_config(os.path.join(ROOT, "configs"), "train_mnist_ddpm", ["trainer.n_epoch=3", "model.model_channels=64",
                                                                        "dataset=mnist_synthetic"])
    assert c.trainer.n_epoch == 3 and c.trainer.project_name == "mnist_ddpm" and c.trainer.use_ema is True
    assert c.dataloader.batch_size == 128 and c.diffuser.sampling_method == "ddpm"
    assert c.optimizer.lr == 1e-4 and isinstance(c.optimizer.lr, float)
    m = instantiate(c.model)
    assert type(m).__name__ == "UNetModel" and m.model_channels == 64 and m.channel_mult == [1, 2, 4, 8]
    opt = instantiate(c.optimizer, params=m.parameters())
    assert type(opt).__name__ == "FusedAdamW" and opt.defaults["weight_decay"] == 0.01
    ds = instantiate(c.dataset.val)
    item = ds[3]
    assert item["model_inputs"]["x"].shape == (1, 32, 32) and item["model_inputs"]["y"].dtype == torch.int64
    c2 = load_config(os.path.join(ROOT, "configs"), "train_cifar10_flow_matching")
    assert c2.diffuser.n_steps == 100 and c2.diffuser.extra_args.logits_normal is True and c2.model.inner_dim == 512
    # Hydra-style group choice: another file of a defaults group
    c3 = load_config(os.path.join(ROOT, "configs"), "train_cifar10_flow_matching", ["optimizer=sgd", "model=sprint", "trainer.n_epoch=2"])
    assert c3.optimizer._target_ == "torch.optim.SGD" and c3.optimizer.nesterov is True and c3.trainer.n_epoch == 2
    assert c3.model._target_.endswith("SprintDiT") and c3.model.drop_rate == 0.75
    # every shipped top-level config composes and names an importable denoiser
    for name in sorted(f[:-5] for f in os.listdir(os.path.join(ROOT, "configs")) if f.endswith(".yaml")):
        cfg = load_config(os.path.join(ROOT, "configs"), name)
        assert cfg.model._target_.split(".")[-1] in ("MMDiT", "UNetModel", "SprintDiT", "DDT"), name
        assert {"trainer", "diffuser", "dataset", "optimizer"} <= set(cfg), name


def test_ema_decay_schedule_matches_ema_pytorch_formula():
    from diffulab_amd.training.ema import EMA

    e = EMA.__new__(EMA)
    torch.nn.Module.__init__(e)
    e.beta, e.update_after_step, e.update_every = 0.999, 0, 10
    e.inv_gamma, e.power, e.min_value = 1.0, 2 / 3, 0.0
    got = []
    for step in (1, 2, 11, 101, 100001):
        e.step = step
        got.append(e.get_current_decay())
    # epoch = step - 1: 0 -> 0 ; 1 -> 1 - 2^(-2/3) ; 10 -> 1 - 11^(-2/3) ; 100 -> 1 - 101^(-2/3) ; large -> clamped to beta
    want = [0.0, 1 - 2 ** (-2 / 3), 1 - 11 ** (-2 / 3), 1 - 101 ** (-2 / 3), 0.999]
    assert all(abs(a - b) < 1e-12 for a, b in zip(got, want)), (got, want)


def test_shard_batch_and_accumulation_boundaries(tmp_path):
    from diffulab_amd.training import BaseTrainer

    t = BaseTrainer(n_epoch=1, gradient_accumulation_step=3, save_path=tmp_path, project_name="p", use_ema=True,
                    ema_update_after_step=2, ema_update_every=10)
    assert t.ema_update_after_step == 6 and t.ema_update_every == 30  # scaled by the accumulation factor (common.py:97-98)
    flags = []
    for _ in range(6):
        flags.append(t.sync_gradients)
        t.end_micro_step()
    assert flags == [False, False, True, False, False, True]
    flags = []
    for _ in t.iterate(range(5)):  # the last batch of a dataloader pass always synchronises and restarts the window
        flags.append(t.sync_gradients)
        t.end_micro_step()
    assert flags == [False, False, True, False, True] and t._accum_step == 0
    t.world, t.rank = 4, 2
    b = {"model_inputs": {"x": torch.arange(8)[:, None], "y": torch.arange(8), "p": 0.1}, "extra": {"captions": list("abcdefgh")}}
    s = t.shard_batch(b)
    assert s["model_inputs"]["x"].flatten().tolist() == [4, 5] and s["model_inputs"]["y"].tolist() == [4, 5]
    assert s["model_inputs"]["p"] == 0.1 and s["extra"]["captions"] == ["e", "f"]
    assert (tmp_path / "p").is_dir()
    import pytest
    with pytest.raises(ValueError):  # a batch the ranks cannot share evenly never reaches shard_batch behind iterate() (below)
        t.shard_batch({"model_inputs": {"x": torch.arange(6)}})


@pytest.mark.parametrize("n,batch,world", [(23, 8, 4), (15, 8, 8), (1281167 % (128 * 7) + 128, 128, 8), (5, 8, 2), (16, 8, 4)])
def test_last_partial_batch_is_completed_like_accelerate_even_batches(tmp_path, n, batch, world):
    """the reference's loader keeps the last partial batch (drop_last=False) and Accelerate (split_batches=True,
    even_batches=True) completes it with samples from the first batch instead of failing mid-epoch.  Pinned against accelerate's own
    BatchSamplerShard: every rank must see exactly the sample indices accelerate would hand it."""
    accelerate = pytest.importorskip("accelerate")
    from accelerate.data_loader import BatchSamplerShard
    from torch.utils.data import BatchSampler, SequentialSampler

    from diffulab_amd.training import BaseTrainer

    class Loader(list):  # (what the trainer sees of a torch DataLoader: iteration and .batch_size)
        batch_size = batch

    data = torch.arange(n)
    loader = Loader(({"model_inputs": {"x": data[i : i + batch, None].float(), "y": data[i : i + batch]}, "extra": {"names": [str(int(j)) for j in data[i : i + batch]]}}
              for i in range(0, n, batch)))
    for rank in range(world):
        want = list(BatchSamplerShard(BatchSampler(SequentialSampler(range(n)), batch, drop_last=False), num_processes=world,
                                      process_index=rank, split_batches=True, even_batches=True))
        t = BaseTrainer(n_epoch=1, save_path=tmp_path, project_name="p")
        t.world, t.rank = world, rank
        got, flags = [], []
        for b in t.iterate(loader):
            sh = t.shard_batch(b)
            assert sh["model_inputs"]["x"].flatten().tolist() == sh["model_inputs"]["y"].tolist() == [int(v) for v in sh["extra"]["names"]]
            got.append(sh["model_inputs"]["y"].tolist())
            flags.append(t.sync_gradients)
            t.end_micro_step()
        assert got == want, (rank, got, want)
        assert flags[-1] and len(got) == len(loader)
    t = BaseTrainer(n_epoch=1, save_path=tmp_path, project_name="p")
    t.world = 3  # a batch size the processes cannot split evenly is refused on the FIRST batch, as accelerate does at prepare time
    with pytest.raises(ValueError, match="round multiple"):
        next(iter(t.iterate(loader)))


class _FakeDiffuser:
    """just enough of ``Diffuser`` for ``BaseTrainer.train`` on CPU: a torch module as denoiser and an MSE ``compute_loss``"""

    model_type = "rectified_flow"
    extra_losses: list = []

    def __init__(self, model, on_loss):
        self.denoiser, self.on_loss = model, on_loss
        self.denoiser.classifier_free = False

    def train(self):
        self.denoiser.train()

    def eval(self):
        self.denoiser.eval()

    def draw_timesteps(self, n):
        return torch.zeros(n)

    def compute_loss(self, model_inputs, timesteps, extra_args):
        self.on_loss()
        return {"loss": torch.nn.functional.mse_loss(self.denoiser(model_inputs["x"]), extra_args["target"])}


def _replay_accum_fixture(tmp_path, true_accumulation: bool):
    import numpy as np

    from diffulab_amd.training import BaseTrainer

    g = np.load(os.path.join(ROOT, "tests", "golden", "accum_k2.npz"))
    model = torch.nn.Linear(g["xs"].shape[2], g["ys"].shape[2])
    with torch.no_grad():
        model.weight.copy_(torch.from_numpy(g["w0"]))
        model.bias.copy_(torch.from_numpy(g["b0"]))
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0 / (1.0 + 0.5 * s))
    seen_w, seen_lr = [], []
    # compute_loss of micro-step i observes the parameters left by micro-step i-1
    d = _FakeDiffuser(model, lambda: (seen_w.append(model.weight.detach().clone().numpy()), seen_lr.append(opt.param_groups[0]["lr"])))
    loader = [{"model_inputs": {"x": torch.from_numpy(g["xs"][i])}, "extra": {"target": torch.from_numpy(g["ys"][i])}}
              for i in range(g["xs"].shape[0])]
    t = BaseTrainer(n_epoch=int(g["n_epoch"]), gradient_accumulation_step=int(g["k"]), save_path=tmp_path, project_name="acc")
    t.reference_accumulation = not true_accumulation
    t.device = torch.device("cpu")
    t.train(d, opt, loader, scheduler=sched, per_batch_scheduler=True)
    seen_w.append(model.weight.detach().clone().numpy())
    seen_lr.append(opt.param_groups[0]["lr"])
    return g, np.stack(seen_w[1:]), np.array(seen_lr[1:])


def test_gradient_accumulation_reproduces_accelerate_fixture(tmp_path):
    """the default reproduces what the reference does under Accelerate with k = 2 (zero_grad is gated on the
    CURRENT micro-step's sync flag, so the update uses (1/k) * grad(last micro-batch); the last batch of a pass always syncs):
    parameters and learning rate after every one of the 10 micro-steps against tests/golden/accum_k2.npz (made with accelerate)"""
    import numpy as np

    g, w, lr = _replay_accum_fixture(tmp_path, true_accumulation=False)
    assert w.shape == g["traj_w"].shape
    np.testing.assert_allclose(w, g["traj_w"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(lr, g["traj_lr"], rtol=1e-12)


def test_true_accumulation_switch_differs_from_the_reference(tmp_path):
    """the documented switch gives textbook accumulation, which is NOT the reference trajectory"""
    import numpy as np

    g, w, lr = _replay_accum_fixture(tmp_path, true_accumulation=True)
    np.testing.assert_allclose(lr, g["traj_lr"], rtol=1e-12)  # same synchronisation points
    assert np.abs(w - g["traj_w"]).max() > 1e-4
    # first window: mean gradient of micro-batches 0 and 1
    import torch as th
    m = th.nn.Linear(g["xs"].shape[2], g["ys"].shape[2])
    with th.no_grad():
        m.weight.copy_(th.from_numpy(g["w0"]))
        m.bias.copy_(th.from_numpy(g["b0"]))
    o = th.optim.AdamW(m.parameters(), lr=1e-2, weight_decay=0.01)
    (0.5 * (th.nn.functional.mse_loss(m(th.from_numpy(g["xs"][0])), th.from_numpy(g["ys"][0]))
            + th.nn.functional.mse_loss(m(th.from_numpy(g["xs"][1])), th.from_numpy(g["ys"][1])))).backward()
    o.step()
    np.testing.assert_allclose(w[1], m.weight.detach().numpy(), atol=2e-7)


def test_precision_types_follow_the_reference_default(tmp_path):
    """``precision_type="no"`` (the reference's fp32 default, trainers/common.py:76,105; configs/trainer/default.yaml:4) is a
    regime of its own since round 4 (engine_f32.py): the trainer accepts it, defaults to it like the reference, and asks the denoiser
    for its fp32 launch sequence at prepare(); a denoiser that has only the bf16 regime refuses there and names the override.
    fp16 / fp8 are still refused at construction (they would silently be something else)."""
    import yaml

    from diffulab_amd import MMDiT, UNetModel
    from diffulab_amd.training import BaseTrainer

    for bad in ("fp16", "fp8"):
        with pytest.raises(NotImplementedError, match="bf16"):
            BaseTrainer(n_epoch=1, precision_type=bad, save_path=tmp_path, project_name="p")
    assert BaseTrainer(n_epoch=1, precision_type="bf16", save_path=tmp_path, project_name="p").precision_type == "bf16"
    assert BaseTrainer(n_epoch=1, save_path=tmp_path, project_name="p").precision_type == "no"  # the reference's default
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert yaml.safe_load(open(os.path.join(root, "configs", "trainer", "default.yaml")))["precision_type"] == "no"
    # the module-level switch the trainer drives
    dit = MMDiT(simple_dit=True, input_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, patch_size=2, depth=1, n_classes=10)
    assert dit.precision == "bf16" and dit.precisions == ("bf16", "fp32")
    assert dit.set_precision("fp32").precision == "fp32" and copy.deepcopy(dit).precision == "fp32"
    with pytest.raises(ValueError):
        dit.set_precision("fp64")
    unet = UNetModel(image_size=[32, 32], in_channels=1, model_channels=32, out_channels=1, num_res_blocks=1, attention_resolutions=[],
                     channel_mult="1,2", n_classes=10)
    assert unet.precisions == ("bf16", "fp32") and unet.set_precision("fp32").precision == "fp32"
    from diffulab_amd import DDT, SprintDiT

    sprint = SprintDiT(simple_dit=True, input_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, patch_size=2, n_classes=10,
                       encoder_depth=1, deep_layers_depth=1, decoder_depth=1)
    assert sprint.precisions == ("bf16", "fp32") and sprint.set_precision("fp32").precision == "fp32"
    ddt = DDT(simple_ddt=True, input_channels=4, inner_dim=128, num_heads=2, patch_size=2, n_classes=10, encoder_depth=1, decoder_depth=1)
    assert ddt.precisions == ("bf16", "fp32") and ddt.set_precision("fp32").precision == "fp32"
    from diffulab_amd import PrecomputedEmbedder

    joint = MMDiT(simple_dit=False, input_channels=4, inner_dim=128, embedding_dim=128, num_heads=2, patch_size=2, depth=1,
                  rope_axes_dim=[16, 24, 24], context_embedder=PrecomputedEmbedder(torch.zeros(8, 64), 4))
    with pytest.raises(NotImplementedError, match="fp32"):  # the joint text-image forms keep the bf16 regime (their configs pin it)
        joint.set_precision("fp32")


def test_average_meter_keeps_the_reference_surface():
    """reference training/utils.py:1-25: keys / avg / sum / count dictionaries, update(val, key, n), reset() -- plain floats and 0-d CPU
    tensors enter at once; (0-d DEVICE tensors are read back when the meter is looked at: tests/test_trainer_gpu.py)"""
    from diffulab_amd.training.utils import AverageMeter

    m = AverageMeter()
    m.update(1.5, "train/loss")
    m.update(torch.tensor(2.5), "train/loss", n=3)
    m.update(4.0, "val/loss")
    assert m.keys == ["train/loss", "val/loss"] and m.avg == {"train/loss": 2.25, "val/loss": 4.0}
    assert m.sum == {"train/loss": 9.0, "val/loss": 4.0} and m.count == {"train/loss": 4, "val/loss": 1}
    m.reset()
    assert m.avg == {"train/loss": 0, "val/loss": 0} and m.count["train/loss"] == 0
    m.update(3.0, "train/loss")
    assert m.avg["train/loss"] == 3.0
