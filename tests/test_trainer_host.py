"""Host-side logic of the trainer mirror (no GPU): config composition, ``_target_`` resolution, EMA decay schedule of
ema_pytorch (restated), batch sharding with ``split_batches=True`` semantics, accumulation boundaries."""

import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_composition_and_instantiate():
    from diffulab_amd.config import instantiate, load_config

    c = load_config(os.path.join(ROOT, "configs"), "train_mnist_ddpm", ["trainer.n_epoch=3", "model.model_channels=64"])
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
    t.world, t.rank = 4, 2
    b = {"model_inputs": {"x": torch.arange(8)[:, None], "y": torch.arange(8), "p": 0.1}, "extra": {"captions": list("abcdefgh")}}
    s = t.shard_batch(b)
    assert s["model_inputs"]["x"].flatten().tolist() == [4, 5] and s["model_inputs"]["y"].tolist() == [4, 5]
    assert s["model_inputs"]["p"] == 0.1 and s["extra"]["captions"] == ["e", "f"]
    assert (tmp_path / "p").is_dir()
