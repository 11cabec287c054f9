"""Counterpart of the reference's examples/train_diffusion.py on the MI355X build: same flow (datasets -> DataLoader ->
denoiser -> Diffuser -> optimizer -> BaseTrainer.train), YAML configs of the same shape under ../configs.

    python examples/train_diffusion.py train_mnist_ddpm trainer.n_epoch=1 dataloader.batch_size=64
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_diffusion.py train_dit_s2_flow_matching
"""

import os
import sys

import torch
from torch.utils.data import DataLoader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from diffulab_amd.config import instantiate, load_config  # noqa: E402
from diffulab_amd.diffuse import Diffuser  # noqa: E402
from diffulab_amd.training import BaseTrainer  # noqa: E402


def train(config_name: str, overrides: list[str]) -> None:
    cfg = load_config(os.path.join(ROOT, "configs"), config_name, overrides)
    train_dataset = instantiate(cfg.dataset.train)
    val_dataset = instantiate(cfg.dataset.val)
    dl_cfg = cfg.get("dataloader", {})
    mk = lambda ds, shuffle: DataLoader(dataset=ds, batch_size=dl_cfg.get("batch_size", 32), shuffle=shuffle,  # noqa: E731
                                        num_workers=dl_cfg.get("num_workers", 0), pin_memory=dl_cfg.get("pin_memory", False),
                                        drop_last=True)
    train_loader, val_loader = mk(train_dataset, dl_cfg.get("shuffle", True)), mk(val_dataset, False)

    denoiser = instantiate(cfg.model)
    print(f"Number of trainable parameters: {sum(p.numel() for p in denoiser.parameters() if p.requires_grad):,}")
    diffuser = Diffuser(denoiser=denoiser, model_type=cfg.diffuser.model_type, n_steps=cfg.diffuser.n_steps,
                        sampling_method=cfg.diffuser.sampling_method, extra_args=dict(cfg.diffuser.get("extra_args", {})))
    optimizer = instantiate(cfg.optimizer, params=denoiser.parameters())
    trainer = BaseTrainer(
        n_epoch=cfg.trainer.n_epoch,
        gradient_accumulation_step=cfg.trainer.gradient_accumulation_step,
        precision_type=cfg.trainer.precision_type,
        project_name=cfg.trainer.project_name,
        use_ema=cfg.trainer.use_ema,
        ema_update_after_step=cfg.trainer.get("ema_update_after_step", 0),
        ema_update_every=cfg.trainer.get("ema_update_every", 10),
        run_config=cfg,
        compile=cfg.trainer.get("compile", False),
        **({"save_path": cfg.trainer.save_path} if "save_path" in cfg.trainer else {}),
    )
    trainer.train(diffuser=diffuser, optimizer=optimizer, train_dataloader=train_loader, val_dataloader=val_loader,
                  log_validation_images=cfg.trainer.log_validation_images, val_steps=cfg.trainer.get("val_steps", 50))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    train(sys.argv[1] if len(sys.argv) > 1 else "train_mnist_ddpm", sys.argv[2:])
