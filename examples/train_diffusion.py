"""Training entry point in the shape of the reference's examples/train_diffusion.py (same imports, same Hydra composition, same
objects in the same order: datasets -> DataLoader -> denoiser -> Diffuser -> optimizer -> BaseTrainer.train).

    python examples/train_diffusion.py --config-name train_mnist_ddpm dataset=mnist_synthetic trainer.n_epoch=1
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_diffusion.py \\
        --config-name train_dit_s2_flow_matching

The reference's OWN script runs unmodified through ``python -m diffulab.run <script> [overrides]``.
hydra-core / omegaconf are used when installed; this image has neither, so the stand-ins of diffulab_amd.compat.hydra_shim are
registered under those names (no network to install them).
"""

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import hydra
except ImportError:
    from diffulab_amd.compat import hydra_shim

    hydra_shim.install()
    import hydra
import torch
from hydra.utils import instantiate
from omegaconf import DictConfig, OmegaConf
from torch.utils.data import DataLoader

from diffulab.diffuse import Diffuser
from diffulab.training import BaseTrainer


@hydra.main(version_base=None, config_path="../configs", config_name="train_mnist_flow_matching")
def train(cfg: DictConfig):
    print(OmegaConf.to_yaml(cfg))
    train_dataset = instantiate(cfg.dataset.train)
    val_dataset = instantiate(cfg.dataset.val)
    dl_cfg = cfg.get("dataloader", {})
    loader = lambda ds, shuffle: DataLoader(dataset=ds, batch_size=dl_cfg.get("batch_size", 32), shuffle=shuffle,  # noqa: E731
                                            num_workers=dl_cfg.get("num_workers", 0), pin_memory=dl_cfg.get("pin_memory", False),
                                            drop_last=dl_cfg.get("drop_last", False))
    train_loader, val_loader = loader(train_dataset, dl_cfg.get("shuffle", True)), loader(val_dataset, False)

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
        run_config=OmegaConf.to_container(cfg, resolve=True),
        compile=cfg.trainer.get("compile", False),
        init_kwargs={"wandb": cfg.trainer.get("wandb", {})},
        **({"save_path": cfg.trainer.save_path} if "save_path" in cfg.trainer else {}),
    )
    if os.environ.get("DIFFULAB_DRY_RUN") == "1":  # construct everything, train nothing (CPU smoke of the entry point)
        print(f"dry run: {type(denoiser).__name__} / {type(optimizer).__name__} / {len(train_loader)} train batches")
        return
    trainer.train(diffuser=diffuser, optimizer=optimizer, train_dataloader=train_loader, val_dataloader=val_loader,
                  log_validation_images=cfg.trainer.log_validation_images, val_steps=cfg.trainer.get("val_steps", 50))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    train()
