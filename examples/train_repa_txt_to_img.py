"""Counterpart of the reference's examples/train_repa_txt_to_img.py: text-to-image SPRINT DiT (joint text-image blocks) + REPA loss,
with precomputed text embeddings behind a PrecomputedEmbedder.

    python examples/train_repa_txt_to_img.py train_imagenet_repa_txt_to_img_sprint dataloader.batch_size=16
"""

import os
import sys

import torch
from torch.utils.data import DataLoader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from diffulab_amd.config import instantiate, load_config  # noqa: E402
from diffulab_amd.diffuse import Diffuser  # noqa: E402
from diffulab_amd.networks.embedders import PrecomputedEmbedder  # noqa: E402
from diffulab_amd.training import BaseTrainer  # noqa: E402
from diffulab_amd.training.losses import RepaLoss  # noqa: E402


def train(config_name: str, overrides: list[str]) -> None:
    cfg = load_config(os.path.join(ROOT, "configs"), config_name, overrides)
    train_dataset, val_dataset = instantiate(cfg.dataset.train), instantiate(cfg.dataset.val)
    ec = cfg.embedder
    g = torch.Generator().manual_seed(7)  # (the reference loads the null embedding from a file written offline)
    embedder = PrecomputedEmbedder(torch.randn(1, ec.context_len, ec.context_dim, generator=g) * 0.5, ec.null_embedding_seq_len)
    denoiser = instantiate(cfg.model, context_embedder=embedder)
    print(f"Number of trainable parameters: {sum(p.numel() for p in denoiser.parameters() if p.requires_grad):,}")
    rp = cfg.get("repa", {})
    repa_loss = RepaLoss(denoiser_dimension=cfg.model.inner_dim, alignment_layer=rp.get("alignment_layer", 2),
                         hidden_dim=rp.get("hidden_dim", 1024), embedding_dim=rp.get("embedding_dim", 384), load_dino=False,
                         use_resampler=cfg.perceiver_resampler.get("use_resampler", False),
                         resampler_params=dict(cfg.perceiver_resampler.get("parameters", {})) or None, coeff=rp.get("coeff", 0.5))
    dl_cfg = cfg.get("dataloader", {})
    mk = lambda ds, shuffle: DataLoader(dataset=ds, batch_size=dl_cfg.get("batch_size", 32), shuffle=shuffle,  # noqa: E731
                                        num_workers=dl_cfg.get("num_workers", 0), pin_memory=dl_cfg.get("pin_memory", False),
                                        drop_last=True)
    train_loader, val_loader = mk(train_dataset, dl_cfg.get("shuffle", True)), mk(val_dataset, False)
    diffuser = Diffuser(denoiser=denoiser, model_type=cfg.diffuser.model_type, n_steps=cfg.diffuser.n_steps,
                        sampling_method=cfg.diffuser.sampling_method, extra_args=dict(cfg.diffuser.get("extra_args", {})),
                        extra_losses=[repa_loss])
    optimizer = instantiate(cfg.optimizer, params=list(denoiser.parameters()) + list(repa_loss.parameters()))
    trainer = BaseTrainer(
        n_epoch=cfg.trainer.n_epoch, gradient_accumulation_step=cfg.trainer.gradient_accumulation_step,
        precision_type=cfg.trainer.precision_type, project_name=cfg.trainer.project_name, use_ema=cfg.trainer.use_ema,
        ema_rate=cfg.trainer.get("ema_rate", 0.9999), ema_update_after_step=cfg.trainer.get("ema_update_after_step", 0),
        ema_update_every=cfg.trainer.get("ema_update_every", 10), run_config=cfg, compile=cfg.trainer.get("compile", False),
        **({"save_path": cfg.trainer.save_path} if "save_path" in cfg.trainer else {}))
    trainer.train(diffuser=diffuser, optimizer=optimizer, train_dataloader=train_loader, val_dataloader=val_loader,
                  log_validation_images=cfg.trainer.log_validation_images, val_steps=cfg.trainer.get("val_steps", 50),
                  p_classifier_free_guidance=cfg.trainer.get("p_classifier_free_guidance", 0.1))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    train(sys.argv[1] if len(sys.argv) > 1 else "train_imagenet_repa_txt_to_img_sprint", sys.argv[2:])
