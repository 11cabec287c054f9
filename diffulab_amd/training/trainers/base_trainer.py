"""``BaseTrainer``: the training loop of the reference (training/trainers/base_trainer.py:104-399) on the MI355X runtime.

``training_step`` keeps the reference's order of operations (base_trainer.py:138-153): zero_grad -> draw timesteps (CPU generator)
-> compute_loss -> every loss into the tracker (the reference's ``.item()``; here a device scalar the meter reads back lazily) -> backward -> optimizer.step -> scheduler -> EMA update.
What Accelerate did implicitly is spelled out (see trainers/common.py): loss / gradient_accumulation_step before backward,
``zero_grad`` / ``step`` / scheduler gated on the synchronising micro-step (so with k > 1 the default reproduces the reference's
"(1/k) * gradient of the last micro-batch" update), EMA on every micro-step, gradient all-reduce overlapped
with backward.
"""

from __future__ import annotations

import logging
from contextlib import contextmanager
from typing import TYPE_CHECKING, Iterable

import torch
from torch.optim.lr_scheduler import LRScheduler
from torch.optim.optimizer import Optimizer

from ...datasets.base import BatchData
from ..ema import EMA
from ...diffuse.utils import to_device
from ..utils import AverageMeter
from .common import Trainer

if TYPE_CHECKING:
    from ...diffuse import Diffuser


class BaseTrainer(Trainer):
    def training_step(
        self,
        diffuser: "Diffuser",
        optimizer: Optimizer,
        batch: BatchData,
        tracker: AverageMeter,
        p_classifier_free_guidance: float = 0,
        scheduler: LRScheduler | None = None,
        per_batch_scheduler: bool = False,
        ema_denoiser: EMA | None = None,
    ) -> None:
        self.begin_micro_step()
        # AcceleratedOptimizer.zero_grad() only acts while gradient_state.sync_gradients is set, and accumulate() sets that flag
        # for the micro-step being entered: the reference therefore clears the window's gradients right before the LAST backward
        if self.sync_gradients if self.reference_accumulation else self.window_start:
            optimizer.zero_grad()
        batch = self.shard_batch(batch)
        model_inputs = self.move_dict_to_device(dict(batch["model_inputs"]))
        t_host = diffuser.draw_timesteps(model_inputs["x"].shape[0])
        timesteps = to_device(t_host, self.device, t_host.dtype)  # (no host synchronisation: pinned staging ring, diffuse/utils.py)
        model_inputs.update({"p": p_classifier_free_guidance})
        extra = self.move_dict_to_device(dict(batch.get("extra", {})))
        losses = diffuser.compute_loss(model_inputs=model_inputs, timesteps=timesteps, extra_args=extra)
        for key, loss in losses.items():
            # the reference reads `loss.item()` here (base_trainer.py:122): a host synchronisation between forward and backward of
            # every step; the meter takes the device scalar and reads it back when it is looked at (same floats, same order)
            tracker.update(loss.detach(), key=f"train/{key}")
        loss = sum(losses.values())
        (loss / self.gradient_accumulation_step).backward()
        if self.sync_gradients:
            self.reduce_extra_grads(diffuser)
            optimizer.step()
            if scheduler is not None and per_batch_scheduler:
                scheduler.step()
        if ema_denoiser is not None:
            ema_denoiser.update()  # counts micro-steps: update_after_step / update_every were scaled in __init__ (common.py:97-98)
        self.end_micro_step()

    @torch.no_grad()
    def validation_step(self, diffuser: "Diffuser", val_batch: BatchData, tracker: AverageMeter) -> None:
        val_batch = self.shard_batch(val_batch)
        model_inputs = self.move_dict_to_device(dict(val_batch["model_inputs"]))
        timesteps = diffuser.draw_timesteps(model_inputs["x"].shape[0]).to(self.device)
        extra_args = self.move_dict_to_device(dict(val_batch.get("extra", {})))
        val_losses = diffuser.compute_loss(model_inputs=model_inputs, timesteps=timesteps, extra_args=extra_args)
        for key, val_loss in val_losses.items():
            tracker.update(val_loss.detach(), key=f"val/{key}")  # (read back when the meter is looked at, like the training losses)

    def train(
        self,
        diffuser: "Diffuser",
        optimizer: Optimizer,
        train_dataloader: Iterable[BatchData],
        val_dataloader: Iterable[BatchData] | None = None,
        scheduler: LRScheduler | None = None,
        per_batch_scheduler: bool = False,
        log_validation_images: bool = True,
        train_embedder: bool = False,
        p_classifier_free_guidance: float = 0.2,
        val_steps: int = 50,
        val_step_shift: float | None = None,
        optimizer_ckpt: str | None = None,
        denoiser_ckpt: str | None = None,
        ema_ckpt: str | None = None,
        epoch_start: int = 0,
    ) -> None:
        if val_step_shift is not None:
            assert diffuser.model_type == "rectified_flow", "Time-shifting during validation is only supported for flow-based models."
        p_cfg = p_classifier_free_guidance if diffuser.denoiser.classifier_free else 0
        ema = self._setup_run(diffuser, optimizer, train_embedder, optimizer_ckpt, denoiser_ckpt, ema_ckpt)
        meter, best = AverageMeter(), float("inf")
        logging.info("Begin training")
        for epoch in range(epoch_start, self.n_epoch):
            self._train_epoch(diffuser, optimizer, train_dataloader, meter, p_cfg, scheduler, per_batch_scheduler, ema)
            self._log_group(meter, "train/", epoch + 1)
            meter.reset()
            if val_dataloader is not None:
                with self._validating(diffuser, ema) as online:
                    for val_batch in self.even_batches(val_dataloader):
                        self.validation_step(diffuser=diffuser, val_batch=val_batch, tracker=meter)
                    val_total = self._log_group(meter, "val/", epoch + 1)
                    if log_validation_images and self.is_main_process:
                        logging.info("creating validation images")
                        self.log_images(diffuser, val_dataloader, epoch, val_steps, step_shift=val_step_shift,
                                        guidance_scale=4 if online.classifier_free else 0)
                if val_total < best:  # (the checkpoint holds the online weights AND the EMA copy, base_trainer.py:246-251)
                    best = val_total
                    self.save_model(optimizer, diffuser, ema, scheduler)
                meter.reset()
            self.wait_for_everyone()
        logging.info("Training complete")

    # ---- the pieces of train() (reference: one function, base_trainer.py:277-399; same order of effects)
    def _setup_run(self, diffuser: "Diffuser", optimizer: Optimizer, train_embedder: bool, optimizer_ckpt: str | None,
                   denoiser_ckpt: str | None, ema_ckpt: str | None) -> EMA | None:
        """checkpoints in, device placement / rank-0 broadcast / reducer (``prepare``), the EMA copy, auxiliary loss heads hooked
        onto the denoiser, a frozen context embedder unless it is trained"""
        if denoiser_ckpt:
            diffuser.denoiser.load_state_dict(torch.load(denoiser_ckpt))
        self.prepare(diffuser, optimizer)
        if optimizer_ckpt:  # after prepare(): the state follows the parameters' device (and the arena exists)
            optimizer.load_state_dict(torch.load(optimizer_ckpt, weights_only=False))
        ema = None
        if self.use_ema:
            ema = EMA(diffuser.denoiser, beta=self.ema_rate, update_after_step=self.ema_update_after_step,
                      update_every=self.ema_update_every).to(self.device)
            if ema_ckpt:
                ema.ema_model.load_state_dict(torch.load(ema_ckpt, weights_only=True))
        for head in diffuser.extra_losses:  # accelerator.prepare(loss) in the reference: device placement, then the hooks
            head.to(self.device)
            head.set_model(diffuser.denoiser)
        self.broadcast_extra_losses(diffuser)
        embedder = getattr(diffuser.denoiser, "context_embedder", None)
        if embedder is not None and not train_embedder:
            for q in embedder.parameters():
                q.requires_grad = False
        return ema

    def _train_epoch(self, diffuser: "Diffuser", optimizer: Optimizer, loader: Iterable[BatchData], meter: AverageMeter, p_cfg: float,
                     scheduler: LRScheduler | None, per_batch_scheduler: bool, ema: EMA | None) -> None:
        diffuser.train()
        for batch in self.iterate(loader):
            self.training_step(diffuser=diffuser, optimizer=optimizer, batch=batch, tracker=meter, p_classifier_free_guidance=p_cfg,
                               scheduler=scheduler, per_batch_scheduler=per_batch_scheduler, ema_denoiser=ema)
        if scheduler is not None and not per_batch_scheduler:
            scheduler.step()

    def _log_group(self, meter: AverageMeter, prefix: str, step: int) -> float:
        """log the rank-mean of every tracked average under ``prefix``; returns their sum (the validation criterion)"""
        total = 0.0
        for key, value in meter.avg.items():
            if key.startswith(prefix):
                mean = self.gather_mean(value)
                self.log({key: mean}, step=step)
                total += mean
        return total

    @contextmanager
    def _validating(self, diffuser: "Diffuser", ema: EMA | None):
        """eval mode with the EMA weights in place of the online ones (and the auxiliary heads hooked onto them); yields the online
        denoiser and puts everything back on exit"""
        diffuser.eval()
        online = diffuser.denoiser
        if ema is not None:
            diffuser.denoiser = ema.ema_model.eval()
            for head in diffuser.extra_losses:
                head.set_model(ema.ema_model)
        try:
            yield online
        finally:
            if ema is not None:
                diffuser.denoiser = online
                for head in diffuser.extra_losses:
                    head.set_model(online)
