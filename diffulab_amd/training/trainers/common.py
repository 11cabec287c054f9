"""Trainer base: process group, device placement, gradient accumulation, checkpoints and logging.

Mirrors ``training/trainers/common.py:25-271`` of the reference (same constructor kwargs, ``move_dict_to_device``,
``save_model`` file names, ``log_images``), with HuggingFace Accelerate replaced by the MI355X-native runtime:
  * one process per GPU, ``torch.distributed`` over RCCL (backend "nccl") when WORLD_SIZE > 1 -- launched by
    ``python -m torch.distributed.run`` exactly like ``accelerate launch`` would;
  * the DDP wrap of ``accelerator.prepare`` becomes ``training.dp.GradReducer`` on the flat gradient arena (bucketed in-place
    all-reduce on a side stream, 1/world folded into the fused AdamW), parameters broadcast from rank 0 at start;
  * ``split_batches=True`` semantics (common.py:104): the DataLoader batch is the GLOBAL batch, each rank takes its contiguous slice;
  * ``accelerator.accumulate`` semantics, INCLUDING the reference's ordering quirk: ``training_step`` opens
    with ``optimizer.zero_grad()`` (base_trainer.py:138), which Accelerate gates on ``sync_gradients`` of the CURRENT micro-step, so
    on the synchronising micro-step the gradients of the k-1 earlier micro-batches are zeroed before its backward and the update
    applies (1/k) * grad(last micro-batch).  That is the default here (``reference_accumulation = True``; pinned by
    ``tests/golden/accum_k2.npz``, produced with accelerate itself).  ``DIFFULAB_TRUE_ACCUMULATION=1`` (or
    ``trainer.reference_accumulation = False``) switches to textbook accumulation: zero at the window's first micro-step, update
    with the mean gradient of the k micro-batches.  In both modes the loss is divided by k, only the synchronising micro-step
    reduces across ranks / steps the optimizer and the scheduler, and the last batch of a dataloader pass always synchronises
    (Accelerate's ``sync_with_dataloader``);
  * precision: ``precision_type="no"`` (the reference's default, common.py:76,105 / configs/trainer/default.yaml:4) selects the
    fp32-class regime -- f32 activations, exact-f32 MFMA products on the f32 parameters (engine_f32.py, unet_engine_f32.py,
    csrc/f32.hip) -- which exists for the class-conditional ``MMDiT`` / ``SprintDiT`` / ``DDT`` and ``UNetModel``; a denoiser without it raises at ``prepare`` and names the override
    (``trainer.precision_type=bf16``).  ``"bf16"`` = bf16 MFMA operands and activations with f32 accumulation, f32 master weights,
    f32 norm statistics / softmax / loss head (what accelerate's bf16 autocast computes); ``"fp16"`` / ``"fp8"`` are refused (not
    built: they would silently be something else);
  * ``compile`` / ``dynamo_plugin_kwargs`` are accepted and ignored (no tracing compiler: the launch sequences are static);
  * wandb (absent, no network) is replaced by a JSON-lines log under ``save_path/metrics.jsonl``.
"""

from __future__ import annotations

import json
import os
from abc import ABC, abstractmethod
from datetime import datetime
from pathlib import Path
from typing import TYPE_CHECKING, Any, Iterable

import torch
import torch.distributed as dist
from torch import Tensor
from torch.optim.lr_scheduler import LRScheduler
from torch.optim.optimizer import Optimizer

from ...datasets.base import BatchData
from ...diffuse.utils import to_device
from ..dp import GradReducer, broadcast_arena
from ..ema import EMA

if TYPE_CHECKING:
    from ...diffuse import Diffuser


class Trainer(ABC):
    def __init__(
        self,
        n_epoch: int,
        gradient_accumulation_step: int = 1,
        precision_type: str = "no",
        save_path: str | Path = Path.home() / "experiments" / f"{datetime.now().strftime('%Y%m%d_%H%M%S')}",
        project_name: str = "my_project",
        run_config: dict[str, Any] | None = None,
        init_kwargs: dict[str, Any] = {},
        use_ema: bool = False,
        ema_rate: float = 0.999,
        ema_update_after_step: int = 0,
        ema_update_every: int = 10,
        compile: bool = False,
        dynamo_plugin_kwargs: dict[str, Any] = {},
    ) -> None:
        if precision_type not in ("no", "bf16"):
            raise NotImplementedError(
                f"precision_type={precision_type!r}: the HIP path has two precision regimes -- 'no' (fp32: f32 activations, exact-f32 "
                "MFMA products) and 'bf16' (bf16 MFMA operands with f32 accumulation, f32 master weights, f32 norm statistics, softmax "
                "and loss).  An fp16 / fp8 run would not be what was asked for, so it is refused.")
        self.precision_type = precision_type
        self.n_epoch = n_epoch
        self.use_ema = use_ema
        self.ema_rate = ema_rate
        self.gradient_accumulation_step = max(1, int(gradient_accumulation_step))
        self.ema_update_after_step = ema_update_after_step * self.gradient_accumulation_step
        self.ema_update_every = ema_update_every * self.gradient_accumulation_step
        self.compile = compile

        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world > 1 and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0" and torch.cuda.is_initialized():
            # (diffulab_amd/__init__.py pins the variable at import; a script that initialised the GPU before importing the package
            # and runs on a host without legacy IPC would fail much later inside RCCL with `hipIpcGetMemHandle: invalid argument`)
            import logging

            logging.warning("HSA_ENABLE_IPC_MODE_LEGACY is not '0' and the HIP runtime is already initialised: export "
                            "HSA_ENABLE_IPC_MODE_LEGACY=0 in the launcher environment if RCCL fails with hipIpcGetMemHandle errors")
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
            self.device = torch.device("cuda", local)
        else:  # host-logic tests only: every denoiser forward raises without a GPU
            self.device = torch.device("cpu")
        if self.world > 1 and not dist.is_initialized():
            from ..dp import configure_rccl_env

            configure_rccl_env()
            dist.init_process_group(backend="nccl" if self.device.type == "cuda" else "gloo")
        self.is_main_process = self.rank == 0

        self.save_path = Path(save_path) / project_name
        if self.is_main_process:
            self.save_path.mkdir(parents=True, exist_ok=True)
            if run_config is not None:
                (self.save_path / "run_config.json").write_text(json.dumps(run_config, indent=1, default=str))
        self._micro = 0            # micro-steps run so far
        self._accum_step = 0       # position inside the accumulation window (accelerate's ``Accelerator.step``)
        self._end_of_dataloader = False
        self.reference_accumulation = os.environ.get("DIFFULAB_TRUE_ACCUMULATION", "0") != "1"
        self._reducer: GradReducer | None = None

    # ------------------------------------------------------------------ accelerate-equivalent plumbing
    def move_dict_to_device(self, batch: dict[str, Any]) -> dict[str, Any]:
        # (host tensors of a batch go through the pinned staging ring: a pageable `.to(device)` would block until the queue has drained)
        return {k: to_device(v, self.device, v.dtype) if isinstance(v, Tensor) else v for k, v in batch.items()}

    def shard_batch(self, batch: dict[str, Any]) -> dict[str, Any]:
        """split_batches=True (common.py:104): rank r keeps rows [r*B/W, (r+1)*B/W) of every tensor / list entry"""
        if self.world == 1:
            return batch
        out: dict[str, Any] = {}
        for k, v in batch.items():
            if isinstance(v, dict):
                out[k] = self.shard_batch(v)
            elif isinstance(v, (Tensor, list)) and len(v) > 0:
                if len(v) % self.world:  # (cannot happen behind iterate() / even_batches(): partial batches are completed there)
                    raise ValueError(f"split_batches: batch entry {k!r} has {len(v)} rows, not divisible by world size {self.world}")
                n = len(v) // self.world
                out[k] = v[self.rank * n : (self.rank + 1) * n]
            else:
                out[k] = v
        return out

    def prepare(self, diffuser: "Diffuser", optimizer: Optimizer) -> None:
        """accelerator.prepare(denoiser, ..., optimizer): device placement, rank-0 broadcast, gradient reducer"""
        den = diffuser.denoiser
        want = "fp32" if self.precision_type == "no" else "bf16"
        if hasattr(den, "set_precision"):
            try:
                den.set_precision(want)
            except NotImplementedError as e:
                raise NotImplementedError(f"{e}.  Set trainer.precision_type=bf16 for this denoiser (the reference's default "
                                          "precision_type='no' is the fp32 regime).") from None
        den.to(self.device)
        if hasattr(den, "engine") and self.device.type == "cuda":
            eng = den.engine  # flattens the parameters into the arena
            if self.is_main_process:  # run header of the JSON-lines log: which launch sequences this run trains on
                with open(self.save_path / "metrics.jsonl", "a") as f:
                    f.write(json.dumps({"run/precision_type": self.precision_type, "run/regime": getattr(den, "precision", "bf16"),
                                        "run/engine": type(eng).__name__, "run/world": self.world}) + "\n")
            if self.world > 1:
                broadcast_arena(den._flat)
                self._reducer = GradReducer(den._flat_grad)
                eng.reducer = self._reducer
                if hasattr(optimizer, "grad_scale"):
                    optimizer.grad_scale = self._reducer.grad_scale
                else:
                    raise RuntimeError("data-parallel training needs diffulab_amd.training.FusedAdamW (grad_scale = 1/world)")

    def broadcast_extra_losses(self, diffuser: "Diffuser") -> None:
        """accelerator.prepare(loss) DDP-wraps the auxiliary loss heads (REPA projector / resampler): every rank starts from
        rank 0's values.  Call after the heads were moved to the device."""
        if self.world == 1:
            return
        for loss in diffuser.extra_losses:
            for t in list(loss.parameters()) + list(loss.buffers()):
                dist.broadcast(t.data, src=0)

    @property
    def sync_gradients(self) -> bool:
        """accelerate ``_do_sync``: every k-th micro-step of the window, and always on the last batch of a dataloader pass"""
        return self._end_of_dataloader or (self._accum_step + 1) % self.gradient_accumulation_step == 0

    @property
    def window_start(self) -> bool:
        return self._accum_step % self.gradient_accumulation_step == 0

    def begin_micro_step(self) -> None:
        if self._reducer is not None:
            self._reducer.sync = self.sync_gradients

    def end_micro_step(self) -> None:
        self._micro += 1
        self._accum_step = 0 if self._end_of_dataloader else self._accum_step + 1

    @staticmethod
    def _rows(batch: Any) -> int | None:
        """row count of the first tensor / list entry of a (nested) batch dict"""
        for v in batch.values():
            n = Trainer._rows(v) if isinstance(v, dict) else (len(v) if isinstance(v, (Tensor, list)) and len(v) > 0 else None)
            if n is not None:
                return n
        return None

    @staticmethod
    def _complete(batch: Any, first: Any, missing: int) -> Any:
        """append `missing` rows taken from the start of `first` (wrapping around it) to every tensor / list entry"""
        out: dict[str, Any] = {}
        for k, v in batch.items():
            if isinstance(v, dict):
                out[k] = Trainer._complete(v, first[k], missing)
            elif isinstance(v, Tensor) and len(v) > 0:
                src = first[k]
                reps = -(-missing // len(src))
                out[k] = torch.cat([v, (src.repeat((reps,) + (1,) * (src.dim() - 1)) if reps > 1 else src)[:missing].to(v.device)])
            elif isinstance(v, list) and len(v) > 0:
                src = first[k]
                out[k] = v + (src * (-(-missing // len(src))))[:missing]
            else:
                out[k] = v
        return out

    def even_batches(self, dataloader: Iterable[BatchData]):
        """Accelerate's even_batches=True under split_batches=True (BatchSamplerShard._iter_with_split): with more than one
        process a last batch that is smaller than the batch size is completed with samples from the FIRST batch of the pass
        (wrapping around it if needed), so every rank gets batch_size / world rows and no rank runs out of data mid-epoch.
        One process: batches pass through untouched (a partial last batch stays partial, as in the reference)."""
        if self.world == 1:
            yield from dataloader
            return
        first, size = None, getattr(dataloader, "batch_size", None)  # (a torch DataLoader knows it; else: the first batch's rows)
        for batch in dataloader:
            if first is None:
                first, size = batch, size or self._rows(batch)
                if size is not None and size % self.world:
                    raise ValueError(f"split_batches=True: the batch size ({size}) must be a round multiple of the number of "
                                     f"processes ({self.world})")
            n = self._rows(batch)
            if size is not None and n is not None and n < size:
                batch = self._complete(batch, first, size - n)
            yield batch

    def iterate(self, dataloader: Iterable[BatchData]):
        """yields the batches of one pass and flags the last one (``GradientState.end_of_dataloader``), which forces a sync"""
        it = iter(self.even_batches(dataloader))
        batch = next(it, None)
        while batch is not None:
            nxt = next(it, None)
            self._end_of_dataloader = nxt is None
            yield batch
            batch = nxt
        self._end_of_dataloader = False

    def reduce_extra_grads(self, diffuser: "Diffuser") -> None:
        """data parallel: parameters of auxiliary loss heads (REPA projector) live outside the denoiser's gradient arena, so
        their gradients are summed here (the 1/world factor is the optimizer's grad_scale, like for the arena)"""
        if self.world == 1:
            return
        for loss in diffuser.extra_losses:
            grads = [p.grad for p in loss.parameters() if p.grad is not None]
            if grads:
                flat = torch.cat([g.reshape(-1) for g in grads])
                dist.all_reduce(flat)
                off = 0
                for g in grads:
                    g.copy_(flat[off : off + g.numel()].view_as(g))
                    off += g.numel()

    def gather_mean(self, value: float) -> float:
        if self.world == 1:
            return float(value)
        t = torch.tensor([value], device=self.device, dtype=torch.float64)
        dist.all_reduce(t)
        return float(t.item() / self.world)

    def log(self, values: dict[str, float], step: int) -> None:
        if self.is_main_process:
            with open(self.save_path / "metrics.jsonl", "a") as f:
                f.write(json.dumps({"step": step, **values}) + "\n")

    def wait_for_everyone(self) -> None:
        if self.world > 1:
            dist.barrier()

    # ------------------------------------------------------------------ checkpoints (file names of common.py:156-176)
    def save_model(self, optimizer: Optimizer, diffuser: "Diffuser", ema_denoiser: EMA | None = None,
                   scheduler: LRScheduler | None = None) -> None:
        if not self.is_main_process:
            return
        cpu = lambda sd: {k: (v.detach().cpu() if isinstance(v, Tensor) else v) for k, v in sd.items()}  # noqa: E731
        torch.save(cpu(diffuser.denoiser.state_dict()), self.save_path / "denoiser.pt")
        torch.save(optimizer.state_dict(), self.save_path / "optimizer.pt")
        if ema_denoiser is not None:
            torch.save(cpu(ema_denoiser.ema_model.state_dict()), self.save_path / "ema.pt")
        if scheduler is not None:
            torch.save(scheduler.state_dict(), self.save_path / "scheduler.pt")
        for extra_loss in diffuser.extra_losses:
            torch.save(cpu(extra_loss.state_dict()), self.save_path / f"{extra_loss.name}.pt")

    @torch.no_grad()
    def log_images(self, diffuser: "Diffuser", val_dataloader: Iterable[BatchData], epoch: int, val_steps: int = 50,
                   step_shift: float | None = None, guidance_scale: float = 0) -> None:
        """common.py:178-271: sample with `val_steps` steps from the labels of one validation batch; the images (mapped from
        [-1,1] to [0,1]) are written to ``save_path/val_images_epoch{N}.pt`` instead of a wandb panel."""
        batch = dict(next(iter(val_dataloader))["model_inputs"])
        batch = self.move_dict_to_device(batch)
        x: Tensor = batch.pop("x")
        original_steps = diffuser.n_steps
        train_shift = None
        if step_shift is not None:
            train_shift = diffuser.diffusion.shift
            diffuser.set_steps(val_steps, shift=step_shift)
        else:
            diffuser.set_steps(val_steps)
        images = diffuser.generate(data_shape=tuple(x.shape), model_inputs=batch, guidance_scale=guidance_scale, use_tqdm=False)["x"]
        images = (images * 0.5 + 0.5).clamp(0, 1).cpu().float()
        torch.save(images, self.save_path / f"val_images_epoch{epoch + 1}.pt")
        if train_shift is not None:
            diffuser.set_steps(original_steps, shift=train_shift)
        else:
            diffuser.set_steps(original_steps)

    @abstractmethod
    def training_step(self, *args: Any, **kwargs: Any) -> None: ...

    @abstractmethod
    def validation_step(self, *args: Any, **kwargs: Any) -> None: ...

    @abstractmethod
    def train(self, *args: Any, **kwargs: Any) -> None: ...
