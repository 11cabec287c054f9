"""Pinned-memory H2D prefetch around a DataLoader (the data edge of SURVEY §8 f4).

The reference leaves host->device copies to ``accelerator.prepare(dataloader)`` (synchronous ``.to(device)`` per batch).  Here the
NEXT batch is staged in pinned host memory and copied on a side HIP stream while the current step runs; the consumer's stream
waits on the copy's event, so the copy costs the step nothing (160 KB/img of f32 latents at 10 k img/s is 1.6 GB/s of PCIe)."""

from __future__ import annotations

from typing import Any, Iterable, Iterator

import torch
from torch import Tensor


def _map(obj: Any, fn) -> Any:
    if isinstance(obj, Tensor):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)) and obj and isinstance(obj[0], (Tensor, dict)):
        return type(obj)(_map(v, fn) for v in obj)
    return obj


class DevicePrefetcher:
    """iterates ``loader`` one batch ahead; tensors of a yielded batch already live on ``device``"""

    def __init__(self, loader: Iterable, device: torch.device | str = "cuda") -> None:
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

    def __len__(self) -> int:
        return len(self.loader)  # type: ignore[arg-type]

    def _stage(self, batch: Any):
        if self.stream is None:
            return batch, None
        with torch.cuda.stream(self.stream):
            moved = _map(batch, lambda t: (t if t.is_pinned() else t.pin_memory()).to(self.device, non_blocking=True))
            ev = self.stream.record_event()
        return moved, ev

    def __iter__(self) -> Iterator:
        it = iter(self.loader)
        nxt = next(it, None)
        staged = self._stage(nxt) if nxt is not None else None
        while staged is not None:
            batch, ev = staged
            nxt = next(it, None)
            staged = self._stage(nxt) if nxt is not None else None  # the copy of batch i+1 overlaps the step on batch i
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
                _map(batch, lambda t: t.record_stream(torch.cuda.current_stream(self.device)) or t)
            yield batch
