"""Carriers shared by the diffusion heads (reference diffuse/utils.py:6-28, samplers/common.py:7-32)."""

from __future__ import annotations

from typing import TypedDict

import torch
from torch import Tensor

try:
    from typing import NotRequired, Required
except ImportError:
    from typing_extensions import NotRequired, Required


class SamplingOutput(TypedDict, total=False):
    x: Required[Tensor]
    estimated_x0: NotRequired[Tensor]
    xt: NotRequired[Tensor]
    xt_mean: NotRequired[Tensor]
    xt_std: NotRequired[Tensor]
    logprob: NotRequired[Tensor]


def f32_table(table_fp64: Tensor, device: torch.device) -> Tensor:
    """fp64 schedule table -> fp32 device table: the same rounding `extract_into_tensor` applies per gather
    (diffuse/utils.py:16 `.float()`), hoisted out of the loop."""
    return table_fp64.to(torch.float32).to(device).contiguous()


class _PinnedRing:
    """pinned staging slots for small host -> device transfers (per-step timestep vectors): a pageable `.to(device)` blocks the host
    until the stream has drained everything queued before it, i.e. once per training step the host loses its whole lead over the GPU
    -- and a step of short kernels (the UNet: 800 launches of 10-90 us) then runs its forward at the host's launch rate.  A slot is
    reused only after the copy that last read it has completed (an event per slot), so the host may run several steps ahead."""

    SLOTS = 8
    CAP_BYTES = 512 << 20  # pinned host memory the ring may hold; least-recently-used size classes are dropped beyond it

    def __init__(self) -> None:
        import collections
        import threading

        # size class (bytes rounded up to a power of two, device) -> [next slot, [(pinned byte buffer, event)], [used]]: multi-aspect-
        # ratio buckets and short last batches share a class instead of pinning a ring per shape
        self.slots: "collections.OrderedDict[tuple, list]" = collections.OrderedDict()
        self.bytes = 0
        self.lock = threading.Lock()  # (a prefetch thread and the training loop may both stage transfers)

    def put(self, t: Tensor, device: torch.device, dtype: torch.dtype) -> Tensor:
        nbytes = t.numel() * torch.empty((), dtype=dtype).element_size()
        cls = 1 << max(6, (nbytes - 1).bit_length())
        key = (cls, device)
        with self.lock:
            ring = self.slots.get(key)
            if ring is None:
                n = self.SLOTS if cls <= (1 << 18) else 3  # (batch-sized tensors: three slots are enough)
                while self.slots and self.bytes + n * cls > self.CAP_BYTES:
                    _, old = self.slots.popitem(last=False)
                    for (buf, ev), used in zip(old[1], old[2]):
                        if used:
                            ev.synchronize()  # (its last copy must have read the buffer before the memory is unpinned)
                        self.bytes -= buf.numel()
                ring = self.slots[key] = [0, [(torch.empty(cls, dtype=torch.uint8).pin_memory(), torch.cuda.Event()) for _ in range(n)], [False] * n]
                self.bytes += n * cls
            else:
                self.slots.move_to_end(key)
            i = ring[0]
            raw, ev = ring[1][i]
            if ring[2][i]:
                ev.synchronize()  # (eight transfers ago: completed long since unless the host is that far ahead)
            buf = raw[:nbytes].view(dtype).view(t.shape)
            buf.copy_(t)  # host-side conversion + copy into the pinned slot
            out = buf.to(device, non_blocking=True)
            ev.record(torch.cuda.current_stream(device))
            ring[2][i] = True
            ring[0] = (i + 1) % len(ring[1])
        return out


_RING = _PinnedRing()


def to_device(t: Tensor, device: torch.device | str, dtype: torch.dtype) -> Tensor:
    """`t.to(device=device, dtype=dtype).contiguous()` without the host-side synchronisation of a pageable copy (small CPU tensors go
    through a ring of pinned slots -- timestep vectors, batches of up to 32 MB --, everything else is the plain call): same values,
    same dtype conversion"""
    device = torch.device(device)
    if t.device.type == "cpu" and device.type == "cuda" and t.numel() * t.element_size() <= (32 << 20) and not t.requires_grad and t.numel() > 0:
        return _RING.put(t.detach(), device, dtype)
    return t.to(device=device, dtype=dtype).contiguous()

