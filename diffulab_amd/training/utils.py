from __future__ import annotations

from typing import Any


class AverageMeter:
    """running means keyed by name (reference training/utils.py:1-25: ``keys`` / ``avg`` / ``sum`` / ``count``, ``update(val, key, n)``,
    ``reset()``).

    ``update`` also takes a 0-d DEVICE tensor (the trainer hands in ``loss.detach()`` instead of ``loss.item()``): the value is read
    back LATER -- when ``avg`` / ``sum`` / ``count`` are looked at, at ``reset()``, or after ``MAX_PENDING`` updates -- with one stacked
    device-to-host copy per flush, and then enters the sums in the order of the calls, as the same Python floats ``.item()`` would
    have produced.  The reference's per-step ``loss.item()`` (base_trainer.py:122) is a host synchronisation between the forward and
    the backward of EVERY step; nothing reads the meter before the end of an epoch."""

    MAX_PENDING = 256

    def __init__(self) -> None:
        self.keys: list[str] = []
        self._avg: dict[str, float] = {}
        self._sum: dict[str, float] = {}
        self._count: dict[str, int] = {}
        self._pending: list[tuple[Any, str, int]] = []

    # the three dictionaries of the reference, flushed on access
    @property
    def avg(self) -> dict[str, float]:
        self.flush()
        return self._avg

    @property
    def sum(self) -> dict[str, float]:
        self.flush()
        return self._sum

    @property
    def count(self) -> dict[str, int]:
        self.flush()
        return self._count

    def reset(self) -> None:
        self.flush()
        for k in self.keys:
            self._avg[k], self._sum[k], self._count[k] = 0, 0, 0

    def _add(self, val: float, key: str, n: int) -> None:
        if key not in self.keys:
            self.keys.append(key)
            self._sum[key], self._count[key] = 0.0, 0
        self._sum[key] += val * n
        self._count[key] += n
        self._avg[key] = self._sum[key] / self._count[key]

    def update(self, val: Any, key: str, n: int = 1) -> None:
        if hasattr(val, "is_cuda") and val.is_cuda and val.dim() == 0:
            if key not in self.keys:  # (the key exists from the first call on, like in the reference)
                self.keys.append(key)
                self._sum[key], self._count[key] = 0.0, 0
                self._avg[key] = 0
            self._pending.append((val.detach(), key, n))
            if len(self._pending) >= self.MAX_PENDING:
                self.flush()
            return
        self.flush()
        self._add(float(val), key, n)

    def flush(self) -> None:
        if not self._pending:
            return
        import torch

        pend, self._pending = self._pending, []
        vals = torch.stack([v.float() if v.dtype != torch.float64 else v for v, _, _ in pend]).tolist()  # one copy, one sync
        for v, (_, key, n) in zip(vals, pend):
            self._add(v, key, n)
