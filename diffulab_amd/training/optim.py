"""Fused AdamW over the flat parameter arena: ONE `dl_adamw_step` launch per optimizer step.

Drop-in `torch.optim.Optimizer` (same hyper-parameters / update rule as `torch.optim.AdamW`, which
`configs/optimizer/adamw.yaml` of the reference instantiates).  When the parameters are views of one flat f32
arena (what `MMDiT.flatten_parameters` sets up) the whole model is updated by a single kernel; otherwise it
launches the same HIP kernel once per tensor.
"""

from __future__ import annotations

import torch
from torch import Tensor

from .. import ops


def _arena_of(params: list[Tensor], use_grad: bool) -> Tensor | None:
    base = None
    for p in params:
        t = p.grad if use_grad else p.data
        if t is None:
            return None
        b = t._base
        if b is None or b.dim() != 1 or b.dtype != torch.float32:
            return None
        if base is None:
            base = b
        elif b.data_ptr() != base.data_ptr():
            return None
    return base


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas: tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2) -> None:
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def _flat(self, group) -> tuple[Tensor, Tensor] | None:
        ps = [p for p in group["params"] if p.requires_grad]
        pb, gb = _arena_of(ps, False), _arena_of(ps, True)
        if pb is None or gb is None or pb.numel() != gb.numel():
            return None
        if any(p.data.storage_offset() != p.grad.storage_offset() for p in ps):
            return None
        if sum(p.numel() for p in ps) < 0.9 * pb.numel():  # the group must own (almost) the whole arena
            return None
        return pb, gb

    def zero_grad(self, set_to_none: bool = True) -> None:
        for group in self.param_groups:
            flat = self._flat(group) if all(p.grad is not None for p in group["params"]) else None
            if flat is not None:
                flat[1].zero_()  # one memset; .grad views stay attached
            else:
                for p in group["params"]:
                    if p.grad is not None:
                        if set_to_none:
                            p.grad = None
                        else:
                            p.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            flat = self._flat(group)
            if flat is not None:
                pb, gb = flat
                st = self.state.setdefault("arena%d" % pb.data_ptr(), {})
                if not st:
                    st["step"], st["m"], st["v"] = 0, torch.zeros_like(pb), torch.zeros_like(pb)
                st["step"] += 1
                ops.adamw_step(pb, gb, st["m"], st["v"], group["lr"], b1, b2, group["eps"], group["weight_decay"], st["step"])
                pb[:0].zero_()  # bumps the arena's version counter (raw-pointer writes do not): shadows get refreshed
            else:
                for p in group["params"]:
                    if p.grad is None:
                        continue
                    st = self.state[p]
                    if not st:
                        st["step"], st["m"], st["v"] = 0, torch.zeros_like(p.data), torch.zeros_like(p.data)
                    st["step"] += 1
                    ops.adamw_step(p.data, p.grad, st["m"], st["v"], group["lr"], b1, b2, group["eps"],
                                   group["weight_decay"], st["step"])
                    p.data[:0].zero_() if p.data.dim() else None
        return loss
