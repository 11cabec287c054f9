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
from ..engine import bump_param_epoch


def _arena_of(params: list[Tensor], use_grad: bool) -> Tensor | None:
    """flat f32 tensor spanning the storage all `params` (or their grads) are views of, else None.
    (`p.data` is not a view object, so `_base` cannot be used: compare the underlying storages.)"""
    store = None
    for p in params:
        t = p.grad if use_grad else p.data
        if t is None or t.dtype != torch.float32 or not t.is_contiguous():
            return None
        s = t.untyped_storage()
        if store is None:
            store = s
        elif s.data_ptr() != store.data_ptr():
            return None
    if store is None:
        return None
    t0 = params[0].grad if use_grad else params[0].data
    return torch.empty(0, dtype=torch.float32, device=t0.device).set_(store, 0, (store.nbytes() // 4,))


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas: tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2) -> None:
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = 1.0  # data parallel: 1/world (gradients are SUM-reduced), folded into the update kernel
        self._legacy_arena_states: list[dict] = []
        self._graph_hyper: list[tuple[Tensor, Tensor]] | None = None  # per group: (device f32[8], pinned host mirror)
        self._graph_buffers: list[tuple[Tensor, Tensor]] | None = None

    # ---- hipGraph support (dl_adamw_step_dev; the captured-step experiment lives in scripts/lab/graph_step.py): inside a captured
    #      step the update kernels read their scalars from a device buffer; the host advances the step count and refreshes that
    #      buffer BEFORE each replay.  The buffers are allocated ONCE: graphs captured earlier hold their addresses, so a second
    #      begin_graph_mode (another input shape being captured) must not replace them.
    def begin_graph_mode(self) -> None:
        if self._graph_hyper is None:
            if self._graph_buffers is None:
                dev = next(p for g in self.param_groups for p in g["params"]).device
                self._graph_buffers = [(torch.zeros(8, device=dev), torch.zeros(8).pin_memory()) for _ in self.param_groups]
            self._graph_hyper = self._graph_buffers
        self._graph_step = max([int(st["step"]) for st in self.state.values() if "step" in st] or [0])

    def in_graph_mode(self) -> bool:
        return self._graph_hyper is not None

    def end_graph_mode(self) -> None:
        self._graph_hyper = None  # (the buffers stay allocated: a graph that is still alive keeps reading them)

    def advance(self) -> None:
        """one optimizer step is about to be replayed: bump the step count everywhere and upload the hyper-parameters"""
        assert self._graph_hyper is not None
        self._graph_step += 1
        for st in self.state.values():
            if "step" in st:
                st["step"] = self._graph_step
        for group, (dev, host) in zip(self.param_groups, self._graph_hyper):
            b1, b2 = group["betas"]
            bc1, bc2 = 1.0 - b1**self._graph_step, 1.0 - b2**self._graph_step
            host.copy_(torch.tensor([group["lr"], b1, b2, group["eps"], group["weight_decay"], group["lr"] / bc1, bc2**-0.5,
                                     self.grad_scale], dtype=torch.float32))
            dev.copy_(host, non_blocking=True)
        bump_param_epoch()

    @staticmethod
    def _arena_key(group, rest: list) -> Tensor:
        skip = {id(p) for p in rest}
        return next(p for p in group["params"] if id(p) not in skip and p.requires_grad)

    def load_state_dict(self, state_dict) -> None:
        """accepts three optimizer.pt layouts: this class's own (arena-sized ``m`` / ``v`` / ``step`` under the first arena
        parameter of a group, per-parameter ``m`` / ``v`` / ``step`` for tensors outside the arena), the reference's
        ``torch.optim.AdamW`` checkpoint (per-parameter ``exp_avg`` / ``exp_avg_sq`` / ``step``: packed into the arena moments at
        the first step, see _adopt_reference_state) and the address-keyed ``arena*`` entries round 1 wrote (matched by size)"""
        super().load_state_dict(state_dict)
        for k in [k for k in self.state if isinstance(k, str) and k.startswith("arena")]:
            self._legacy_arena_states.append(self.state.pop(k))

    def _adopt_reference_state(self, pb: Tensor, inside: list, key: Tensor) -> None:
        """torch.optim.AdamW state (the reference's optimizer.pt, base_trainer.py:246-251) -> arena moments: every arena
        parameter's exp_avg / exp_avg_sq is copied to its offset, the step count is the parameters' common one"""
        m, v = torch.zeros_like(pb), torch.zeros_like(pb)
        steps = set()
        base = pb.storage_offset()
        for p in inside:
            st = self.state.get(p)
            if not st:
                continue
            if "exp_avg" not in st or "exp_avg_sq" not in st:
                raise RuntimeError(f"optimizer state of an arena parameter has keys {sorted(st)}: expected FusedAdamW's "
                                   "('m', 'v', 'step') or torch.optim.AdamW's ('exp_avg', 'exp_avg_sq', 'step')")
            o = p.data.storage_offset() - base
            m[o : o + p.numel()].copy_(st["exp_avg"].reshape(-1).to(pb.device, torch.float32))
            v[o : o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1).to(pb.device, torch.float32))
            steps.add(int(st["step"]))
            if p is not key:
                del self.state[p]
        if len(steps) > 1:
            raise RuntimeError(f"torch.optim.AdamW checkpoint with different step counts per parameter ({sorted(steps)}): the arena "
                               "update has one")
        st = self.state[key]
        st.clear()
        st["step"], st["m"], st["v"] = (steps.pop() if steps else 0), m, v

    @staticmethod
    def _adopt_reference_param_state(st: dict, p: Tensor) -> None:
        """a parameter OUTSIDE the arena (REPA projector / resampler tensors that share the denoiser's param group,
        examples/train_repa.py) resumed from a torch.optim.AdamW checkpoint: rename exp_avg / exp_avg_sq to this class's m / v"""
        if "m" not in st and "exp_avg" in st and "exp_avg_sq" in st:
            st["m"] = st.pop("exp_avg").to(device=p.device, dtype=torch.float32).contiguous()
            st["v"] = st.pop("exp_avg_sq").to(device=p.device, dtype=torch.float32).contiguous()
            st.pop("max_exp_avg_sq", None)
            st["step"] = int(st.get("step", 0))

    def _flat(self, group) -> tuple[Tensor, Tensor, list] | None:
        """`_flat_scan(group)`, remembered per group: a hit is re-validated by pointer compares only (every arena parameter's data and
        gradient still sit at the recorded addresses, the group's parameter list is the same objects) -- the full scan costs 1.7 ms
        on the UNet's ~500 tensors and ran twice per training step (zero_grad, step)"""
        cache = self.__dict__.setdefault("_flat_cache", {})
        hit = cache.get(id(group))
        if hit is not None:
            plist, n, pins, res = hit
            if plist is group["params"] and len(plist) == n:
                for p, dp, gp in pins:
                    g = p.grad
                    if g is None or g.data_ptr() != gp or p.data_ptr() != dp or not p.requires_grad:
                        break
                else:
                    if all(p.grad is None or not p.requires_grad or id(p) in res[3] for p in res[2]):  # (nobody outside joined the arena)
                        return res[:3]
        res = self._flat_scan(group)
        if res is None:
            cache.pop(id(group), None)
            return None
        ids = {id(p) for p in group["params"]} - {id(p) for p in res[2]}
        inside = [p for p in group["params"] if id(p) in ids]
        # (outside tensors are re-scanned when one of them gains a gradient: it might be an arena view that had none yet)
        outside_with_grad = {id(p) for p in res[2] if p.grad is not None and p.requires_grad}
        cache[id(group)] = (group["params"], len(group["params"]), [(p, p.data_ptr(), p.grad.data_ptr()) for p in inside],
                            (res[0], res[1], res[2], outside_with_grad))
        return res

    def _flat_scan(self, group) -> tuple[Tensor, Tensor, list] | None:
        """(param arena, grad arena, parameters NOT in it) when (most of) the group lives in one flat arena -- e.g. a denoiser's
        arena plus the few tensors of an auxiliary loss head (REPA projector) in the same param group -- else None"""
        ps = [p for p in group["params"] if p.requires_grad]
        by_store: dict[int, list] = {}
        for p in ps:
            if p.grad is None or p.data.dtype != torch.float32:
                continue
            by_store.setdefault(p.data.untyped_storage().data_ptr(), []).append(p)
        if not by_store:
            return None
        inside = max(by_store.values(), key=lambda v: sum(q.numel() for q in v))
        pb, gb = _arena_of(inside, False), _arena_of(inside, True)
        if pb is None or gb is None or pb.numel() != gb.numel():
            return None
        if any(p.data.storage_offset() != p.grad.storage_offset() for p in inside):
            return None
        if sum(p.numel() for p in inside) < 0.9 * pb.numel():  # the group must own (almost) the whole arena
            return None
        ids = {id(p) for p in inside}
        return pb, gb, [p for p in group["params"] if id(p) not in ids]

    def zero_grad(self, set_to_none: bool = True) -> None:
        for group in self.param_groups:
            flat = self._flat(group)
            rest = group["params"]
            if flat is not None:
                flat[1].zero_()  # one memset; .grad views stay attached
                rest = flat[2]
            for p in rest:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            if self._graph_hyper is not None:  # captured step: scalars come from the device buffer, state must already exist
                hyper = self._graph_hyper[gi][0]
                flat = self._flat(group)
                rest = group["params"]
                if flat is not None:
                    pb, gb, rest = flat
                    st = self.state[self._arena_key(group, rest)]
                    if "m" not in st:
                        raise RuntimeError("graph mode needs the arena moments: run one eager step() after loading a checkpoint")
                    ops.adamw_step_dev(pb, gb, st["m"], st["v"], hyper)
                for p in rest:
                    if p.grad is not None:
                        st = self.state[p]
                        self._adopt_reference_param_state(st, p)
                        ops.adamw_step_dev(p.data, p.grad, st["m"], st["v"], hyper)
                continue
            flat = self._flat(group)
            rest = group["params"]
            if flat is not None:
                pb, gb, rest = flat
                # The arena's moments are keyed on the group's first arena parameter: Optimizer.state_dict() maps parameter keys
                # to stable indices, so a checkpoint resumes in a new process (a key derived from the arena's address would not)
                key = self._arena_key(group, rest)
                st = self.state[key]
                if not st:
                    for i, old in enumerate(self._legacy_arena_states):  # address-keyed entries of round-1 checkpoints
                        if old["m"].numel() == pb.numel():
                            st.update(self._legacy_arena_states.pop(i))
                            break
                if st and "m" not in st:  # a torch.optim.AdamW checkpoint (per-parameter exp_avg / exp_avg_sq)
                    skip = {id(p) for p in rest}
                    self._adopt_reference_state(pb, [p for p in group["params"] if id(p) not in skip and p.requires_grad], key)
                if not st:
                    st["step"], st["m"], st["v"] = 0, torch.zeros_like(pb), torch.zeros_like(pb)
                if st["m"].device != pb.device or st["m"].numel() != pb.numel():  # state loaded before the model moved
                    assert st["m"].numel() == pb.numel(), "optimizer state does not match the parameter arena"
                    st["m"], st["v"] = st["m"].to(pb.device), st["v"].to(pb.device)
                st["step"] = int(st["step"]) + 1
                ops.adamw_step(pb, gb, st["m"], st["v"], group["lr"], b1, b2, group["eps"], group["weight_decay"], st["step"],
                               self.grad_scale)
                bump_param_epoch()  # raw-pointer writes bump no torch version counter: tell the engines
            if rest:
                for p in rest:
                    if p.grad is None:
                        continue
                    st = self.state[p]
                    self._adopt_reference_param_state(st, p)
                    if not st:
                        st["step"], st["m"], st["v"] = 0, torch.zeros_like(p.data), torch.zeros_like(p.data)
                    if st["m"].device != p.device:
                        st["m"], st["v"] = st["m"].to(p.device), st["v"].to(p.device)
                    st["step"] = int(st["step"]) + 1
                    ops.adamw_step(p.data, p.grad, st["m"], st["v"], group["lr"], b1, b2, group["eps"],
                                   group["weight_decay"], st["step"], self.grad_scale)
            bump_param_epoch()
        return loss
