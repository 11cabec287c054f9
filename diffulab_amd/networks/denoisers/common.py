"""Denoiser plugin interface (mirrors networks/denoisers/common.py:8-46 of the reference)."""

from __future__ import annotations

import copy
import logging
from abc import ABC, abstractmethod
from typing import Any, TypedDict

import torch
import torch.nn as nn
from torch import Tensor
from ... import tuning

try:  # python >= 3.11
    from typing import NotRequired, Required
except ImportError:  # python 3.10 (this image)
    from typing_extensions import NotRequired, Required


class ModelInput(TypedDict, total=False):
    x: Required[Tensor]
    p: NotRequired[float]            # label-drop probability (classifier-free guidance)
    y: NotRequired[Tensor]           # class labels
    initial_context: NotRequired[Any]
    x_context: NotRequired[Tensor]   # concatenated to x along channels


class ModelOutput(TypedDict, total=False):
    x: Required[Tensor]
    features: NotRequired[list[Tensor]]
    repa_features: NotRequired[list[Tensor]]


class Denoiser(nn.Module, ABC):
    classifier_free: bool

    def __init__(self) -> None:
        super().__init__()

    @abstractmethod
    def forward(self, x: Tensor, timesteps: Tensor, *args: Any, **kwargs: Any) -> ModelOutput: ...


class EngineFn(torch.autograd.Function):
    """autograd seam shared by the HIP denoisers: forward/backward of the whole network are the engine's launch sequences;
    parameter gradients are accumulated straight into the flat gradient arena (``p.grad`` are views of it)."""

    @staticmethod
    def forward(ctx, module: "FlatArenaDenoiser", x: Tensor, t: Tensor, y_eff: Tensor | None, anchor: Tensor, taps: tuple = ()):
        """taps: indices of blocks whose output (the residual stream after the block, bf16 [B, N, D]) is returned as extra
        differentiable outputs -- what a forward hook on ``denoiser.layers[i]`` sees in the reference (RePA, repa.py:133-134)"""
        ctx.module, ctx.taps = module, tuple(taps)
        ctx.set_materialize_grads(False)
        ctx.serial = module._next_serial()
        pred = module._engine.forward(x, t, y_eff, train=True).clone()
        ctx.pred_shape = pred.shape
        if not taps:
            return pred
        return (pred, *(module._engine.feature(k) for k in taps))

    @staticmethod
    def backward(ctx, dpred: Tensor | None, *dfeats):
        m = ctx.module
        grads = {k: g for k, g in zip(ctx.taps, dfeats) if g is not None}
        if dpred is not None or grads:
            if ctx.serial != m.__dict__.get("_fwd_serial"):
                # the engine keeps ONE set of saved activations (the last train-mode forward): a backward through an older
                # forward of the same module would silently differentiate the newer one's activations
                raise RuntimeError(f"{type(m).__name__}: backward through a forward that is no longer the module's latest "
                                   "train-mode forward (two grad-enabled forwards before one backward are not supported: "
                                   "call backward after each forward, or concatenate the batches)")
            m._prepare_grads()
            if dpred is None:
                dpred = torch.zeros(ctx.pred_shape, device=m._engine.dev)
            m._engine.backward(dpred.contiguous().float(), grads)
        return None, None, None, None, None, None


class FlatArenaDenoiser(Denoiser):
    """Denoiser whose ``nn.Module`` tree only OWNS parameters: every parameter (and gradient) is a view into one flat f32 HBM
    arena driven by an engine object (``layout.entries`` / ``layout.view`` / ``bind`` / ``forward`` / ``backward``).
    Subclasses implement ``_make_engine(device)``."""

    def __init__(self) -> None:
        super().__init__()
        for k in ("_engine", "_flat", "_flat_grad", "_anchor"):
            object.__setattr__(self, k, None)

    def _make_engine(self, device: torch.device):  # pragma: no cover - abstract
        raise NotImplementedError

    # ---- precision regime.  "bf16": bf16 MFMA operands and activations with f32 accumulation / statistics (the reference's
    #      precision_type="bf16"); "fp32": f32 activations and exact-f32 MFMA products on the f32 parameters (the reference's default
    #      precision_type="no", trainers/common.py:76,105).  Denoisers that have an fp32 engine list it in `precisions`.
    precisions: tuple[str, ...] = ("bf16",)

    @property
    def precision(self) -> str:
        return self.__dict__.get("_precision", "bf16")

    def set_precision(self, precision: str) -> "FlatArenaDenoiser":
        if precision not in ("bf16", "fp32"):
            raise ValueError(f"precision must be 'bf16' or 'fp32' (got {precision!r})")
        if precision not in self.precisions:
            raise NotImplementedError(f"diffulab_amd.{type(self).__name__} has no {precision} launch sequence (built: "
                                      f"{', '.join(self.precisions)}); the fp32-class regime exists for the class-conditional forms (MMDiT / SprintDiT simple_dit=True, DDT simple_ddt=True) and UNetModel")
        if precision != self.precision:
            object.__setattr__(self, "_precision", precision)
            object.__setattr__(self, "_carried_reducer", getattr(self._engine, "reducer", None))  # (flatten_parameters re-attaches it)
            object.__setattr__(self, "_engine", None)  # the next forward re-flattens onto the other engine (same arena layout)
            object.__setattr__(self, "_graphs", None)
        return self

    def __deepcopy__(self, memo):  # EMA wrappers deep-copy the module: copy parameters, not the engine/workspace
        saved = {k: self.__dict__.get(k) for k in ("_engine", "_flat", "_flat_grad", "_anchor", "_graphs", "_plist", "_owners")}
        for k in saved:
            object.__setattr__(self, k, None)
        try:
            cls = self.__class__
            new = cls.__new__(cls)
            memo[id(self)] = new
            for k, v in self.__dict__.items():
                new.__dict__[k] = copy.deepcopy(v, memo)
        finally:
            for k, v in saved.items():
                object.__setattr__(self, k, v)
        return new

    def _named(self) -> dict[str, nn.Parameter]:
        return dict(self.named_parameters())

    def _next_serial(self) -> int:
        """stamps a train-mode forward (the engine's saved activations belong to the latest one only)"""
        n = self.__dict__.get("_fwd_serial", 0) + 1
        object.__setattr__(self, "_fwd_serial", n)
        return n

    def _param_version(self) -> int:
        """sum of the parameters' version counters: in-place writes through a parameter (``load_state_dict``, stock
        optimizers, ``p.copy_``) are invisible to the arena's own counter the bf16 weight shadows used to be keyed on.
        The parameter list is cached between re-flattenings: a parameter object that is replaced or added (``load_state_dict(
        assign=True)``, a re-registered ``nn.Parameter``) is no longer a view of the arena, so ``engine`` re-flattens before the next
        forward (``_is_flat``) and ``flatten_parameters`` drops this cache."""
        plist = self.__dict__.get("_plist")
        if plist is None:
            plist = list(self.parameters())
            object.__setattr__(self, "_plist", plist)
        return sum(p._version for p in plist)

    def _param_owners(self) -> list:
        """[(owning module's _parameters dict, key, parameter, byte offset in the arena, name)], cached between re-flattenings: the
        per-step checks below walk this list (a dict lookup and a pointer compare per parameter) instead of the module tree --
        `named_parameters()` over the UNet's ~3000 modules cost 1.3 ms per call, four calls per training step.  A parameter object
        that is REPLACED in its module is seen (`is not`), which re-flattens and rebuilds the list."""
        owners = self.__dict__.get("_owners")
        if owners is None:
            lay = self._engine.layout
            owners = []
            for mname, mod in self.named_modules():
                for key, q in mod._parameters.items():
                    name = f"{mname}.{key}" if mname else key
                    if q is not None and name in lay.entries:  # (a tied parameter's second name is not in the layout: named_parameters() dedups)
                        owners.append((mod._parameters, key, q, 4 * lay.entries[name][0], name))
            object.__setattr__(self, "_owners", owners)
        return owners

    def _is_flat(self) -> bool:
        if self._flat is None or self._engine is None:
            return False
        base = self._flat.data_ptr()
        for params, key, q, off, _ in self._param_owners():
            if params.get(key) is not q or q.data_ptr() != base + off:
                object.__setattr__(self, "_owners", None)
                return False
        return True

    def flatten_parameters(self, device: torch.device | str | None = None) -> None:
        """(re)pack every parameter into the flat f32 arena on ``device`` and point ``.data`` / ``.grad`` at views."""
        named = self._named()
        dev = torch.device(device) if device is not None else next(iter(named.values())).device
        if dev.type != "cuda":
            raise RuntimeError(f"diffulab_amd.{type(self).__name__} runs on an MI355X only: move the module to 'cuda' "
                               "(no CPU fallback)")
        # a gradient reducer attached by the trainer's prepare() survives a re-flattening / an engine of the other precision: it
        # is handed to the new engine and re-pointed at the new gradient arena below (dropping it would silently stop the
        # data-parallel gradient exchange)
        reducer = getattr(self._engine, "reducer", None) or self.__dict__.get("_carried_reducer")
        if self._engine is None or self._engine.dev != dev or getattr(self._engine, "precision", "bf16") != self.precision:
            object.__setattr__(self, "_engine", self._make_engine(dev))
        lay = self._engine.layout
        assert set(named) == set(lay.entries), set(named) ^ set(lay.entries)
        # the arena must be ordinary (version-tracked) tensors even when the first forward happens inside
        # torch.inference_mode() (Flow.denoise is decorated with it)
        with torch.inference_mode(False), torch.no_grad():
            flat = torch.zeros(lay.size, device=dev, dtype=torch.float32)
            grad = torch.zeros(lay.size, device=dev, dtype=torch.float32)
            for name, p in named.items():
                v = lay.view(flat, name)
                v.copy_(p.detach().to(device=dev, dtype=torch.float32))
                if p.grad is not None:
                    lay.view(grad, name).copy_(p.grad.to(device=dev, dtype=torch.float32))
                p.data = v
                p.grad = lay.view(grad, name)
            anchor = torch.zeros(1, device=dev, requires_grad=True)
        object.__setattr__(self, "_flat", flat)
        object.__setattr__(self, "_flat_grad", grad)
        object.__setattr__(self, "_anchor", anchor)
        object.__setattr__(self, "_plist", None)  # (the parameter objects may be new ones: _param_version rebuilds its list)
        object.__setattr__(self, "_owners", None)
        self._engine.bind(flat, grad)
        if reducer is not None:
            reducer.rebind(grad)
            self._engine.reducer = reducer
            object.__setattr__(self, "_carried_reducer", None)

    def _prepare_grads(self) -> None:
        """called at the start of every backward: honour optimizer.zero_grad(set_to_none=True) (torch default) by
        zeroing the arena once and re-attaching the .grad views."""
        lay, grad = self._engine.layout, self._flat_grad
        first = next(iter(self.parameters()))
        if first.grad is None:
            grad.zero_()
        base = grad.data_ptr()
        for _, _, p, off, name in self._param_owners():
            g = p.grad
            if g is None or g.data_ptr() != base + off:
                p.grad = lay.view(grad, name)

    def zero_grad(self, set_to_none: bool = False) -> None:  # one memset instead of one kernel per tensor
        if self._flat_grad is not None and self._is_flat():
            self._flat_grad.zero_()
            self._prepare_grads()
        else:
            super().zero_grad(set_to_none=set_to_none)

    @property
    def engine(self):
        if not self._is_flat():
            self.flatten_parameters()
        return self._engine

    # ---- classifier-free guidance: the conditional and the label-dropped forward of a sampler step as ONE forward
    cfg_pair_capable = False  # True where `p` reaches nothing but the label drop (class-conditional MMDiT / DDT / UNetModel)

    def _effective_labels(self, y_eff: Tensor, p: float) -> Tensor:
        """LabelEmbed.drop_labels (nn.py:149): a torch device draw, the same one as the reference's; inside forward_cfg_pair the
        second half of the rows is dropped instead (what p = 1 does to every row)"""
        forced = self.__dict__.get("_forced_drop")
        if forced is not None:
            return torch.where(forced, self.n_classes, y_eff)
        if p > 0:
            return torch.where(torch.rand(y_eff.size(), device=y_eff.device) < p, self.n_classes, y_eff)
        return y_eff

    def forward_cfg_pair(self, timesteps: Tensor, **inputs: Any) -> tuple[Tensor, Tensor] | None:
        """(prediction with the labels, prediction with every label dropped) -- the two forwards a guided sampler step makes
        (flow.py:256-259, gaussian_diffusion.py one_step_denoise) -- from ONE forward over [x ; x] with the label rows
        [y ; dropped]: every kernel of the path is per row / per sample, so the two halves are the values of the two separate
        forwards, while the weights stream once and every launch has twice the rows.  The p = 1 forward's ``torch.rand(B)`` is still
        drawn (and discarded) so the device RNG stream seen by a stochastic sampler step is the reference's.  None: not applicable
        to this denoiser / these inputs (the caller then makes the two forwards)."""
        y = inputs.get("y")
        if (not self.cfg_pair_capable or not tuning.on("DL_CFG_PAIR") or y is None or inputs.get("initial_context") is not None
                or getattr(self, "label_embed", None) is None or not getattr(self, "classifier_free", False)
                or torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters())
                or any(getattr(layer, "_forward_hooks", None) for layer in getattr(self, "layers", ()))):
            return None
        x = inputs["x"]
        B = x.shape[0]
        dev = self.engine.dev
        torch.rand(y.size(), device=dev)  # the draw of the p = 1 forward (every row is dropped whatever it returns)
        two = {k: (torch.cat([v, v], dim=0) if isinstance(v, Tensor) and v.dim() > 0 and v.shape[0] == B else v)
               for k, v in inputs.items() if k != "p"}
        forced = torch.zeros(2 * B, dtype=torch.bool, device=dev)
        forced[B:] = True
        object.__setattr__(self, "_forced_drop", forced)
        try:
            pred = self(**two, timesteps=torch.cat([timesteps, timesteps], dim=0), p=0)["x"]
        finally:
            object.__setattr__(self, "_forced_drop", None)
        return pred[:B], pred[B:]

    def _run(self, x: Tensor, t: Tensor, y_eff: Tensor | None, taps: tuple = ()):
        """prediction (and, with taps, the tapped block outputs)"""
        eng = self.engine
        eng.param_version = self._param_version()
        need_grad = torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters())
        if need_grad:
            return EngineFn.apply(self, x, t, y_eff, self._anchor, tuple(taps))
        if taps:  # validation with an auxiliary loss on intermediate features: run the keep-everything sequence eagerly
            self._next_serial()  # (this overwrites the saved activations of an earlier grad-enabled forward)
            pred = eng.forward(x, t, y_eff, train=True).clone()
            return (pred, *(eng.feature(k).clone() for k in taps))
        return self._infer(eng, x, t, y_eff)

    # hooks of the hipGraph replay for denoisers whose launch sequence reads further per-call state held on the engine
    def _graph_key(self, eng) -> tuple | None:
        """hashable part of the capture key beyond the input shapes; None: this call cannot be replayed (run it eagerly)"""
        return ()

    def _graph_inputs(self, eng) -> tuple:
        """further per-call input tensors the launch sequence reads through the engine (entries may be None)"""
        return ()

    def _graph_set_inputs(self, eng, tensors: tuple) -> None:
        """point the engine at `tensors` (the static copies a captured graph reads)"""

    def _infer(self, eng, x: Tensor, t: Tensor, y_eff: Tensor | None) -> Tensor:
        """inference forward: the engine's launch sequence is static per input shape, so it is captured once into a hipGraph
        and replayed (sampler loops at small batch are launch-bound: ~110 launches per DiT-S forward).  DL_HIPGRAPH=0
        disables the capture."""
        extra = self._graph_key(eng)
        if not tuning.on("DL_HIPGRAPH") or extra is None:
            return eng.forward(x, t, y_eff, train=False).clone()
        graphs = self.__dict__.get("_graphs")
        if graphs is None:
            graphs = {}
            object.__setattr__(self, "_graphs", graphs)
        ins = self._graph_inputs(eng)
        key = (id(eng), tuple(x.shape), y_eff is not None, extra,
               tuple(None if i is None else (tuple(i.shape), i.dtype) for i in ins))
        ent = graphs.get(key)
        if ent is None:
            out = eng.forward(x, t, y_eff, train=False).clone()  # eager: allocates workspaces / tables, refreshes the shadows
            if len(graphs) >= 8:
                graphs.clear()
            with torch.inference_mode(False), torch.no_grad():
                xs, ts = x.detach().clone(), t.detach().clone()
                ys = y_eff.detach().clone() if y_eff is not None else None
                statics = tuple(None if i is None else i.detach().clone() for i in ins)
                self._graph_set_inputs(eng, statics)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(g):
                        out_s = eng.forward(xs, ts, ys, train=False, refresh=False)
                    # the entry holds the workspace it was captured on: the engine's per-shape cache may evict that workspace
                    # later (train shape, then many inference shapes), and a replay must never touch freed memory
                    graphs[key] = (g, xs, ts, ys, out_s, statics, getattr(eng, "ws", None))
                except Exception as e:  # capture refused: stay eager for this shape (and say so once)
                    logging.warning("hipGraph capture of the inference forward failed (%s): running eagerly", e)
                    graphs[key] = False
            return out
        if ent is False:
            return eng.forward(x, t, y_eff, train=False).clone()
        g, xs, ts, ys, out_s, statics, _ws_alive = ent
        eng.refresh_shadows()
        xs.copy_(x)
        ts.copy_(t)
        if ys is not None:
            ys.copy_(y_eff)
        for st, cur in zip(statics, ins):
            if st is not None:
                st.copy_(cur)
        self._graph_set_inputs(eng, statics)
        g.replay()
        return out_s.clone()
