"""LAB CODE (moved out of the package in round 3: measured slower than the eager launches on this ROCm, DESIGN.md section 6 --
kept for runtimes whose graph launch is cheaper; nothing under diffulab_amd/ imports it).

The whole training step as ONE hipGraph (zero_grad -> noise + loss head -> denoiser forward -> backward with its side-stream
weight-gradient GEMMs -> fused AdamW), for the launch-bound configurations.

A training step of the joint text-image SPRINT model (BASELINE config 5) is ~1400 kernel launches; at the yaml's batch size the
host needs 34.7 ms to issue them while the GPU needs far less (scripts/train_step_bench.py).  The launch sequence of an engine is
static for a given input shape, so after a few eager steps (workspaces, optimizer state) the step is captured once per shape and
replayed: the host's work per step becomes the CPU timestep draw, a few input copies into static buffers and one graph launch.

What makes the step capturable: every kernel takes its stream explicitly; the side stream is forked from and joined to the main
stream inside the step; random draws on the device (noise, label / context / token drops) go through torch's graph-safe generator;
the timesteps are drawn on the CPU as in the reference (flow.py:168-197) and copied into a static device buffer before the replay;
FusedAdamW reads its scalars from a device buffer (dl_adamw_step_dev) that the host refreshes before every replay.

Reference semantics kept: the statement order of training_step (base_trainer.py:138-151); the loss values are the static device
scalars the caller reads back (``.item()``), exactly where the reference does.  Only plain steps are captured
(gradient_accumulation_step == 1, single process); everything else runs eagerly.
"""

from __future__ import annotations

from typing import Any, Callable

import torch
from torch import Tensor


def _map(obj: Any, fn: Callable[[Tensor], Any]) -> Any:
    if isinstance(obj, Tensor):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map(v, fn) for v in obj)
    return obj


def _signature(obj: Any) -> Any:
    if isinstance(obj, Tensor):
        return (tuple(obj.shape), obj.dtype, obj.device.type)
    if isinstance(obj, dict):
        return tuple((k, _signature(v)) for k, v in sorted(obj.items()))
    if isinstance(obj, (list, tuple)):
        return tuple(_signature(v) for v in obj)
    return obj if isinstance(obj, (int, float, str, bool, type(None))) else type(obj).__name__


def _copy_into(dst: Any, src: Any) -> None:
    if isinstance(dst, Tensor):
        dst.copy_(src, non_blocking=True)
    elif isinstance(dst, dict):
        for k in dst:
            _copy_into(dst[k], src[k])
    elif isinstance(dst, (list, tuple)):
        for d, s in zip(dst, src):
            _copy_into(d, s)


class GraphedTrainStep:
    def __init__(self, diffuser, optimizer, warmup: int = 3) -> None:
        if not hasattr(optimizer, "begin_graph_mode"):
            raise TypeError("GraphedTrainStep needs diffulab_amd.training.FusedAdamW (its scalars can live on the device)")
        self.diffuser, self.optimizer, self.warmup = diffuser, optimizer, warmup
        self._seen: dict[Any, int] = {}
        self._graphs: dict[Any, tuple] = {}

    def _eager(self, model_inputs: dict, timesteps: Tensor, extra: dict) -> dict[str, Tensor]:
        self.optimizer.zero_grad()
        losses = self.diffuser.compute_loss(model_inputs=model_inputs, timesteps=timesteps, extra_args=extra)
        sum(losses.values()).backward()
        if self.optimizer.in_graph_mode():  # a graph of another shape is alive: the update reads the device scalars, refresh them
            self.optimizer.advance()
        self.optimizer.step()
        return losses

    def __call__(self, model_inputs: dict, timesteps: Tensor, extra: dict | None = None) -> dict[str, Tensor]:
        """one optimizer step on this batch; returns the loss dict (device scalars, valid until the next call)"""
        extra = extra or {}
        key = (_signature(model_inputs), _signature(timesteps), _signature(extra))
        n = self._seen.get(key, 0)
        self._seen[key] = n + 1
        if n < self.warmup:  # eager steps allocate the engine's workspace for this shape and the optimizer state
            return self._eager(model_inputs, timesteps, extra)
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._capture(key, model_inputs, timesteps, extra)  # (records the step and runs it once on this batch)
            return self._eager(model_inputs, timesteps, extra) if ent is False else ent[4]
        if ent is False:
            return self._eager(model_inputs, timesteps, extra)
        graph, s_in, s_t, s_extra, s_losses = ent
        _copy_into(s_in, model_inputs)
        s_t.copy_(timesteps, non_blocking=True)
        _copy_into(s_extra, extra)
        self.optimizer.advance()
        graph.replay()
        return s_losses

    def _capture(self, key, model_inputs: dict, timesteps: Tensor, extra: dict):
        dev = next(self.diffuser.denoiser.parameters()).device
        clone = lambda t: t.detach().to(dev).clone()  # noqa: E731
        s_in, s_t, s_extra = _map(model_inputs, clone), clone(timesteps), _map(extra, clone)
        opt = self.optimizer
        opt.begin_graph_mode()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        try:
            opt.advance()  # the captured update is the next optimizer step
            with torch.cuda.graph(graph):
                opt.zero_grad(set_to_none=False)
                # compute_loss overwrites model_inputs["x"] with z_t: give it a fresh dict over the static tensors
                losses = self.diffuser.compute_loss(model_inputs=dict(s_in), timesteps=s_t, extra_args=s_extra)
                sum(losses.values()).backward()
                opt.step()
            # (capture does not execute: the step that was just recorded has not run yet)
            _copy_into(s_in, model_inputs)
            s_t.copy_(timesteps, non_blocking=True)
            _copy_into(s_extra, extra)
            graph.replay()
            ent = (graph, s_in, s_t, s_extra, losses)
        except Exception as e:  # capture refused: stay eager for this shape, and say so
            import logging

            logging.warning("hipGraph capture of the training step failed (%s): running this shape eagerly", e)
            if not any(v not in (None, False) for v in self._graphs.values()):  # other shapes' graphs keep the device scalars
                opt.end_graph_mode()
            torch.cuda.synchronize()
            self._graphs[key] = False
            return False
        self._graphs[key] = ent
        return ent
