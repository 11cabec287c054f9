"""Data-parallel gradient reduction for the flat gradient arena (one process per GPU, RCCL over xGMI).

Replaces the DDP wrap that `accelerator.prepare` installs in the reference (base_trainer.py:277-279):
instead of ~7 autograd-hooked 25 MB buckets of scattered parameter tensors, the engine's backward calls
`ready(lo, hi)` as soon as a contiguous range of the gradient arena is final (blocks finish in reverse order), and
each range is summed across ranks with ONE large in-place all-reduce issued on a side stream, overlapping the
remaining backward.  xGMI is point-to-point (per-link bound), so few large messages beat many small ones; the
1/world averaging is folded into the optimizer kernel (`FusedAdamW.grad_scale`), not a separate pass.
Semantics kept from the reference: gradients are averaged over ranks; with gradient accumulation only the sync
micro-step reduces (`no_sync` otherwise).
"""

from __future__ import annotations

import os
import time

import torch
import torch.distributed as dist
from torch import Tensor


def configure_rccl_env() -> None:
    """call BEFORE the process group creates its RCCL communicator (bench.py, Trainer.__init__): cap RCCL's channels -- one resident
    workgroup each -- at the number of compute units the engines leave free for them (DL_DP_RESERVE_CUS, engine._main_wgs).  The
    exchange moves ~160 MB per 20 ms step (~14 GB/s per GPU with reduce-scatter + all-gather at 8 ranks): a few channels carry that;
    what matters is that their workgroups never take a CU a one-workgroup-per-CU kernel was sized for.
    An explicit NCCL_MAX_NCHANNELS in the environment wins."""
    from .. import tuning

    channels = tuning.integer("DL_DP_RESERVE_CUS", 0)
    if channels > 0:
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(channels))


class GradReducer:
    def __init__(self, flat_grad: Tensor, bucket_bytes: int = 48 << 20, group=None, backend: str | None = None, comm=None) -> None:
        """backend "torch" (default): torch.distributed collectives (backend nccl = RCCL);  "abi" (or DIFFULAB_DP_BACKEND=abi):
        the same exchange through libdiffulab_comm.so (include/diffulab_comm.h: dl_reduce_scatter_allgather_async on the
        library's own communicator and comm stream), for hosts that drive the C ABI without torch.distributed"""
        self.flat = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.backend = backend or os.environ.get("DIFFULAB_DP_BACKEND", "torch")
        self.comm = comm
        if self.backend == "abi" and self.comm is None and flat_grad.is_cuda and (self.world > 1 or comm is not None):
            from .._comm import Communicator

            self.comm = Communicator(dist.get_rank(group), self.world, flat_grad.device.index)
        # DIFFULAB_DP_BUCKET_MB overrides the bucket size; a value larger than the arena means ONE all-reduce in finish(), after
        # the backward has ended (nothing overlaps).  Why one might want that: every heavy kernel of the step is one workgroup per
        # CU with all of the CU's registers, so RCCL's workgroups cannot co-reside with them -- they
        # take whole CUs at kernel boundaries, and a persistent launch sized for 256 CUs that finds fewer free runs a second round.
        # Unmeasured (no multi-GPU box was available to the builder): the knob makes the A/B a one-line experiment.
        mb = os.environ.get("DIFFULAB_DP_BUCKET_MB")
        if mb:
            bucket_bytes = int(float(mb) * (1 << 20))
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.enabled = self.world > 1
        self.sync = True  # False inside a gradient-accumulation micro-step (no_sync)
        self.comm_stream = torch.cuda.Stream() if (self.enabled and flat_grad.is_cuda) else None
        self._pending: list[tuple[int, int]] = []
        self._extra: list = []
        self._works = []
        self.measure = False  # bench: record how long the compute stream waits in finish() = the exposed part of the reduction
        self.tail_events: list[tuple] = []
        # Overlap or not is decided by measurement (DIFFULAB_DP_OVERLAP=auto, the default): every heavy kernel of the step is one
        # workgroup per CU with the CU's whole register file, so a resident communication workgroup makes the kernels beside it
        # run a second round (foreign workgroups on 8 CUs for half of the step cost 22 %), while the single
        # exchange after the backward costs its transfer time.  Which is cheaper depends on RCCL's residency on the node at hand:
        # synchronising steps 2-7 run overlapped, 8-13 with one exchange after the backward (median of the backward-to-finish
        # interval per mode, MAX over ranks); the overlapped default is only left when the single exchange wins by > 3 %.  "1" / "0" pin a mode.
        # A pinned mode is a pin from the first step on: a user who sets "0" because overlapped collectives misbehave on their node
        # must never run the overlapped schedule.  DIFFULAB_DP_MEASURE=1 opts a pinned run into timing BOTH schedules
        # during the warm-up steps anyway (both timings land in `tuned`, the pin still decides afterwards).
        mode = os.environ.get("DIFFULAB_DP_OVERLAP", "auto")
        self.overlap = mode != "0"
        self._pinned: bool | None = None if mode == "auto" else (mode != "0")
        self.tuned: dict | None = None
        measure = mode == "auto" or os.environ.get("DIFFULAB_DP_MEASURE", "0") == "1"
        self._tune = {"step": 0, "marks": []} if (measure and self.enabled) else None
        if self._tune is not None:
            self.overlap = True  # the measurement starts with the overlapped schedule

    def rebind(self, flat_grad: Tensor) -> None:
        """point the reducer at a NEW gradient arena of the same layout (the denoiser re-flattened its parameters or switched
        its precision regime after prepare()): nothing may be in flight"""
        if self._pending or self._works:
            raise RuntimeError("GradReducer.rebind: a gradient exchange is in flight (finish() the step first)")
        if flat_grad.numel() != self.flat.numel():
            raise RuntimeError(f"GradReducer.rebind: arena size changed ({self.flat.numel()} -> {flat_grad.numel()})")
        self.flat = flat_grad

    # -- called by the engine's backward, ranges arrive high-to-low as blocks finish
    def ready(self, lo: int, hi: int, extra_events=(), flush: bool = False) -> None:
        """[lo, hi) of the arena is final once the current stream AND `extra_events` (side-stream producers) are reached.
        flush: reduce what is pending now even if the bucket is not full (the engine asks for it on the last blocks of the
        backward, so that what is left for finish() -- the exposed part of the exchange -- is small)."""
        if not (self.enabled and self.sync):
            return
        self._tune_begin()
        if hi > lo:
            self._pending.append((lo, hi))
        self._extra.extend(extra_events)
        if self.overlap and (flush or sum(h - l for l, h in self._pending) >= self.bucket_elems):
            self._flush()

    def _flush(self) -> None:
        """one all-reduce per contiguous run of the pending ranges (a block's own parameters and its slice of the stacked adaLN
        matrix sit in different parts of the arena)"""
        if not self._pending:
            return
        runs: list[list[int]] = []
        for lo, hi in sorted(self._pending):
            if runs and lo <= runs[-1][1]:
                assert lo == runs[-1][1], "gradient ranges handed to the reducer must not overlap"
                runs[-1][1] = hi
            else:
                runs.append([lo, hi])
        self._pending.clear()
        for i, (lo, hi) in enumerate(runs):
            self._reduce(lo, hi, first=(i == 0))

    def _reduce(self, lo: int, hi: int, first: bool) -> None:
        chunk = self.flat[lo:hi]
        if self.comm is not None:  # C-ABI path: the library owns the comm stream and the ordering events
            if first:
                for e in self._extra:
                    self.comm.after_event(e.cuda_event)
                self._extra.clear()
            self.comm.all_reduce_async(chunk.data_ptr(), chunk.numel(), torch.cuda.current_stream().cuda_stream)
            return
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record()  # everything that produced this range is on the compute stream before this point
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                if first:  # (the later runs of this flush follow on the same comm stream)
                    for e in self._extra:
                        self.comm_stream.wait_event(e)
                    self._extra.clear()
                self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:  # CPU / gloo (tests)
            self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self) -> None:
        """reduce what is left and make the compute stream wait for every collective."""
        if not (self.enabled and self.sync):
            self._pending.clear()
            return
        self._flush()
        if self.comm is not None:
            self.comm.wait(torch.cuda.current_stream().cuda_stream)
            self._tune_tick()
            return
        e0 = None
        if self.measure and self.comm_stream is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in self._works:
            w.wait()
        self._works.clear()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.tail_events.append((e0, e1))
        self._tune_tick()

    TUNE_SKIP, TUNE_STEPS, TUNE_MARGIN = 2, 6, 1.03

    @property
    def tuning(self) -> bool:
        """True while the schedule decision is still being measured (the first TUNE_SKIP + 2 TUNE_STEPS synchronising steps)"""
        return self._tune is not None

    def _tune_begin(self) -> None:
        """first ready() of a synchronising step: the start mark of the interval the decision is taken on -- from the first final
        gradient range to the end of finish(), i.e. the part of the step the exchange can influence.  Dataloader stalls, loss
        read-backs, EMA and logging between two backward passes are outside of it."""
        st = self._tune
        if st is None or st.get("open") is not None:
            return
        if self.flat.is_cuda:
            mark = torch.cuda.Event(enable_timing=True)
            mark.record()
        else:
            mark = time.perf_counter()
        st["open"] = mark

    def _tune_tick(self) -> None:
        """end mark of a synchronising step (stream events: no host sync until the decision); see __init__"""
        st = self._tune
        if st is None:
            return
        start = st.pop("open", None)
        if self.flat.is_cuda:
            mark = torch.cuda.Event(enable_timing=True)
            mark.record()
        else:
            mark = time.perf_counter()
        if start is not None:
            st["marks"].append((start, mark))
        st["step"] += 1
        a, b = self.TUNE_SKIP + self.TUNE_STEPS, self.TUNE_SKIP + 2 * self.TUNE_STEPS
        if st["step"] == a:  # steps [SKIP, a) ran overlapped
            self.overlap = False
        elif st["step"] == b:
            m = st["marks"]

            def median_ms(i: int, j: int) -> float:
                if self.flat.is_cuda:
                    m[j - 1][1].synchronize()
                    v = sorted(x.elapsed_time(y) for x, y in m[i:j])
                else:
                    v = sorted((y - x) * 1e3 for x, y in m[i:j])
                return v[len(v) // 2] if v else 0.0

            t = torch.tensor([median_ms(self.TUNE_SKIP, a), median_ms(a, b)], dtype=torch.float64,
                             device=self.flat.device if self.flat.is_cuda else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)  # every rank takes the same decision
            t_overlap, t_after = float(t[0]), float(t[1])
            # the overlapped schedule is the default and is only left for a clear win of the single exchange; a pinned mode wins
            # over the measurement (which is reported all the same)
            measured = t_overlap <= self.TUNE_MARGIN * t_after
            self.overlap = measured if self._pinned is None else self._pinned
            self.tuned = {"mode": "overlapped" if self.overlap else "after_backward", "overlapped_ms_per_step": round(t_overlap, 3),
                          "after_backward_ms_per_step": round(t_after, 3),
                          "decided": "measured" if self._pinned is None else "DIFFULAB_DP_OVERLAP pinned (measurement would pick %s)"
                          % ("overlapped" if measured else "after_backward"),
                          "interval": "first final gradient range -> end of finish(), median of %d steps per mode" % self.TUNE_STEPS}
            if (dist.get_rank(self.group) if dist.is_initialized() else 0) == 0:
                import logging

                logging.getLogger("diffulab_amd.dp").info("gradient exchange schedule: %s", self.tuned)
            self._tune = None

    def exposed_ms(self) -> float | None:
        """mean time the compute stream spent waiting for the collectives in finish() over the measured steps"""
        if not self.tail_events:
            return None
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self.tail_events) / len(self.tail_events)
        self.tail_events.clear()
        return ms

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world


def broadcast_arena(flat_params: Tensor, src: int = 0, group=None) -> None:
    """DDP constructor semantics: every rank starts from rank 0's parameters (one 160 MB broadcast for DiT-S/2)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)
