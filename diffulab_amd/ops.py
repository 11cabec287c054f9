"""Tensor-level wrappers over the C ABI (``include/diffulab_hip.h``).

Torch is plumbing only here: it owns the device buffers and the stream; every computation below is a
hand-written HIP kernel in ``libdiffulab_hip.so``.  All functions write into caller-provided outputs
(so the engines can run allocation-free / graph-capturable) and return the output for convenience.
There is no fallback path: a missing library or a CPU tensor raises.
"""

from __future__ import annotations

import ctypes
import os

import torch
from torch import Tensor

from ._lib import lib
from . import tuning

BF16, F32 = 0, 1
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2
LOSS_FLOW, LOSS_EPS = 0, 1
MEAN_TYPES = {"epsilon": 0, "xstart": 1, "xprev": 2}
PATCH_CPP, PATCH_PPC = 0, 1


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _s() -> int:
    """the caller's current HIP stream (raw handle)"""
    if _raw_stream is not None:  # 0.1 us instead of 2.6 us for constructing a torch Stream object per launch
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t: Tensor | None) -> int | None:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("diffulab_amd ops need device tensors (no CPU fallback exists)")
    return t.data_ptr()


_LIB = None


def _call(name: str, *args) -> None:
    global _LIB
    if _LIB is None:
        _LIB = lib()
    _LIB.call(name, *args)
    if _REC is not None:
        _REC.calls.append((_LIB._fns[name][0], args))


# ------------------------------------------------------------------ launch plans
class LaunchPlan:
    """One engine pass (a training forward, a backward) as a flat list of C calls with their final arguments, recorded while the
    pass runs once and re-issued by `replay()` afterwards: the launch sequence of an engine is static for a given input shape, and
    on the host-bound configurations (the UNet at its configured batch: 716 launches, ~24 us of Python per launch against ~5 us
    inside the HIP runtime) the step time IS the time the host needs to walk the engine's Python.  (A hipGraph of the same step
    replays SLOWER than the eager launches on this runtime: DESIGN.md section 6, round 6.)  What makes a pass recordable: every
    action is a C-ABI call with plain arguments -- kernels, `dl_memset_zero`, `dl_stream_wait_stream` -- or a registered host
    closure (`rec`); every buffer the calls name is kept alive by the plan (`keep`); per-call inputs live in static buffers the
    caller copies into before a replay."""

    __slots__ = ("calls", "held", "state", "inputs", "out", "bwd")

    def __init__(self) -> None:
        self.calls: list = []
        self.held: list = []
        self.state = self.inputs = self.out = self.bwd = None

    def replay(self) -> None:
        for fn, args in self.calls:
            if fn(*args):
                raise RuntimeError(f"launch plan: a replayed call failed: {lib().cdll.dl_last_error().decode()}")


_REC: LaunchPlan | None = None


class recording:
    """``with ops.recording(plan):`` every ops call inside is executed AND appended to the plan"""

    def __init__(self, plan: LaunchPlan) -> None:
        self.plan = plan

    def __enter__(self) -> LaunchPlan:
        global _REC
        self.prev, _REC = _REC, self.plan
        return self.plan

    def __exit__(self, *exc) -> None:
        global _REC
        _REC = self.prev


def is_recording() -> bool:
    return _REC is not None


def keep(obj):
    """the plan being recorded (if any) holds a reference to `obj` (a buffer or table whose raw address a recorded call carries)"""
    if _REC is not None:
        _REC.held.append(obj)
    return obj


def rec(fn) -> None:
    """run the host closure `fn` now; a plan being recorded re-runs it at the same position of every replay"""
    fn()
    if _REC is not None:
        def again(f=fn):
            f()
            return 0
        _REC.calls.append((again, ()))


def zero_(t: Tensor) -> Tensor:
    """t.zero_() as a C call on the current stream (recordable); t contiguous"""
    assert t.is_contiguous()
    _call("dl_memset_zero", _p(t), t.numel() * t.element_size(), _s())
    return t


def stream_wait(waiter: int, waited: int) -> None:
    """raw stream handles: what is queued on `waited` so far happens before what is queued on `waiter` from now on"""
    if waiter != waited:
        _call("dl_stream_wait_stream", waiter, waited)


# ------------------------------------------------------------------ diffusion heads
def flow_add_noise(x: Tensor, noise: Tensor, t: Tensor, out: Tensor | None = None) -> Tensor:
    out = torch.empty_like(x) if out is None else out
    B = x.shape[0]
    _call("dl_flow_add_noise", _p(x), _p(noise), _p(t), _p(out), B, x.numel() // B, _s())
    return out


def ddpm_add_noise(x: Tensor, noise: Tensor, t: Tensor, sqrt_ab: Tensor, ab: Tensor, out: Tensor | None = None) -> Tensor:
    out = torch.empty_like(x) if out is None else out
    B = x.shape[0]
    _call("dl_ddpm_add_noise", _p(x), _p(noise), _p(t), _p(sqrt_ab), _p(ab), _p(out), B, x.numel() // B, _s())
    return out


def mse_loss_fwd(pred: Tensor, a: Tensor, b: Tensor | None, mode: int) -> Tensor:
    n = pred.numel()
    npart = lib().call("dl_mse_loss_partials", n)
    partial = torch.empty(npart, device=pred.device, dtype=torch.float32)
    loss = torch.empty((), device=pred.device, dtype=torch.float32)
    _call("dl_mse_loss_fwd", _p(pred), _p(a), _p(b), _p(partial), _p(loss), n, mode, _s())
    return loss


def mse_loss_bwd(pred: Tensor, a: Tensor, b: Tensor | None, gscale: float, mode: int, out: Tensor | None = None,
                 gscale_dev: Tensor | None = None) -> Tensor:
    out = torch.empty_like(pred) if out is None else out
    _call("dl_mse_loss_bwd", _p(pred), _p(a), _p(b), float(gscale), _p(gscale_dev), _p(out), pred.numel(), mode, _s())
    return out


def flow_x_to_v(z: Tensor, xhat: Tensor, t: Tensor) -> Tensor:
    v = torch.empty_like(z)
    B = z.shape[0]
    _call("dl_flow_x_to_v", _p(z), _p(xhat), _p(t), _p(v), B, z.numel() // B, _s())
    return v


def flow_x_to_v_bwd(dv: Tensor, t: Tensor) -> Tensor:
    dx = torch.empty_like(dv)
    B = dv.shape[0]
    _call("dl_flow_x_to_v_bwd", _p(dv), _p(t), _p(dx), B, dv.numel() // B, _s())
    return dx


# ------------------------------------------------------------------ sampler steps
def euler_step(x: Tensor, v: Tensor, v_uncond: Tensor | None, guidance: float, t_curr: float, t_prev: float,
               want_x0: bool = True) -> tuple[Tensor, Tensor | None]:
    xp = torch.empty_like(x)
    x0 = torch.empty_like(x) if want_x0 else None
    _call("dl_euler_step", _p(x), _p(v), _p(v_uncond), float(guidance), float(t_curr), float(t_curr - t_prev), _p(xp),
          _p(x0), x.numel(), _s())
    return xp, x0


def euler_maruyama_step(x, v, v_uncond, guidance, noise, x_prev_in, t_curr, t_prev, sigma):
    dt = t_curr - t_prev
    std = sigma * dt**0.5
    xp, mean, x0, lp = (torch.empty_like(x) for _ in range(4))
    _call("dl_euler_maruyama_step", _p(x), _p(v), _p(v_uncond), float(guidance), _p(noise), _p(x_prev_in), float(t_curr),
          float(dt), float(sigma), float(std), _p(xp), _p(mean), _p(x0), _p(lp), x.numel(), _s())
    return xp, mean, x0, lp, std


def ddpm_step(pred, pred_uncond, guidance, xt, noise, t, tables, mean_type: int, clamp_x: bool):
    B = xt.shape[0]
    xp, x0, mean, std, lp = (torch.empty_like(xt) for _ in range(5))
    _call("dl_ddpm_step", _p(pred), _p(pred_uncond), float(guidance), _p(xt), _p(noise), _p(t), _p(tables),
          tables.shape[1], mean_type, int(clamp_x), _p(xp), _p(x0), _p(mean), _p(std), _p(lp), B, xt.numel() // B, _s())
    return xp, x0, mean, std, lp


def ddim_step(pred, pred_uncond, guidance, xt, noise, t, tables, coefs, mean_type: int, clamp_x: bool, eta: float):
    B = xt.shape[0]
    xp, x0, mean = (torch.empty_like(xt) for _ in range(3))
    std = torch.empty_like(xt) if eta > 0 else None
    lp = torch.empty_like(xt) if eta > 0 else None
    _call("dl_ddim_step", _p(pred), _p(pred_uncond), float(guidance), _p(xt), _p(noise), _p(t), _p(tables),
          tables.shape[1], mean_type, _p(coefs), int(clamp_x), float(eta), _p(xp), _p(x0), _p(mean), _p(std), _p(lp), B,
          xt.numel() // B, _s())
    return xp, x0, mean, std, lp


# ------------------------------------------------------------------ GEMMs
def gemm_nt(a: Tensor, b: Tensor, out: Tensor, *, bias: Tensor | None = None, act: int = ACT_NONE,
            pre_out: Tensor | None = None, resid: Tensor | None = None, gate: Tensor | None = None,
            rows_per_gate: int = 1, M: int | None = None, N: int | None = None, K: int | None = None,
            scratch: Tensor | None = None) -> Tensor:
    """out[M,N] = epilogue(a[M,K] @ b[N,K]^T); a/b bf16 2-D (row stride = .stride(0)); out bf16 or f32.
    scratch (f32, plain f32 output only): the bit-reproducible form -- a split contraction keeps its partial images there and
    folds them in a fixed order instead of meeting in f32 atomics (dl_gemm_nt_f32_det)."""
    M = a.shape[0] if M is None else M
    N = b.shape[0] if N is None else N
    K = a.shape[1] if K is None else K
    if scratch is not None:
        assert out.dtype == torch.float32 and bias is None and act == ACT_NONE and pre_out is None and resid is None
        _call("dl_gemm_nt_f32_det", _p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K, _p(scratch), scratch.numel(),
              _s())
        return out
    _call("dl_gemm_nt", _p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K, _p(bias), act,
          F32 if out.dtype == torch.float32 else BF16, _p(pre_out), _p(resid), resid.stride(0) if resid is not None else 0,
          _p(gate), gate.stride(0) if gate is not None else 0, rows_per_gate, _s())
    return out


class _NtProblem(ctypes.Structure):
    """ctypes mirror of dl_nt_problem_t (include/diffulab_hip.h)"""

    _fields_ = [("A", ctypes.c_void_p), ("lda", ctypes.c_int64), ("B", ctypes.c_void_p), ("ldb", ctypes.c_int64), ("C", ctypes.c_void_p),
                ("ldc", ctypes.c_int64), ("M", ctypes.c_int64), ("N", ctypes.c_int64), ("K", ctypes.c_int64), ("bias", ctypes.c_void_p),
                ("resid", ctypes.c_void_p), ("ldr", ctypes.c_int64)]


def gemm_nt_pair(problems: list[tuple]) -> None:
    """two independent bf16 products out_i[M_i, N_i] = a_i[M_i, K_i] @ b_i[N_i, K_i]^T (+ bias_i, + resid_i) as ONE launch when both are
    small (dl_gemm_nt_pair; otherwise the two gemm_nt launches it stands for): entries (a, b, out, bias, resid, M, N, K), M / N / K may
    be None.  Bit-identical to the separate calls."""
    arr = (_NtProblem * len(problems))()
    for q, (a, b, out, bias, resid, M, N, K) in zip(arr, problems):
        assert out.dtype == torch.bfloat16
        q.A, q.lda, q.B, q.ldb, q.C, q.ldc = _p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0)
        q.M, q.N, q.K = (a.shape[0] if M is None else M), (b.shape[0] if N is None else N), (a.shape[1] if K is None else K)
        q.bias, q.resid, q.ldr = _p(bias), _p(resid), (resid.stride(0) if resid is not None else 0)
    keep(arr)
    _call("dl_gemm_nt_pair", ctypes.addressof(arr), len(problems), _s())


def gemm_nt_swiglu(x: Tensor, w_perm: Tensor, u: Tensor | None, h: Tensor) -> bool:
    """fused MLP-up GEMM + SwiGLU; u=None skips the pre-activation store (inference).  False when the shape has no fused
    kernel (caller falls back to gemm_nt + swiglu_fwd)"""
    rc = lib().cdll.dl_gemm_nt_swiglu(_p(x), x.stride(0), _p(w_perm), w_perm.stride(0), _p(u), u.stride(0) if u is not None else 0, _p(h),
                                      h.stride(0), x.shape[0], h.shape[1], x.shape[1], _s())
    if rc == -3:
        return False
    if rc != 0:
        raise RuntimeError(f"dl_gemm_nt_swiglu failed ({rc}): {lib().cdll.dl_last_error().decode()}")
    return True


def mlp_recompute_ok(M: int, D: int, F: int) -> bool:
    """shapes for which the training step keeps no SwiGLU pre-activations (the backward recomputes them: csrc/mlp_bwd.hip) --
    the rule implies that the fused MLP-up forward serves the shape too; DL_MLP_RECOMPUTE=0 turns it off"""
    import os

    # F % 128: the recompute kernel's unit tile (the fused MLP-up forward takes any 2F % 128 with >= 64 tiles).  D <= 512: measured
    # -3 % on the DiT-S/2 step (D = 384, F = 1536); at D = 768, F = 3072 it is neutral (DiT-B + REPA, B = 128) to +3..5 % (joint MMDiT:
    # the wider contraction makes the recomputed GEMM cost more than the bytes it saves).
    return (tuning.on("DL_MLP_RECOMPUTE") and M % 256 == 0 and D % 64 == 0 and D <= 512 and F % 128 == 0
            and (M // 256) * (F // 128) >= 64)


def mlp_u_buffer(alloc, M: int, D: int, F: int, train: bool):
    """the [M, 2F] pre-activation buffer of a block's workspace, or None when the training backward recomputes it"""
    return None if (train and mlp_recompute_ok(M, D, F)) else alloc(M, 2 * F)


def mlp_swiglu_bwd(dt: Tensor, w2t: Tensor, x: Tensor, wp: Tensor, u: Tensor | None, dh: Tensor, du: Tensor) -> None:
    """du = SwiGLU backward of dh = dt @ w2t^T: from the saved pre-activations u, or (u None) recomputed per tile"""
    if u is None:
        if not mlp_dswiglu_recompute(x, wp, dt, w2t, du):
            raise RuntimeError("no saved SwiGLU pre-activations and no recompute kernel for this shape")
        return
    gemm_nt(dt, w2t, dh)
    swiglu_bwd(dh, u, du)


def mlp_dswiglu_recompute(x: Tensor, wp: Tensor, dt: Tensor, w2t: Tensor, du: Tensor) -> bool:
    """du = SwiGLU backward of dh = dt @ w2t^T with u = x @ wp^T recomputed per tile (nothing saved, nothing written but du);
    False when the shape has no such kernel (caller: dgrad GEMM + swiglu_bwd on the saved u)"""
    M, F = x.shape[0], w2t.shape[0]
    return _maybe("dl_mlp_dswiglu_recompute", _p(x), x.stride(0), _p(wp), wp.stride(0), _p(dt), dt.stride(0), _p(w2t), w2t.stride(0),
                  _p(du), du.stride(0), M, F, x.shape[1], dt.shape[1], _s())


def _padded_rows(t: Tensor) -> Tensor:
    """a row buffer allocated as the leading slice of a zero-padded parent (rows rounded up to 64): the parent"""
    base = t._base
    if base is not None and base.dim() == t.dim() and base.shape[1:] == t.shape[1:] and base.data_ptr() == t.data_ptr() \
            and base.shape[0] == (t.shape[0] + 63) // 64 * 64 and base.stride() == t.stride():
        return base
    return t


def gemm_tn(a: Tensor, b: Tensor, out: Tensor, *, M: int | None = None, N: int | None = None, max_wgs: int = 0,
            scratch: Tensor | None = None) -> Tensor:
    """out[M,N] (f32) += a[R,M]^T @ b[R,N].  max_wgs > 0 caps the persistent workgroups (side-stream wgrads).  R must be a
    multiple of 64: operands that are leading slices of zero-padded row buffers are widened to their parents.
    scratch (f32, >= M*N): the bit-reproducible form (dl_gemm_tn_det: partial images per split + fixed-order fold, no atomics)."""
    M = a.shape[1] if M is None else M
    N = b.shape[1] if N is None else N
    if a.shape[0] % 64:
        a, b = _padded_rows(a), _padded_rows(b)
        assert a.shape[0] == b.shape[0], "gemm_tn operands with a ragged row count must be zero-padded row buffers"
    if scratch is not None:
        _call("dl_gemm_tn_det", _p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, a.shape[0], _p(scratch),
              scratch.numel(), _s())
        return out
    _call("dl_gemm_tn_ex", _p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, a.shape[0], int(max_wgs), _s())
    return out


class _WGrad(ctypes.Structure):
    """ctypes mirror of dl_wgrad_t (include/diffulab_hip.h)"""

    _fields_ = [("dy", ctypes.c_void_p), ("ld_dy", ctypes.c_int64), ("x", ctypes.c_void_p), ("ld_x", ctypes.c_int64),
                ("g", ctypes.c_void_p), ("m_out", ctypes.c_int64), ("n_in", ctypes.c_int64)]


def gemm_tn_group(probs: list[tuple[Tensor, Tensor, Tensor]], slab: Tensor, max_wgs: int = 0) -> bool:
    """g_p[M_p, N_p] (f32, contiguous) += dy_p[R, M_p]^T @ x_p[R, N_p] for up to four (dy, x, g) over the same R rows in ONE launch
    without atomics (bit-reproducible): partial tiles per token range go to `slab` (f32 scratch, >= sum M_p N_p elements; 8 x that
    fills the chip) and are folded in a fixed order.  False: the 384 x 192 tile does not divide a shape (nothing was launched)."""
    arr = (_WGrad * len(probs))()
    if probs[0][0].shape[0] % 64:  # leading slices of zero-padded row buffers: widen to the parents like gemm_tn
        probs = [(_padded_rows(dy), _padded_rows(x), g) for dy, x, g in probs]
    R = probs[0][0].shape[0]
    for q, (dy, x, g) in zip(arr, probs):
        assert dy.shape[0] == R and x.shape[0] == R and g.is_contiguous()
        q.dy, q.ld_dy, q.x, q.ld_x, q.g, q.m_out, q.n_in = _p(dy), dy.stride(0), _p(x), x.stride(0), _p(g), g.shape[0], g.shape[1]
    global _LIB
    if _LIB is None:
        _LIB = lib()
    fn = _LIB._fns["dl_gemm_tn_group"][0]
    rc = fn(ctypes.addressof(arr), len(probs), R, _p(slab), slab.numel(), int(max_wgs), _s())
    if rc == -3:  # DL_ERR_UNSUPPORTED
        return False
    if rc != 0:
        raise RuntimeError(f"dl_gemm_tn_group failed ({rc}): {_LIB.cdll.dl_last_error().decode()}")
    return True


_SCRATCH: dict = {}


def shared_scratch(device, floats: int) -> Tensor:
    """grow-only f32 scratch per device for the partial-image forms (gemm_tn / gemm_nt / colsum with scratch=) of modules that own
    no workspace (REPA projection head, Perceiver resampler).  Its users run on one stream in issue order, each launch pair
    (partials, fold) completes before the next begins, so one buffer serves them all."""
    key = (torch.device(device).type, torch.device(device).index)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < floats:
        buf = torch.empty(int(floats), device=device, dtype=torch.float32)
        _SCRATCH[key] = buf
    return buf


def wgrad_tile_ok(m_out: int, n_in: int) -> bool:
    """dl_gemm_tn_group has a tile for a [m_out, n_in] weight gradient (whole 384 x 192 or whole 256 x 256 tiles)"""
    return (m_out % 384 == 0 and n_in % 192 == 0) or (m_out % 256 == 0 and n_in % 256 == 0)


class WgradGroups:
    """Collects the weight-gradient problems (dy, x, g) an engine's backward produces and issues them as dl_gemm_tn_group launches:
    up to four problems over the SAME token rows per launch, no atomics, bit-reproducible (csrc/gemm_w4.hip).  Problems are keyed
    by their row count (the two token streams of a joint block have different ones); a key's list is launched when it holds four
    problems and at flush().  `launch(fn)` runs fn where the engine wants its weight gradients (side stream behind an event).
    Shapes the 384 x 192 tile does not divide fall back to one dl_gemm_tn_ex launch per problem."""

    def __init__(self, slab: Tensor, launch, max_wgs: int = 0) -> None:
        self.slab, self.launch, self.max_wgs = slab, launch, max_wgs
        self.pending: dict[int, list[tuple[Tensor, Tensor, Tensor]]] = {}

    @staticmethod
    def slab_floats(D: int, F: int, ranges: int = 8) -> int:
        """scratch for the four linears of a block with inner width D and MLP width F (qkv, proj, up, down) at `ranges` token ranges"""
        return ranges * (4 * D * D + 3 * D * F)

    @staticmethod
    def widths_ok(D: int, F: int) -> bool:
        """the four linears of a block with inner width D and MLP width F are whole 384 x 192 tiles or whole 256 x 256 tiles -- or
        (round 6: 640-wide models) at least one 256 x 256 tile in each direction with widths that are multiples of 8: the last tile of
        a row / column is then shifted back to end at the edge"""
        if (D % 384 == 0 and F % 192 == 0) or (D % 256 == 0 and F % 256 == 0):
            return True
        r256 = lambda v: (v + 255) // 256 * 256  # noqa: E731
        return D % 8 == 0 and F % 8 == 0 and D >= 256 and F >= 256 and 4 * r256(D) <= 5 * D and 4 * r256(F) <= 5 * F  # (<= 25 % recomputed)

    @staticmethod
    def shapes_ok(D: int, F: int, rows: int) -> bool:
        return WgradGroups.widths_ok(D, F) and rows % 32 == 0 and rows >= 2048

    def add(self, dy: Tensor, x: Tensor, g: Tensor) -> None:
        lst = self.pending.setdefault(dy.shape[0], [])
        lst.append((dy, x, g))
        if len(lst) == 4:
            self._issue(dy.shape[0])

    def _issue(self, key: int) -> None:
        probs = self.pending.pop(key, [])
        if not probs:
            return

        def run() -> None:
            if not gemm_tn_group(probs, self.slab, max_wgs=self.max_wgs):
                for dy, x, g in probs:
                    gemm_tn(dy, x, g, max_wgs=self.max_wgs)

        self.launch(run)

    def flush(self) -> None:
        for key in list(self.pending):
            self._issue(key)


def grouped_wgrad_fn(G, slab: Tensor | None, on_side, max_wgs: int = 0):
    """the `wgrad(dy, x, parameter name)` hook of an engine's backward: with a slab the block's linears are collected and issued as
    atomics-free dl_gemm_tn_group launches (WgradGroups; `wgrad.flush()` issues what is pending -- call it at every block end and
    before the side stream is joined), without one each is a dl_gemm_tn_ex launch.  G(name) -> the f32 gradient view; on_side(fn)
    runs fn where the engine wants its weight gradients (side stream behind an event of the main stream)."""
    groups = WgradGroups(slab, on_side, max_wgs=max_wgs) if slab is not None else None

    def wgrad(x_grad: Tensor, x_in: Tensor, gname: str) -> None:
        if groups is not None:
            groups.add(x_grad, x_in, G(gname))
        else:
            on_side(lambda: gemm_tn(x_grad, x_in, G(gname), max_wgs=max_wgs))

    wgrad.flush = groups.flush if groups is not None else (lambda: None)
    return wgrad


# ------------------------------------------------------------------ block kernels
def ln_modulate_fwd(x, w, b, scale, shift, rows_per_mod, eps, out, mean, rstd, t=None, gate=None, x_out=None):
    """out = modulate(LN(x')) with x' = x (+ gate * t, written to x_out, when the gated residual is fused in)"""
    M, D = x.shape
    _call("dl_ln_modulate_fwd", _p(x), _p(w), _p(b), _p(scale), _p(shift), scale.stride(0), rows_per_mod, float(eps),
          _p(out), _p(mean), _p(rstd), _p(t), _p(gate), gate.stride(0) if gate is not None else 0, _p(x_out), M, D, _s())


def ln_modulate_bwd(dout, x, w, b, scale, rows_per_mod, mean, rstd, dres, dx, dscale, dshift, dwb_partial, gate_t=None,
                    gate=None, dt=None, dgate=None):
    """dscale / dshift (/ dgate): f32 views [groups, D] with the same row stride, accumulated into; dwb_partial f32
    [groups, 2, D] or None.  gate_t / gate / dt / dgate: fused backward of the gated residual that follows (see header)."""
    M, D = x.shape
    assert dscale.dtype == torch.float32 and dshift.dtype == torch.float32
    assert dgate is None or (dgate.dtype == torch.float32 and dgate.stride(0) == dscale.stride(0))
    _call("dl_ln_modulate_bwd", _p(dout), _p(x), _p(w), _p(b), _p(scale), scale.stride(0), rows_per_mod, _p(mean),
          _p(rstd), _p(dres), _p(dx), _p(dscale), _p(dshift), dscale.stride(0), _p(dwb_partial), _p(gate_t), _p(gate),
          gate.stride(0) if gate is not None else 0, _p(dt), _p(dgate), M, D, _s())


def row_gemm_ok(M: int, D: int, rows_per_mod: int) -> bool:
    """shapes served by the row-complete GEMMs (csrc/gemm_ln.hip): a 256 x 384 tile is a whole sample's whole rows.
    DL_ROW_GEMM=0 restores the GEMM + row-kernel launch pairs (A/B switch)"""
    return tuning.on("DL_ROW_GEMM") and D == 384 and rows_per_mod == 256 and M % 256 == 0 and M // 256 >= 8


def ln_modulate_gemm_fwd(a, w_sh, resid, gate, ln_w, ln_b, scale, shift, rows_per_mod, eps, t_out, x_out, xm_out, mean, rstd,
                         K=None) -> bool:
    """t = a @ w_sh^T; x' = resid + gate * t; xm = modulate(LN(x')) in one launch; False when the shape has no such kernel"""
    M, D = xm_out.shape
    return _maybe("dl_ln_modulate_gemm_fwd", _p(a), a.stride(0), _p(w_sh), w_sh.stride(0), M, a.shape[1] if K is None else K, _p(resid),
                  _p(gate), gate.stride(0) if gate is not None else 0, _p(ln_w), _p(ln_b), _p(scale), _p(shift), scale.stride(0),
                  rows_per_mod, float(eps), _p(t_out), _p(x_out), _p(xm_out), _p(mean), _p(rstd), D, _s())


def ln_modulate_gemm_bwd(a, wt_sh, x, ln_w, ln_b, scale, rows_per_mod, mean, rstd, dres, dx, dscale, dshift, dwb_partial,
                         gate_t=None, gate=None, dt=None, dgate=None, K=None) -> bool:
    """dout = a @ wt_sh^T (never written) -> LayerNorm-modulate backward; dscale / dshift / dgate / dwb_partial are WRITTEN"""
    M, D = x.shape
    assert dscale.dtype == torch.float32 and dshift.dtype == torch.float32 and dshift.stride(0) == dscale.stride(0)
    assert dgate is None or (dgate.dtype == torch.float32 and dgate.stride(0) == dscale.stride(0))
    return _maybe("dl_ln_modulate_gemm_bwd", _p(a), a.stride(0), _p(wt_sh), wt_sh.stride(0), M, a.shape[1] if K is None else K, _p(x),
                  _p(ln_w), _p(ln_b), _p(scale), scale.stride(0), rows_per_mod, _p(mean), _p(rstd), _p(dres), _p(dx), _p(dscale),
                  _p(dshift), dscale.stride(0), _p(dwb_partial), _p(gate_t), _p(gate), gate.stride(0) if gate is not None else 0,
                  _p(dt), _p(dgate), D, _s())


def gemm_nt_qk_norm_rope(a, w_qkv, scale_q, scale_k, cos, sin, qkv, q, k, rrms, B, N, H, dh, rot, eps=1e-6, n_off=0) -> bool:
    """qkv = a @ w_qkv^T plus RMSNorm + RoPE + head split of its q / k thirds in one launch"""
    return _maybe("dl_gemm_nt_qk_norm_rope", _p(a), a.stride(0), _p(w_qkv), w_qkv.stride(0), B, N, H, dh, rot, float(eps),
                  _p(scale_q), _p(scale_k), _p(cos), _p(sin), _p(qkv), _p(q), _p(k), _p(rrms), q.shape[2], n_off, _s())


def ln_modulate_bwd_tok(dout, x, w, b, scale, mean, rstd, dres, dx, dscale, dshift, dwb_partial, gate_t=None, gate=None, dt=None,
                        dgate=None):
    """per-token modulation: scale / gate rows per token; dscale / dshift / dgate bf16 row windows written per token"""
    M, D = x.shape
    assert dscale.dtype == torch.bfloat16 and dshift.stride(0) == dscale.stride(0)
    _call("dl_ln_modulate_bwd_tok", _p(dout), _p(x), _p(w), _p(b), _p(scale), scale.stride(0), _p(mean), _p(rstd), _p(dres), _p(dx),
          _p(dscale), _p(dshift), dscale.stride(0), _p(dwb_partial), dwb_partial.shape[0] if dwb_partial is not None else 0,
          _p(gate_t), _p(gate), gate.stride(0) if gate is not None else 0, _p(dt), _p(dgate), M, D, _s())


def ddt_cond_fwd(enc, temb, B, N, out):
    _call("dl_ddt_cond_fwd", _p(enc), enc.stride(0), _p(temb), temb.stride(0), B, N, out.shape[1], _p(out), _s())


def ddt_cond_bwd(dsz, enc, temb, B, N, denc, dtemb):
    _call("dl_ddt_cond_bwd", _p(dsz), _p(enc), enc.stride(0), _p(temb), temb.stride(0), B, N, dsz.shape[1], _p(denc), _p(dtemb), _s())


def gate_bwd(dout, t, gate, rows_per_mod, dt, dgate):
    M, D = dout.shape
    _call("dl_gate_bwd", _p(dout), _p(t), _p(gate), gate.stride(0), rows_per_mod, _p(dt), _p(dgate), dgate.stride(0), M,
          D, _s())


def qk_norm_rope_fwd(qkv, scale_q, scale_k, cos, sin, q, k, v, rrms, B, N, H, dh, rot, eps=1e-6, pos=None, n_off=0):
    """pos: int32 [B*N] table row of every token (None: its index in the sequence); q / k / v are [B, H, n_dst, dh] and the N
    tokens are written to rows [n_off, n_off + N)"""
    _call("dl_qk_norm_rope_fwd_ex", _p(qkv), _p(scale_q), _p(scale_k), _p(cos), _p(sin), _p(q), _p(k), _p(v), _p(rrms), B, N,
          H, dh, rot, float(eps), _p(pos), q.shape[2], n_off, _s())


def qk_norm_rope_bwd(dq, dk, dv, qkv, scale_q, scale_k, cos, sin, rrms, dqkv, dscale, B, N, H, dh, rot, pos=None, n_off=0):
    _call("dl_qk_norm_rope_bwd_ex", _p(dq), _p(dk), _p(dv), _p(qkv), _p(scale_q), _p(scale_k), _p(cos), _p(sin), _p(rrms),
          _p(dqkv), _p(dscale), B, N, H, dh, rot, _p(pos), dq.shape[2], n_off, _s())


def copy_rows3d(src, src_bs, src_rs, dst, dst_bs, dst_rs, B, rows, cols):
    """dst[b, r, :cols] = src[b, r, :cols] with explicit batch / row strides (elements); src / dst may be offset views"""
    _call("dl_copy_rows3d", _p(src), src_bs, src_rs, _p(dst), dst_bs, dst_rs, B, rows, cols, _s())


# ------------------------------------------------------------------ SPRINT token routing
def gather_tokens(src, idx, dst, B, N, k, D, keep=None):
    _call("dl_gather_tokens", _p(src), src.stride(0), _p(idx), _p(keep), _p(dst), dst.stride(0), B, N, k, D, _s())


def scatter_tokens_add(src, idx, dst, B, N, k, D):
    _call("dl_scatter_tokens_add", _p(src), src.stride(0), _p(idx), _p(dst), dst.stride(0), B, N, k, D, _s())


def restore_tokens(xd, inv, mask, out, B, N, k, D):
    _call("dl_restore_tokens", _p(xd), xd.stride(0), _p(inv), _p(mask), _p(out), out.stride(0), B, N, k, D, _s())


def masked_colsum(x, sel, out, R, C):
    _call("dl_masked_colsum", _p(x), x.stride(0), _p(sel), _p(out), R, C, _s())


def gated_residual_fwd(x, t, gate, rows_per_mod, out):
    M, D = x.shape
    _call("dl_gated_residual_fwd", _p(x), _p(t), _p(gate), gate.stride(0), rows_per_mod, _p(out), out.stride(0), M, D, _s())


def attn_fwd(q, k, v, out, lse, B, H, N, dh, scale):
    _call("dl_attn_fwd", _p(q), _p(k), _p(v), _p(out), _p(lse), B, H, N, dh, float(scale), _s())


def attn_bwd(q, k, v, out, dout, lse, dq, dk, dv, B, H, N, dh, scale):
    _call("dl_attn_bwd", _p(q), _p(k), _p(v), _p(out), _p(dout), _p(lse), _p(dq), _p(dk), _p(dv), B, H, N, dh,
          float(scale), _s())


def attn_needs_padding(N: int) -> bool:
    """token counts the resident / tiled attention kernels do not take as they are (multiples of 64 up to 256, of 256 up to 2048):
    the caller pads q / k / v to a multiple of 256 rows and masks the pad keys (dl_attn_fwd_ex / _bwd_ex with a key bias)"""
    return N % 64 != 0 or (N > 256 and N % 256 != 0)


def v_in_place(N: int) -> bool:
    """up to 256 tokens the attention kernels address V / dV inside the token-major qkv / dqkv rows (no head-split copy of V);
    DL_ATTN_V_IN_PLACE=0 restores the head-major V buffers (A/B switch)"""
    return N <= 256 and N % 64 == 0 and tuning.on("DL_ATTN_V_IN_PLACE")


def dit_block_fwd(blk, train: bool) -> None:
    """one adaLN-zero DiT block forward issued by the library (blk: diffulab_amd._block.DitBlock)"""
    _call("dl_dit_block_fwd", ctypes.addressof(blk), int(train), _s())


def dit_block_bwd(blk, main_stream: int, side_stream: int, side_wgs: int) -> None:
    _call("dl_dit_block_bwd", ctypes.addressof(blk), main_stream, side_stream, int(side_wgs))


def attn_fwd_qkv(q, k, qkv, out, lse, B, H, N, dh, scale):
    """N <= 256: V is read in place from the v third of the token-major qkv rows [B*N, 3*H*dh]"""
    D = H * dh
    _call("dl_attn_fwd_sv", _p(q), _p(k), qkv.data_ptr() + 2 * D * 2, N * 3 * D, dh, 3 * D, _p(out), _p(lse), B, H, N, dh,
          float(scale), _s())


def gemm_nt_ssq(a, w_sh, out, ssq) -> bool:
    """out[M, N] = a w_sh^T (bf16, persistent 256 x 384 tiles) + ssq[M, T] += row sums of squares of the first T = ssq.shape[1] 384-wide
    column tiles (ssq f32, zeroed by the caller): the qkv GEMM that also leaves the QK-RMSNorm statistics.  False: no such kernel for
    the shape (the caller keeps gemm_nt + qk_norm_rope_fwd)"""
    return _maybe("dl_gemm_nt_ssq", _p(a), a.stride(0), _p(w_sh), w_sh.stride(0), _p(out), out.stride(0), a.shape[0], w_sh.shape[0],
                  a.shape[1], _p(ssq), ssq.shape[1], _s())


def attn_fwd_qkn(qkv, ssq, scale_q, scale_k, cos, sin, q, k, rrms, out, lse, B, H, N, dh, rot, scale, eps=1e-6):
    """attention forward from the PRE-NORM token-major qkv rows: QK-RMSNorm (statistics = ssq of gemm_nt_ssq) + RoPE applied as q
    and k are staged; writes the normalised q, k head-major and rrms [M, 2] for the backward kernels (N <= 256)"""
    _call("dl_attn_fwd_qkn", _p(qkv), _p(ssq), _p(scale_q), _p(scale_k), _p(cos), _p(sin), float(eps), rot, _p(q), _p(k), _p(rrms), _p(out),
          _p(lse), B, H, N, dh, float(scale), _s())


def attn_bwd_qkv(q, k, qkv, out, dout, lse, dq, dk, dqkv, B, H, N, dh, scale):
    """N <= 256: V read from qkv, dV written into the v third of dqkv [B*N, 3*H*dh] (qk_norm_rope_bwd then takes dv=None)"""
    D = H * dh
    _call("dl_attn_bwd_sv", _p(q), _p(k), qkv.data_ptr() + 2 * D * 2, N * 3 * D, dh, 3 * D, _p(out), _p(dout), _p(lse), _p(dq),
          _p(dk), dqkv.data_ptr() + 2 * D * 2, N * 3 * D, dh, 3 * D, B, H, N, dh, float(scale), _s())


def attn_bwd_tok(q, k, qkv, out, dout, lse, dqkv, B, H, N, dh, scale):
    """N <= 256: V read from qkv; dQ, dK AND dV written token-major into dqkv [B*N, 3*H*dh] (qk_norm_rope_bwd_inplace follows)"""
    _call("dl_attn_bwd_tok", _p(q), _p(k), _p(qkv), _p(out), _p(dout), _p(lse), _p(dqkv), B, H, N, dh, float(scale), _s())


def qk_inplace_ok(ws: dict, B: int, tokens: int) -> bool:
    """an engine with the partials scratch ws["qk_part"] takes the token-major attention backward + in-place QK-norm backward for
    this launch: V in place (<= 256 tokens per sample) and at least 32768 token rows (below that the two-kernel form is faster)"""
    return ws.get("qk_part") is not None and v_in_place(tokens) and B * tokens >= 32768


def qk_norm_rope_bwd_inplace(qkv, scale_q, scale_k, cos, sin, rrms, dqkv, dscale, partials, B, N, H, dh, rot, pos=None) -> bool:
    """the q / k thirds of dqkv: gradient after norm + RoPE (token-major) -> gradient of the pre-norm q / k, in place; dscale f32
    [2, D] += the scale gradients through `partials` (f32 scratch >= 1024 * 2 * D) in a fixed order.  False: inner width > 512"""
    assert partials.numel() >= 1024 * 2 * H * dh and partials.dtype == torch.float32
    return _maybe("dl_qk_norm_rope_bwd_inplace", _p(qkv), _p(scale_q), _p(scale_k), _p(cos), _p(sin), _p(rrms), _p(dqkv), _p(dscale),
                  _p(partials), B, N, H, dh, rot, _p(pos), _s())


def attn_fwd_ex(q, k, v, out, lse, B, H, Nq, Nk, dh, scale, key_bias=None):
    """Nq queries against Nk keys (multiples of 256), optional additive key bias f32 [B, Nk] (0 / -inf)"""
    _call("dl_attn_fwd_ex", _p(q), _p(k), _p(v), _p(out), _p(lse), B, H, Nq, Nk, dh, float(scale), _p(key_bias), _s())


def attn_bwd_ex(q, k, v, out, dout, lse, dq, dk, dv, B, H, Nq, Nk, dh, scale, key_bias=None):
    _call("dl_attn_bwd_ex", _p(q), _p(k), _p(v), _p(out), _p(dout), _p(lse), _p(dq), _p(dk), _p(dv), B, H, Nq, Nk, dh,
          float(scale), _p(key_bias), _s())


def swiglu_fwd(u, h):
    _call("dl_swiglu_fwd", _p(u), _p(h), u.shape[0], h.shape[1], _s())


def swiglu_bwd(dh, u, du):
    _call("dl_swiglu_bwd", _p(dh), _p(u), _p(du), u.shape[0], dh.shape[1], _s())


# ------------------------------------------------------------------ stem / head / conditioning
def patchify(x, tok, p, order):
    B, C, H, W = x.shape
    _call("dl_patchify", _p(x), _p(tok), B, C, H, W, p, tok.stride(0), order, _s())


def unpatchify(tok, img, p):
    B, C, H, W = img.shape
    _call("dl_unpatchify", _p(tok), _p(img), B, C, H, W, p, tok.stride(0), _s())


def timestep_embedding(t, out, max_period=10000.0):
    _call("dl_timestep_embedding", _p(t), _p(out), out.shape[0], out.shape[1], float(max_period), _s())


def cond_combine_fwd(e, table, idx, emb, act):
    _call("dl_cond_combine_fwd", _p(e), _p(table), _p(idx), _p(emb), _p(act), e.shape[0], e.shape[1], _s())


def cond_combine_bwd(dact, emb, idx, demb, demb16, dtable):
    _call("dl_cond_combine_bwd", _p(dact), _p(emb), _p(idx), _p(demb), _p(demb16), _p(dtable), emb.shape[0], emb.shape[1],
          _s())


def silu_bwd(dy, pre, dx):
    _call("dl_silu_bwd", _p(dy), _p(pre), _p(dx), dy.numel(), _s())


def gelu_bwd(dy, pre, dx):
    _call("dl_gelu_bwd", _p(dy), _p(pre), _p(dx), dy.numel(), _s())


def heads_split_rope(src, dst, B, H, n_src, n_off, cos=None, sin=None, rot=0):
    """src [B*n_src, ld] -> dst [B, H, n_dst, 64] rows n_off.. (+ RoPE on the first `rot` channels when cos/sin are given)"""
    _call("dl_heads_split_rope", _p(src), src.stride(0), _p(dst), B, H, n_src, dst.shape[2], n_off, _p(cos), _p(sin), rot, _s())


def heads_merge_rope_bwd(dst_grad, src_grad, B, H, n_src, n_off, cos=None, sin=None, rot=0, accumulate=False):
    _call("dl_heads_merge_rope_bwd", _p(dst_grad), _p(src_grad), src_grad.stride(0), B, H, n_src, dst_grad.shape[2], n_off,
          _p(cos), _p(sin), rot, int(accumulate), _s())


def colsum(x, out, R=None, C=None, scratch=None):
    """out[c] += sum_r x[r, c]; scratch (f32): the bit-reproducible form (row-slab partials + fixed-order fold, dl_colsum_det)"""
    R = x.shape[0] if R is None else R
    C = x.shape[1] if C is None else C
    if scratch is not None:
        _call("dl_colsum_det", _p(x), F32 if x.dtype == torch.float32 else BF16, x.stride(0), _p(out), R, C, _p(scratch), scratch.numel(),
              _s())
        return
    _call("dl_colsum", _p(x), F32 if x.dtype == torch.float32 else BF16, x.stride(0), _p(out), R, C, _s())


class _ColsumDesc(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("ld", ctypes.c_int64), ("out", ctypes.c_void_p), ("R", ctypes.c_int64), ("C", ctypes.c_int64)]


def colsum_batched(items: list) -> None:
    """items: [(x bf16 [R, ld] rows, out f32 [C], R, C)]: out[c] += sum_r x[r, c] for all of them in one launch per 24"""
    arr = (_ColsumDesc * len(items))()
    for i, (x, out, R, C) in enumerate(items):
        assert x.dtype == torch.bfloat16 and out.dtype == torch.float32
        arr[i] = _ColsumDesc(_p(x), x.stride(0), _p(out), R, C)
    _call("dl_colsum_batched", ctypes.cast(arr, ctypes.c_void_p), len(items), _s())


def reduce_rows_f32(partial, out, G, n, clear=False):
    _call("dl_reduce_rows_f32", _p(partial), _p(out), G, n, int(clear), _s())


def reduce_rows_batched_f32(partial, partial_stride, out, out_stride, K, G, n):
    """K folds in one launch: out[k * out_stride + j] += sum_g partial[k * partial_stride + g * n + j] (deterministic)"""
    _call("dl_reduce_rows_batched_f32", _p(partial), partial_stride, _p(out), out_stride, K, G, n, _s())


def cosine_rows_fwd(p, d, cosv, pn2, dn2, eps=1e-8):
    M, E = p.shape
    _call("dl_cosine_rows_fwd", _p(p), p.stride(0), _p(d), d.stride(0), _p(cosv), _p(pn2), _p(dn2), M, E, float(eps), _s())


def cosine_rows_bwd(p, d, cosv, pn2, dn2, gscale, gscale_dev, dp, eps=1e-8):
    M, E = p.shape
    _call("dl_cosine_rows_bwd", _p(p), p.stride(0), _p(d), d.stride(0), _p(cosv), _p(pn2), _p(dn2), float(gscale),
          _p(gscale_dev), _p(dp), dp.stride(0), M, E, float(eps), _s())


# ------------------------------------------------------------------ optimizer side
def adamw_step(p, g, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    _call("dl_adamw_step", _p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
          float(wd), float(1.0 - beta1**step), float(1.0 - beta2**step), float(grad_scale), _s())


def adamw_step_dev(p, g, m, v, hyper):
    """AdamW with the scalars read from the device buffer `hyper` f32 [8] (graph-replayable launch)"""
    _call("dl_adamw_step_dev", _p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), _s())


def cast_weight(src, dst, dst_t):
    R, C = src.shape
    _call("dl_cast_weight", _p(src), R, C, _p(dst), dst.stride(0) if dst is not None else 0, _p(dst_t),
          dst_t.stride(0) if dst_t is not None else 0, _s())


class CastTable:
    """device table for dl_cast_weights_batched: every bf16 shadow of a network refreshed by ONE launch.
    entries: (src f32 [R, C], dst | None, dst_t | None, dst_swiglu | None)"""

    def __init__(self, entries: list[tuple[Tensor, Tensor | None, Tensor | None, Tensor | None]]) -> None:
        import ctypes

        class Desc(ctypes.Structure):
            _fields_ = [("src", ctypes.c_void_p), ("R", ctypes.c_int64), ("C", ctypes.c_int64), ("dst", ctypes.c_void_p),
                        ("ld_dst", ctypes.c_int64), ("dst_t", ctypes.c_void_p), ("ld_t", ctypes.c_int64),
                        ("dst_swiglu", ctypes.c_void_p), ("ld_swiglu", ctypes.c_int64), ("tile_begin", ctypes.c_int64),
                        ("tiles_c", ctypes.c_int64)]

        arr = (Desc * len(entries))()
        tiles = 0
        self.keep = entries  # the table holds raw pointers: keep the tensors alive
        for i, (src, dst, dst_t, dst_g) in enumerate(entries):
            R, C = src.shape
            assert src.is_contiguous() and src.dtype == torch.float32
            cmax = max(C, dst.stride(0) if dst is not None else 0, dst_g.stride(0) if dst_g is not None else 0)
            rmax = max(R, dst_t.stride(0) if dst_t is not None else 0)
            tc, tr = (cmax + 31) // 32, (rmax + 31) // 32
            arr[i] = Desc(_p(src), R, C, _p(dst), dst.stride(0) if dst is not None else 0, _p(dst_t),
                          dst_t.stride(0) if dst_t is not None else 0, _p(dst_g), dst_g.stride(0) if dst_g is not None else 0,
                          tiles, tc)
            tiles += tc * tr
        self.n, self.tiles = len(entries), tiles
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.dev = host.to(entries[0][0].device)

    def run(self) -> None:
        _call("dl_cast_weights_batched", _p(self.dev), self.n, self.tiles, _s())


class ConvCastTable:
    """device table for dl_cast_conv3x3_weights_batched: the forward / data-gradient shadows of every 3x3 convolution weight of a
    network refreshed by ONE launch.  entries: (w f32 [Co, Ci, 3, 3], wf [Co, 9 Ci], wd [Ci, 9 Co]); `accepts` says which weights fit"""

    @staticmethod
    def accepts(w: Tensor, wf: Tensor, wd: Tensor) -> bool:
        co, ci = w.shape[0], w.shape[1]
        return co % 32 == 0 and ci % 32 == 0 and wf.stride(0) == 9 * ci and wd.stride(0) == 9 * co

    def __init__(self, entries: list[tuple[Tensor, Tensor, Tensor]]) -> None:
        import ctypes

        class Desc(ctypes.Structure):
            _fields_ = [("w", ctypes.c_void_p), ("Co", ctypes.c_int64), ("Ci", ctypes.c_int64), ("wf", ctypes.c_void_p),
                        ("ldf", ctypes.c_int64), ("wd", ctypes.c_void_p), ("ldd", ctypes.c_int64), ("tile_begin", ctypes.c_int64)]

        arr = (Desc * len(entries))()
        tiles = 0
        self.keep = entries  # the table holds raw pointers: keep the tensors alive
        for i, (w, wf, wd) in enumerate(entries):
            assert w.is_contiguous() and w.dtype == torch.float32 and self.accepts(w, wf, wd)
            co, ci = w.shape[0], w.shape[1]
            arr[i] = Desc(_p(w), co, ci, _p(wf), wf.stride(0), _p(wd), wd.stride(0), tiles)
            tiles += (co // 32) * (ci // 32)
        self.n, self.tiles = len(entries), tiles
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.dev = host.to(entries[0][0].device)

    def run(self) -> None:
        _call("dl_cast_conv3x3_weights_batched", _p(self.dev), self.n, self.tiles, _s())


class ConvFoldTable:
    """device table for dl_conv3x3_wgrad_fold_batched: every staged convolution weight gradient (g f32 [9 Ci, ldg], transposed as
    dl_conv3x3_wgrad_tn leaves it, or f32 [n_img, 9 Ci, ldg]: the partial images of dl_conv3x3_wgrad_tn_parts, added in image
    order) folded into its [Co, Ci, 3, 3] gradient by ONE launch.  entries: (g, dw)"""

    TAP_SPLIT_MIN_IMAGES = 8  # DL_FOLD_TAP_SPLIT_MIN_IMAGES of include/diffulab_hip.h

    @staticmethod
    def accepts(co: int, ci: int) -> bool:
        return co % 32 == 0 and ci % 32 == 0

    def __init__(self, entries: list[tuple[Tensor, Tensor]]) -> None:
        import ctypes

        class Desc(ctypes.Structure):
            _fields_ = [("g", ctypes.c_void_p), ("ldg", ctypes.c_int64), ("dw", ctypes.c_void_p), ("Co", ctypes.c_int64),
                        ("Ci", ctypes.c_int64), ("tile_begin", ctypes.c_int64), ("n_img", ctypes.c_int64), ("img_stride", ctypes.c_int64)]

        arr = (Desc * len(entries))()
        tiles = 0
        self.keep = entries  # the table holds raw pointers: keep the tensors alive
        for i, (g, dw) in enumerate(entries):
            co, ci = dw.shape[0], dw.shape[1]
            assert dw.is_contiguous() and dw.dtype == torch.float32 and g.dtype == torch.float32 and self.accepts(co, ci)
            if g.dim() == 3:  # partial images
                assert g.shape[1] >= 9 * ci and g.stride(1) >= co
                arr[i] = Desc(_p(g), g.stride(1), _p(dw), co, ci, tiles, g.shape[0], g.stride(0))
                if g.shape[0] >= self.TAP_SPLIT_MIN_IMAGES:  # one tap of a channel tile per workgroup, 16-byte loads
                    assert g.stride(1) % 4 == 0 and g.stride(0) % 4 == 0 and _p(g) % 16 == 0
                    tiles += 8 * (co // 32) * (ci // 32)
            else:
                assert g.shape[0] >= 9 * ci and g.stride(0) >= co
                arr[i] = Desc(_p(g), g.stride(0), _p(dw), co, ci, tiles, 0, 0)
            tiles += (co // 32) * (ci // 32)
        self.n, self.tiles = len(entries), tiles
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.dev = host.to(entries[0][0].device)

    def run(self, clear: bool = True) -> None:
        _call("dl_conv3x3_wgrad_fold_batched", _p(self.dev), self.n, self.tiles, int(clear), _s())


def cast_weight_swiglu(src, dst):
    _call("dl_cast_weight_swiglu", _p(src), src.shape[0] // 2, src.shape[1], _p(dst), dst.stride(0), _s())


def cast_f32_to_bf16(src, dst):
    _call("dl_cast_f32_to_bf16", _p(src), _p(dst), src.numel(), _s())


def cast2d_f32_to_bf16(src, dst):
    """dst[r, c] = bf16(src[r, c]) over 2-D views whose rows may be windows of wider rows"""
    _call("dl_cast2d_f32_to_bf16", _p(src), src.stride(0), _p(dst), dst.stride(0), src.shape[0], src.shape[1], _s())


def cast_bf16_to_f32(src, dst):
    _call("dl_cast_bf16_to_f32", _p(src), _p(dst), src.numel(), _s())


def ema_update(ema, p, beta):
    _call("dl_ema_update", _p(ema), _p(p), float(beta), p.numel(), _s())


# ------------------------------------------------------------------ UNet kernels (NHWC bf16 token rows)
def nchw_to_nhwc(x, out, B, C, HW):
    _call("dl_nchw_to_nhwc", _p(x), _p(out), B, C, HW, out.stride(0), _s())


def nhwc_to_nchw(x, out, B, C, HW):
    _call("dl_nhwc_to_nchw", _p(x), _p(out), B, C, HW, x.stride(0), _s())


def gn_stats(x, stats, B, HW, C, G=32, eps=1e-5):
    _call("dl_gn_stats", _p(x), _p(stats), B, HW, C, G, float(eps), _s())


def gn_apply_fwd(x, stats, w, b, film_scale, film_shift, silu, out, B, HW, C, G=32):
    _call("dl_gn_apply_fwd", _p(x), _p(stats), _p(w), _p(b), _p(film_scale), _p(film_shift),
          film_scale.stride(0) if film_scale is not None else 0, int(silu), _p(out), B, HW, C, G, _s())


def gn_fwd(x, stats, w, b, film_scale, film_shift, silu, out, B, HW, C, G=32, eps=1e-5):
    """statistics (written to `stats`) + normalise / FiLM / SiLU in one call"""
    _call("dl_gn_fwd", _p(x), _p(w), _p(b), _p(film_scale), _p(film_shift), film_scale.stride(0) if film_scale is not None else 0,
          int(bool(silu)), _p(out), _p(stats), B, HW, C, G, float(eps), _s())


def gn_bwd(dout, x, stats, w, b, film_scale, film_shift, silu, dres, dx, dw, db, dfilm_scale, dfilm_shift, scratch, B, HW, C,
           G=32):
    _call("dl_gn_bwd", _p(dout), _p(x), _p(stats), _p(w), _p(b), _p(film_scale), _p(film_shift),
          film_scale.stride(0) if film_scale is not None else 0, int(silu), _p(dres), _p(dx), _p(dw), _p(db), _p(dfilm_scale),
          _p(dfilm_shift), dfilm_scale.stride(0) if dfilm_scale is not None else 0, _p(scratch), B, HW, C, G, _s())


def im2col3x3(x, cols, B, H, W, C):
    _call("dl_im2col3x3", _p(x), x.stride(0), _p(cols), B, H, W, C, cols.shape[0], cols.stride(0), _s())


def _maybe(name: str, *args) -> bool:
    """call an entry point that may answer DL_ERR_UNSUPPORTED (-3): False then, True on success"""
    fn = getattr(lib().cdll, name)
    rc = fn(*args)
    if rc == -3:
        return False
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib().cdll.dl_last_error().decode()}")
    if _REC is not None:
        _REC.calls.append((fn, args))
    return True


def conv3x3_nt(x, B, H, W, ci, wf, out, co, bias, resid, zero, scratch=None) -> bool:
    """implicit-GEMM 3x3 conv (forward, or data gradient with the rotated shadow); `scratch` (f32; the library splits K into as many
    partial images of B*H*W*co floats as it holds, up to eight) lets the low-resolution levels split K.  False -> use im2col3x3 +
    gemm_nt"""
    assert scratch is None or scratch.dtype == torch.float32
    return _maybe("dl_conv3x3_nt", _p(x), x.stride(0), B, H, W, ci, _p(wf), wf.stride(0), _p(out), out.stride(0), co, _p(bias),
                  _p(resid), resid.stride(0) if resid is not None else 0, _p(zero), _p(scratch),
                  scratch.numel() if scratch is not None else 0, _s())


def conv3x3_wgrad_tn(x, B, H, W, ci, dy, co, g, zero, max_wgs: int = 0) -> bool:
    """implicit-GEMM transposed weight gradient g[(tap, ci), co] +=; max_wgs > 0 caps the persistent workgroups (side-stream
    launches beside the convolution chain); False -> use im2col3x3 + gemm_tn"""
    return _maybe("dl_conv3x3_wgrad_tn", _p(x), x.stride(0), B, H, W, ci, _p(dy), dy.stride(0), dy.shape[0], co, _p(g),
                  g.stride(0), _p(zero), int(max_wgs), _s())


def conv3x3_wgrad_nparts(H: int, W: int, ci: int, co: int, R: int, max_wgs: int = 0) -> int:
    """partial images dl_conv3x3_wgrad_tn_parts writes for this shape and map on this device (0: shape not supported)"""
    return int(lib().cdll.dl_conv3x3_wgrad_tn_nparts(H, W, ci, co, R, int(max_wgs)))


def conv3x3_wgrad_tn_parts(x, B, H, W, ci, dy, co, g, zero, max_wgs: int = 0) -> None:
    """the transposed weight gradient as partial images g[s] f32 [n_parts, 9 ci (padded), co]: plain stores, no atomics; folded in a
    fixed order by ConvFoldTable"""
    _call("dl_conv3x3_wgrad_tn_parts", _p(x), x.stride(0), B, H, W, ci, _p(dy), dy.stride(0), dy.shape[0], co, _p(g), g.stride(1),
          g.stride(0), g.shape[0], _p(zero), int(max_wgs), _s())


def cast_conv3x3_weight(w, wf, wd):
    _call("dl_cast_conv3x3_weight", _p(w), w.shape[0], w.shape[1], _p(wf), wf.stride(0), _p(wd), wd.stride(0), _s())


def conv3x3_wgrad_fold(g, dw):
    _call("dl_conv3x3_wgrad_fold", _p(g), g.stride(0), _p(dw), dw.shape[0], dw.shape[1], _s())


def reduce2x2(x, out, B, Ho, Wo, C, scale):
    _call("dl_reduce2x2", _p(x), _p(out), B, Ho, Wo, C, float(scale), _s())


def expand2x2(x, out, B, Hi, Wi, C, scale):
    _call("dl_expand2x2", _p(x), _p(out), B, Hi, Wi, C, float(scale), _s())


def resample2x2_pair(x0, out0, x1, out1, B, Hs, Ws, C, scale, expand: bool):
    """reduce2x2 / expand2x2 of two tensors of the same geometry (Hs x Ws = the small map) in one launch"""
    _call("dl_resample2x2_pair", _p(x0), _p(out0), _p(x1), _p(out1), B, Hs, Ws, C, float(scale), int(expand), _s())


def pick2x2(x, out, B, Ho, Wo, C):
    _call("dl_pick2x2", _p(x), _p(out), B, Ho, Wo, C, _s())


def stuff2x2(dy, out, B, Hi, Wi, C):
    _call("dl_stuff2x2", _p(dy), _p(out), B, Hi, Wi, C, _s())


def rowbias_add(x, e, out, B, HW, C):
    _call("dl_rowbias_add", _p(x), _p(e), e.stride(0), _p(out), B, HW, C, _s())


def rowbias_bwd(dy, de, B, HW, C):
    _call("dl_rowbias_bwd", _p(dy), _p(de), de.stride(0), B, HW, C, _s())


def attn_small_fwd(q, k, v, out, probs, B, n, H, dh):
    _call("dl_attn_small_fwd", _p(q), _p(k), _p(v), q.stride(0), k.stride(0), _p(out), out.stride(0), _p(probs), B, n, H, dh,
          float(dh) ** -0.5, _s())


def attn_small_bwd(q, k, v, dout, probs, dq, dk, dv, B, n, H, dh):
    assert dq.stride(0) == q.stride(0) and dk.stride(0) == k.stride(0) == dv.stride(0) == v.stride(0)
    _call("dl_attn_small_bwd", _p(q), _p(k), _p(v), q.stride(0), k.stride(0), _p(dout), dout.stride(0), _p(probs), _p(dq),
          _p(dk), _p(dv), B, n, H, dh, float(dh) ** -0.5, _s())


def add_bf16(a, b, out):
    _call("dl_add_bf16", _p(a), _p(b), _p(out), a.numel(), _s())


def copy2d_bf16(src, dst, rows, cols):
    _call("dl_copy2d_bf16", _p(src), src.stride(0), _p(dst), dst.stride(0), rows, cols, _s())


# ------------------------------------------------------------------ fp32-class regime (csrc/f32.hip, dl_f32_*)
class _F32Gemm(ctypes.Structure):
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p), ("lda", ctypes.c_int64),
                ("ldb", ctypes.c_int64), ("ldc", ctypes.c_int64), ("M", ctypes.c_int64), ("N", ctypes.c_int64), ("K", ctypes.c_int64),
                ("trans_a", ctypes.c_int32), ("trans_b", ctypes.c_int32), ("batch1", ctypes.c_int64), ("batch2", ctypes.c_int64),
                ("stride_a1", ctypes.c_int64), ("stride_a2", ctypes.c_int64), ("stride_b1", ctypes.c_int64),
                ("stride_b2", ctypes.c_int64), ("stride_c1", ctypes.c_int64), ("stride_c2", ctypes.c_int64), ("alpha", ctypes.c_float),
                ("bias", ctypes.c_void_p), ("act", ctypes.c_int32), ("pre_out", ctypes.c_void_p), ("accumulate", ctypes.c_int32),
                ("scratch", ctypes.c_void_p), ("scratch_floats", ctypes.c_int64)]


def _addr(t, off: int = 0) -> int:
    """device address of element `off` of an f32 tensor (or of a raw int address)"""
    if isinstance(t, int):
        return t + 4 * off
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError("dl_f32_* ops take f32 device tensors (no CPU fallback exists)")
    return t.data_ptr() + 4 * off


def f32_gemm(A, B, C, M: int, N: int, K: int, *, lda: int, ldb: int, ldc: int, ta: bool = False, tb: bool = False, a_off: int = 0,
             b_off: int = 0, c_off: int = 0, batch: tuple[int, int] = (1, 1), sa=(0, 0), sb=(0, 0), sc=(0, 0), alpha: float = 1.0,
             bias=None, act: int = ACT_NONE, pre_out=None, accumulate: bool = False, scratch=None) -> None:
    """C[M, N] = act(alpha * op(A) op(B) + bias) (+= with accumulate) on the exact-f32 MFMA; ta: A is [K, M]; tb: B is [K, N]
    (default: A [M, K], B [N, K] = torch Linear weight); *_off: element offsets into the tensors (column windows of token rows);
    batch = (b1, b2) with element strides sa / sb / sc per level"""
    d = _F32Gemm(_addr(A, a_off), _addr(B, b_off), _addr(C, c_off), lda, ldb, ldc, M, N, K, int(ta), int(tb), batch[0], batch[1],
                 sa[0], sa[1], sb[0], sb[1], sc[0], sc[1], float(alpha), _p(bias), act, _p(pre_out), int(accumulate),
                 _p(scratch), scratch.numel() if scratch is not None else 0)
    _call("dl_f32_gemm", ctypes.addressof(d), _s())


def f32_linear(x, w, out, *, bias=None, act: int = ACT_NONE, pre_out=None, M: int | None = None, scratch=None) -> None:
    """out[M, out_f] = act(x[M, in_f] w[out_f, in_f]^T + bias)   (nn.Linear forward); scratch: split-K partial images for products
    with few output tiles and a long contraction (the deep UNet levels' convolutions)"""
    M = x.shape[0] if M is None else M
    f32_gemm(x, w, out, M, w.shape[0], w.shape[1], lda=x.stride(0), ldb=w.stride(0), ldc=out.stride(0), bias=bias, act=act,
             pre_out=pre_out, scratch=scratch)


def f32_linear_dgrad(dy, w, dx, *, M: int | None = None, scratch=None) -> None:
    """dx[M, in_f] = dy[M, out_f] w[out_f, in_f]"""
    M = dy.shape[0] if M is None else M
    f32_gemm(dy, w, dx, M, w.shape[1], w.shape[0], lda=dy.stride(0), ldb=w.stride(0), ldc=dx.stride(0), tb=True, scratch=scratch)


def f32_linear_wgrad(dy, x, g, *, R: int | None = None, scratch=None) -> None:
    """g[out_f, in_f] += dy[R, out_f]^T x[R, in_f]"""
    R = dy.shape[0] if R is None else R
    f32_gemm(dy, x, g, g.shape[0], g.shape[1], R, lda=dy.stride(0), ldb=x.stride(0), ldc=g.stride(0), ta=True, tb=True,
             accumulate=True, scratch=scratch)


def f32_ln_modulate_fwd(x, w, b, scale, shift, rows_per_mod, eps, out, mean, rstd, t=None, gate=None, x_out=None):
    M, D = x.shape
    _call("dl_f32_ln_modulate_fwd", _p(x), _p(w), _p(b), _p(scale), _p(shift), scale.stride(0), rows_per_mod, float(eps), _p(out),
          _p(mean), _p(rstd), _p(t), _p(gate), gate.stride(0) if gate is not None else 0, _p(x_out), M, D, _s())


def f32_ln_modulate_bwd(dout, x, w, b, scale, rows_per_mod, mean, rstd, dres, dx, dscale, dshift, dwb_partial, gate_t=None,
                        gate=None, dt=None, dgate=None):
    M, D = x.shape
    _call("dl_f32_ln_modulate_bwd", _p(dout), _p(x), _p(w), _p(b), _p(scale), scale.stride(0), rows_per_mod, _p(mean), _p(rstd),
          _p(dres), _p(dx), _p(dscale), _p(dshift), dscale.stride(0), _p(dwb_partial), _p(gate_t), _p(gate),
          gate.stride(0) if gate is not None else 0, _p(dt), _p(dgate), M, D, _s())


def f32_qk_norm_rope_fwd(qkv, scale_q, scale_k, cos, sin, qk, rrms, B, N, H, dh, rot, eps=1e-6, pos=None):
    """pos: int32 [B*N] rotary table row of every token (SPRINT's kept tokens) or None"""
    _call("dl_f32_qk_norm_rope_fwd", _p(qkv), qkv.stride(0), _p(scale_q), _p(scale_k), _p(cos), _p(sin), _p(qk), _p(rrms), B, N, H, dh,
          rot, float(eps), _p(pos), _s())


def f32_qk_norm_rope_bwd(dqk, qkv, scale_q, scale_k, cos, sin, rrms, dqkv, partials, B, N, H, dh, rot, pos=None):
    _call("dl_f32_qk_norm_rope_bwd", _p(dqk), _p(qkv), qkv.stride(0), _p(scale_q), _p(scale_k), _p(cos), _p(sin), _p(rrms), _p(dqkv),
          dqkv.stride(0), _p(partials), B, N, H, dh, rot, _p(pos), _s())


def f32_gather_tokens(src, idx, dst, B, N, k, D, keep=None):
    """dst[b, j, :] = src[b, idx[b, j], :] on f32 rows: a bytewise row copy, i.e. dl_gather_tokens over 2 D two-byte columns"""
    _call("dl_gather_tokens", _p(src), 2 * src.stride(0), _p(idx), _p(keep), _p(dst), 2 * dst.stride(0), B, N, k, 2 * D, _s())


def f32_scatter_tokens_add(src, idx, dst, B, N, k, D):
    _call("dl_f32_scatter_tokens_add", _p(src), src.stride(0), _p(idx), _p(dst), dst.stride(0), B, N, k, D, _s())


def f32_restore_tokens(xd, inv, mask, out, B, N, k, D):
    _call("dl_f32_restore_tokens", _p(xd), xd.stride(0), _p(inv), _p(mask), _p(out), out.stride(0), B, N, k, D, _s())


def f32_masked_colsum(x, sel, out, R, C, scratch):
    """out[c] += sum over rows with sel[row] < 0 of x[row, c]: partial images per 256-row slab in `scratch`, fixed-order fold"""
    slabs = min(512, (R + 255) // 256)
    assert scratch.numel() >= slabs * C
    _call("dl_f32_masked_colsum_partials", _p(x), x.stride(0), _p(sel), _p(scratch), slabs, R, C, _s())
    reduce_rows_batched_f32(scratch, 0, out, 0, 1, slabs, C)


def f32_ddt_cond_fwd(enc, temb, B, N, out):
    _call("dl_f32_ddt_cond_fwd", _p(enc), enc.stride(0), _p(temb), temb.stride(0), B, N, enc.shape[1], _p(out), _s())


def f32_ddt_cond_bwd(dsz, enc, temb, B, N, denc, dtemb):
    _call("dl_f32_ddt_cond_bwd", _p(dsz), _p(enc), enc.stride(0), _p(temb), temb.stride(0), B, N, enc.shape[1], _p(denc), _p(dtemb), _s())


def f32_gated_residual_fwd(x, t, gate, rows_per_mod, out):
    M, D = x.shape
    _call("dl_f32_gated_residual_fwd", _p(x), _p(t), _p(gate), gate.stride(0), rows_per_mod, _p(out), out.stride(0), M, D, _s())


def f32_gate_bwd(dout, t, gate, rows_per_mod, dt, dgate):
    M, D = dout.shape
    _call("dl_f32_gate_bwd", _p(dout), _p(t), _p(gate), gate.stride(0), rows_per_mod, _p(dt), _p(dgate), dgate.stride(0), M, D, _s())


def f32_attn_fwd(q, k, v, out, probs, B, n, H, dh):
    """AttentionBlock core (unet.py:311-318) over f32 token rows for ANY token count / head width: heads are dh-wide column blocks of
    the rows (addressed by stride in the batched GEMMs); probs f32 [B, H, n, n] = softmax(scale Q K^T) is kept for the backward"""
    sc = float(dh) ** -0.5
    f32_gemm(q, k, probs, n, n, dh, lda=q.stride(0), ldb=k.stride(0), ldc=n, batch=(B, H), sa=(n * q.stride(0), dh),
             sb=(n * k.stride(0), dh), sc=(H * n * n, n * n), alpha=sc)
    f32_softmax_fwd(probs, B * H * n, n)
    f32_gemm(probs, v, out, n, dh, n, lda=n, ldb=v.stride(0), ldc=out.stride(0), tb=True, batch=(B, H), sa=(H * n * n, n * n),
             sb=(n * v.stride(0), dh), sc=(n * out.stride(0), dh))


def f32_attn_bwd(q, k, v, dout, probs, dP, dq, dk, dv, B, n, H, dh):
    """dP: f32 scratch [B, H, n, n] (overwritten: dP, then dS)"""
    sc = float(dh) ** -0.5
    pb, hb = (H * n * n, n * n), dict(batch=(B, H))
    f32_gemm(dout, v, dP, n, n, dh, lda=dout.stride(0), ldb=v.stride(0), ldc=n, sa=(n * dout.stride(0), dh),
             sb=(n * v.stride(0), dh), sc=pb, **hb)                                                   # dP = dO V^T
    f32_gemm(probs, dout, dv, n, dh, n, lda=n, ldb=dout.stride(0), ldc=dv.stride(0), ta=True, tb=True, sa=pb,
             sb=(n * dout.stride(0), dh), sc=(n * dv.stride(0), dh), **hb)                            # dV = P^T dO
    f32_softmax_bwd(probs, dP, B * H * n, n)                                                          # dS over dP
    f32_gemm(dP, k, dq, n, dh, n, lda=n, ldb=k.stride(0), ldc=dq.stride(0), tb=True, sa=pb, sb=(n * k.stride(0), dh),
             sc=(n * dq.stride(0), dh), alpha=sc, **hb)                                               # dQ = scale dS K
    f32_gemm(dP, q, dk, n, dh, n, lda=n, ldb=q.stride(0), ldc=dk.stride(0), ta=True, tb=True, sa=pb,
             sb=(n * q.stride(0), dh), sc=(n * dk.stride(0), dh), alpha=sc, **hb)                     # dK = scale dS^T Q


def f32_softmax_fwd(s, rows, cols):
    _call("dl_f32_softmax_fwd", _p(s), rows, cols, _s())


def f32_softmax_bwd(p, dp, rows, cols):
    _call("dl_f32_softmax_bwd", _p(p), _p(dp), rows, cols, _s())


def f32_swiglu_fwd(u, h):
    _call("dl_f32_swiglu_fwd", _p(u), _p(h), u.shape[0], h.shape[1], _s())


def f32_swiglu_bwd(dh, u, du):
    _call("dl_f32_swiglu_bwd", _p(dh), _p(u), _p(du), u.shape[0], dh.shape[1], _s())


def f32_add(a, b, out):
    _call("dl_f32_add", _p(a), _p(b), _p(out), a.numel(), _s())


def f32_silu_bwd(dy, pre, dx):
    _call("dl_f32_silu_bwd", _p(dy), _p(pre), _p(dx), dy.numel(), _s())


def f32_patchify(x, tok, p, order):
    B, C, H, W = x.shape
    _call("dl_f32_patchify", _p(x), _p(tok), B, C, H, W, p, tok.stride(0), order, _s())


def f32_timestep_embedding(t, out, max_period=10000.0):
    _call("dl_f32_timestep_embedding", _p(t), _p(out), out.shape[0], out.shape[1], float(max_period), _s())


def f32_cond_combine_fwd(e, table, idx, emb, act):
    _call("dl_f32_cond_combine_fwd", _p(e), _p(table), _p(idx), _p(emb), _p(act), e.shape[0], e.shape[1], _s())


def f32_cond_combine_bwd(dact, emb, idx, demb, dtable):
    _call("dl_f32_cond_combine_bwd", _p(dact), _p(emb), _p(idx), _p(demb), _p(dtable), dact.shape[0], dact.shape[1], _s())


# ---- fp32-class regime of the UNet: same call shapes as the bf16 wrappers above (unet_engine_f32.py swaps them in by name)
def f32_im2col3x3(x, cols, B, H, W, C):
    _call("dl_f32_im2col3x3", _p(x), x.stride(0), _p(cols), B, H, W, C, _s())


def f32_col2im3x3(dcols, dx, B, H, W, C):
    _call("dl_f32_col2im3x3", _p(dcols), _p(dx), dx.stride(0), B, H, W, C, _s())


def f32_gn_stats(x, stats, B, HW, C, G=32, eps=1e-5):
    _call("dl_f32_gn_stats", _p(x), _p(stats), B, HW, C, G, float(eps), _s())


def f32_gn_apply_fwd(x, stats, w, b, film_scale, film_shift, silu, out, B, HW, C, G=32):
    _call("dl_f32_gn_apply_fwd", _p(x), _p(stats), _p(w), _p(b), _p(film_scale), _p(film_shift),
          film_scale.stride(0) if film_scale is not None else 0, int(silu), _p(out), B, HW, C, G, _s())


def f32_gn_bwd(dout, x, stats, w, b, film_scale, film_shift, silu, dres, dx, dw_partial, db_partial, dfilm_scale, dfilm_shift, B, HW, C,
               G=32):
    assert dfilm_scale is None or dfilm_scale.stride(0) == film_scale.stride(0)
    _call("dl_f32_gn_bwd", _p(dout), _p(x), _p(stats), _p(w), _p(b), _p(film_scale), _p(film_shift),
          film_scale.stride(0) if film_scale is not None else 0, int(silu), _p(dres), _p(dx), _p(dw_partial), _p(db_partial),
          _p(dfilm_scale), _p(dfilm_shift), B, HW, C, G, _s())


def f32_resample2x2(x, out, B, Hs, Ws, C, scale, mode):
    _call("dl_f32_resample2x2", _p(x), _p(out), B, Hs, Ws, C, float(scale), mode, _s())


def f32_nchw_to_nhwc(x, out, B, C, HW):
    _call("dl_f32_nchw_to_nhwc", _p(x), _p(out), B, C, HW, out.stride(0), _s())


def f32_nhwc_to_nchw(x, out, B, C, HW):
    _call("dl_f32_nhwc_to_nchw", _p(x), _p(out), B, C, HW, x.stride(0), _s())


def f32_copy2d(src, dst, rows, cols):
    _call("dl_f32_copy2d", _p(src), src.stride(0), _p(dst), dst.stride(0), rows, cols, _s())


def f32_rowbias_add(x, e, out, B, HW, C):
    _call("dl_f32_rowbias_add", _p(x), _p(e), e.stride(0), _p(out), B, HW, C, _s())


def f32_rowbias_bwd(dy, de, B, HW, C):
    _call("dl_f32_rowbias_bwd", _p(dy), _p(de), de.stride(0), B, HW, C, _s())


def low_priority_stream(device) -> "torch.cuda.Stream":
    """torch stream object over a HIP stream of the device's lowest priority"""
    handle = ctypes.c_void_p()
    rng = (ctypes.c_int * 2)()
    with torch.cuda.device(device):
        lib().call("dl_stream_create_low_priority", ctypes.byref(handle), rng)
    st = torch.cuda.ExternalStream(handle.value, device=device)
    st.priority_range = (rng[0], rng[1])
    return st


def masked_stream(pattern: str, device) -> "torch.cuda.Stream":
    """torch stream object over a CU-masked HIP stream; pattern "i<k>" enables every k-th CU, "b<n>" the first n CUs"""
    import ctypes

    n_cu = torch.cuda.get_device_properties(device).multi_processor_count
    if pattern[0] == "i":
        bits = [i % int(pattern[1:]) == 0 for i in range(n_cu)]
    elif pattern[0] == "b":
        bits = [i < int(pattern[1:]) for i in range(n_cu)]
    else:
        raise ValueError(f"unknown CU mask pattern {pattern!r}")
    words = (n_cu + 31) // 32
    arr = (ctypes.c_uint32 * words)()
    for i, on in enumerate(bits):
        if on:
            arr[i // 32] |= 1 << (i % 32)
    handle = ctypes.c_void_p()
    with torch.cuda.device(device):
        lib().call("dl_stream_create_masked", arr, words, ctypes.byref(handle))
    return torch.cuda.ExternalStream(handle.value, device=device)
