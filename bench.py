#!/usr/bin/env python
"""Headline benchmark: train images/sec/node, DiT-S/2 rectified-flow training on 256x256-image latents (4x32x32).

    python bench.py                                  # = --gpus 1 --steps 100 --warmup 20; ~1 min with the CPU leg
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full training step of the reference's BaseTrainer.training_step (base_trainer.py:138-153) on one
per-GPU batch of synthetic latents: zero_grad -> draw_timesteps (CPU generator, H2D) -> noise + flow loss head ->
DiT forward -> backward -> [N>1: RCCL gradient all-reduce overlapped with backward] -> AdamW.  Weak scaling: the
per-GPU batch is fixed (256), `value` is the whole-job images/s.  Nothing under /root/reference is read.

Extra objects on the JSON line:
  roofline     -- the DOMINANT kernel of the step = the kernel name with the largest summed launch time in a profiled replay of
                  the same step (HIP events around every GEMM / attention / row-kernel launch, recorded on the stream the launch
                  goes to) -- the kernel `rocprofv3 --kernel-trace --stats` ranks first (profiles/r02_*_kernel_stats.txt):
                  algorithmic FLOPs of its launches / their durations; every other instrumented kernel under `kernels` /
                  `hbm_bound_kernels`; `traffic` from the newest committed PMC pass; the whole-step figure (47.2 GFLOP/img x
                  img/s) as `step_achieved`.
  cpu_baseline -- the CPU oracle (a port of the reference path, oracle/) timed on this host's cores by the protocol of
                  BASELINE.md section 3 / SURVEY 8(d): B=32, 1 warm-up + 3 timed steps, thread count printed; rank 0 at N=1 only.
  dp           -- N>1: rccl_ranks, the exposed (non-overlapped) part of the gradient all-reduce per step and which schedule of the
                  exchange the warm-up measurement kept (overlapped buckets | one exchange after the backward) (`--dp-backend gloo`
                  is the dry mode of this branch for boxes with fewer GPUs than ranks; a GPU test runs it on 2 ranks).
  config.loss_curve_rel_err[_fp32] -- 20 AdamW steps of DiT-S/2 against the reference's own fp32 loss curve (committed fixture): the
                  largest / mean per-step relative error of the bf16 regime (the timed one) and of the fp32 regime (precision_type="no",
                  the reference's default; north_star's 1e-4 bar) -- N=1, outside the timed region.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

# the host driver of this pool only supports dmabuf IPC: without this RCCL's buffer exchange between the ranks' processes fails with
# `hipIpcGetMemHandle: invalid argument`.  It is exported on the boxes already; a launch line that builds its own environment
# (torch.distributed.run --no-python, a scheduler wrapper) may drop it, so it is pinned before the HIP runtime loads.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md; AMD's 5 PF figure is 2:1 sparse)
S2 = dict(simple_dit=True, input_channels=4, output_channels=4, inner_dim=384, embedding_dim=384, num_heads=6,
          mlp_ratio=4, patch_size=2, depth=12, n_classes=1000, classifier_free=True)


def train_flops_per_image(D=384, E=384, L=12, N=256, pC=16) -> float:
    """F_fwd = L(32 N D^2 + 4 N^2 D + 12 E D) + 4 N pC D + 2(256 E + E^2) + 4 E D ; train = 3 F_fwd"""
    fwd = L * (32 * N * D * D + 4 * N * N * D + 12 * E * D) + 4 * N * pC * D + 2 * (256 * E + E * E) + 4 * E * D
    return 3.0 * fwd


def _host_threads() -> int:
    """cores this process may really use: affinity mask and cgroup cpu quota, capped (an oversubscribed torch
    thread pool on a 256-thread host is orders of magnitude slower than a right-sized one)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def cpu_baseline(batch: int = 32, n_timed: int = 3, budget_s: float = 150.0) -> dict:
    """CPU oracle (port of the reference path) by the protocol of BASELINE.md section 3 / SURVEY 8(d): DiT-S/2 flow train steps at
    B=32, one warm-up then 3 timed steps (cut short only if a pathological host exceeds budget_s)."""
    from oracle import diffusion as od
    from oracle import dit as odit
    from oracle import synth

    torch.set_num_threads(_host_threads())
    cfg = odit.DiTConfig()
    P = {k: v.requires_grad_(True) for k, v in synth.dit_params(odit.param_shapes(cfg), seed=7).items()}
    opt = torch.optim.AdamW(list(P.values()), lr=1e-4, weight_decay=0.01)
    x0 = synth.normal("cpu.x0", (batch, 4, 32, 32))
    y = synth.integers("cpu.y", (batch,), 1000)
    times: list[float] = []
    start = time.perf_counter()
    s = 0
    while True:
        noise = synth.normal(f"cpu.n{s}", (batch, 4, 32, 32))
        t = torch.sigmoid(synth.normal(f"cpu.t{s}", (batch,)))
        t0 = time.perf_counter()
        opt.zero_grad()
        pred = odit.dit_forward(P, od.flow_add_noise(x0, t, noise), t, y, cfg)
        od.flow_loss(pred, x0, noise).backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        s += 1
        if s >= 2 and (time.perf_counter() - start > budget_s or s >= 1 + n_timed):
            break
        if s == 1 and times[0] > budget_s:  # pathological host: keep the single (cold) measurement
            break
    timed = times[1:] if len(times) > 1 else times
    dt = sum(timed) / len(timed)
    return {"value": round(batch / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle fp32 DiT-S/2 flow train step, B={batch}, {len(times) - len(timed)} warm-up + {len(timed)} timed steps"}


def loss_curve_rel_err(dev, precision: str = "bf16") -> dict:
    """SURVEY 8(c)(viii) / north_star "loss curve matching CPU reference": 20 AdamW steps of DiT-S/2 (B=4, fixed synthetic data)
    on the HIP path against the curve the REFERENCE itself produced in fp32 (tests/golden/loss_curve.npz, written by
    tests/golden/make_golden.py from /root/reference); returns the largest and the mean per-step relative error.  The oracle
    package only supplies the seeded input generator here (checker use, outside the timed region)."""
    import numpy as np

    from diffulab_amd import Diffuser, MMDiT
    from diffulab_amd.training import FusedAdamW
    from oracle import dit as odit
    from oracle import synth

    ref = np.load(os.path.join(ROOT, "tests", "golden", "loss_curve.npz"))["losses"]
    m = MMDiT(**S2)
    m.load_state_dict(synth.dit_params(odit.param_shapes(odit.DiTConfig()), seed=7))
    m.set_precision(precision)  # "bf16": the timed regime; "fp32": precision_type="no", the reference's default (engine_f32.py)
    m = m.to(dev)
    opt = FusedAdamW(m.parameters(), lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8)
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=50)
    B = 4
    x0, y = synth.normal("curve.x0", (B, 4, 32, 32)).to(dev), synth.integers("curve.y", (B,), 1000).to(dev)
    got = []
    for s in range(len(ref)):
        noise = synth.normal(f"curve.noise{s}", (B, 4, 32, 32)).to(dev)
        t = synth.uniform(f"curve.t{s}", (B,), lo=0.02, hi=0.98)
        opt.zero_grad()
        loss = d.compute_loss({"x": x0, "y": y, "p": 0.0}, timesteps=t, noise=noise)["loss"]
        loss.backward()
        opt.step()
        got.append(loss.item())
    err = np.abs(np.array(got) - ref) / ref
    out = {"steps": len(ref), "max": float(f"{err.max():.3e}"), "mean": float(f"{err.mean():.3e}"),
           "against": "the reference's fp32 curve (tests/golden/loss_curve.npz)", "compute": precision}
    if precision == "bf16":
        # what the REFERENCE loses on the same loop under torch.autocast("cpu", bfloat16) against its own fp32 curve
        # (tests/golden/loss_curve_autocast.npz, make_golden.py curve_autocast): the yardstick of the bf16 regime
        ra = np.load(os.path.join(ROOT, "tests", "golden", "loss_curve_autocast.npz"))["rel_err_vs_fp32"]
        out["reference_under_autocast"] = {"max": float(f"{ra.max():.3e}"), "mean": float(f"{ra.mean():.3e}"),
                                           "hip_max_over_reference_max": float(f"{err.max() / ra.max():.3f}"),
                                           "hip_mean_over_reference_mean": float(f"{err.mean() / ra.mean():.3f}")}
    return out


def fp32_regime_step(dev, B: int, flops_per_image: float) -> dict:
    """the same training step at the same batch in the fp32-class regime (precision_type="no", the reference's default): a reported
    side figure -- the timed region and `value` stay the bf16 regime's"""
    from diffulab_amd import Diffuser, MMDiT
    from diffulab_amd.training import FusedAdamW

    torch.manual_seed(1)
    m = MMDiT(**S2)
    m.set_precision("fp32")
    m = m.to(dev)
    opt = FusedAdamW(m.parameters(), lr=1e-4, weight_decay=0.01)
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=50, extra_args={"logits_normal": True})
    x0, y = torch.randn(B, 4, 32, 32, device=dev), torch.randint(0, 1000, (B,), device=dev)

    def step():
        opt.zero_grad()
        d.compute_loss({"x": x0, "y": y, "p": 0.1}, timesteps=d.draw_timesteps(B).to(dev))["loss"].backward()
        opt.step()

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    return {"ms_per_step": round(ms, 2), "images_per_s": round(B / ms * 1e3, 1), "tflops": round(B * flops_per_image / ms / 1e9, 1),
            "peak_tflops_f32_mfma": 157.3, "steps": n, "engine": type(m.engine).__name__}


def _tn_variant(R: int, M: int, N: int) -> str:
    """which kernel dl_gemm_tn dispatches to (csrc/gemm.hip dl_gemm_tn_ex, default variant)"""
    tiles_m = -(-M // 384)
    if M % 384 == 0 and N % 192 == 0 and R // 32 >= 64:
        return "gemm_tn_w4_k"
    if N % 128 == 0 and R // 64 >= 64 and 5 * M >= 3 * tiles_m * 384:
        return "gemm_tn_big_k"
    return "gemm_tn_k"


def _nt_variant(M: int, N: int, K: int, plain: bool) -> str:
    """which kernel dl_gemm_nt dispatches to (csrc/gemm.hip dispatch_big, default variant)"""
    if M % 256 == 0:
        if plain and N % 384 == 0 and (M // 256) * (N // 384) >= 64:
            return "gemm_nt_big_k<384,2,0>"
        if N % 192 == 0 and (M // 256) * (N // 192) >= 64:
            return "gemm_nt_big_k<192,2,%d>" % (0 if plain else 1)
    return "gemm_nt_k"


def roofline_replay(step, model, images_per_s_per_gpu: float) -> dict:
    """Profiled replay of the same training step: HIP events around every MFMA GEMM launch, recorded on the stream the
    launch goes to (the wgrad GEMMs run on the engine's side stream).  The dominant kernel is the NT GEMM family
    `gemm_nt_big_k` (every linear's forward and data-gradient: 2/3 of the step's FLOPs, all on the critical path);
    `achieved` = algorithmic FLOPs of its launches (2 M N K each) / their summed launch durations.  `traffic` is the
    HBM bytes per launch of the same kernels from the committed PMC passes (profiles/r01_m_pmc_traffic.json:
    FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, calibrated on adamw_k), null if that file is absent."""
    from diffulab_amd import ops

    rec: list[tuple[str, torch.cuda.Event, torch.cuda.Event, float]] = []
    tn_shapes: dict[tuple[str, int, int, int], int] = {}  # weight-gradient launches of the replay: (kernel, R, M, N) -> count
    last_group: list = []  # [probs, slab] of the last grouped weight-gradient launch
    orig = {n: getattr(ops, n) for n in ("gemm_nt", "gemm_nt_swiglu", "gemm_tn", "attn_fwd_qkv", "attn_bwd_qkv", "attn_fwd", "attn_bwd",
                                         "attn_bwd_tok", "mlp_dswiglu_recompute", "ln_modulate_gemm_fwd", "ln_modulate_gemm_bwd",
                                         "gemm_nt_qk_norm_rope", "gemm_tn_group", "gemm_nt_ssq", "attn_fwd_qkn")}

    def timed(kind: str):
        fn = orig[kind]

        def wrapper(a, b, *rest, **kw):
            if kind == "gemm_nt":
                M, N, K = kw.get("M") or a.shape[0], kw.get("N") or b.shape[0], kw.get("K") or a.shape[1]
                plain = not any(kw.get(k) is not None for k in ("bias", "pre_out", "resid")) and not kw.get("act") \
                    and rest[0].dtype == torch.bfloat16
                name, fl = _nt_variant(M, N, K, plain), 2.0 * M * N * K
            elif kind == "gemm_nt_swiglu":
                name, fl = "gemm_nt_big_k<384,2,2>", 2.0 * a.shape[0] * b.shape[0] * a.shape[1]
            elif kind in ("ln_modulate_gemm_fwd", "ln_modulate_gemm_bwd"):  # (a [M, K], w [384, K], ...): row-complete 256x384 tiles
                K = kw.get("K") or a.shape[1]
                name, fl = "gemm_nt_rows_k<%d>" % (0 if kind.endswith("fwd") else 1), 2.0 * a.shape[0] * 384 * K
            elif kind == "gemm_nt_ssq":  # (a [M, K], w [N, K], out, ssq): the qkv GEMM + QK-norm row statistics
                name, fl = "gemm_nt_big_k<384,2,3>", 2.0 * a.shape[0] * b.shape[0] * a.shape[1]
            elif kind == "attn_fwd_qkn":  # (qkv, ssq, sq, sk, cos, sin, q, k, rrms, out, lse, B, H, N, dh, rot, scale)
                Bq, Hq, Nq, dq = rest[9], rest[10], rest[11], rest[12]
                # (from three (sample, head) items per CU the entry point runs the persistent + pipelined form, csrc/attention.hip)
                pipe = Nq == 256 and Bq * Hq >= 3 * torch.cuda.get_device_properties(0).multi_processor_count
                name, fl = ("attn_fwd_qkn_pipe_k" if pipe else "attn_fwd_qkn_k"), 4.0 * Bq * Hq * Nq * Nq * dq
            elif kind == "gemm_nt_qk_norm_rope":
                name, fl = "gemm_nt_rows_k<2>", 2.0 * a.shape[0] * b.shape[0] * a.shape[1]
            elif kind == "gemm_tn_group":  # (probs = [(dy, x, g), ...], slab): one launch + the fold
                name, fl = "gemm_tn_group_k", sum(2.0 * dy.shape[0] * g.shape[0] * g.shape[1] for dy, _, g in a)
                tn_shapes[(name, a[0][0].shape[0], tuple((g.shape[0], g.shape[1]) for _, _, g in a), 0)] = \
                    tn_shapes.get((name, a[0][0].shape[0], tuple((g.shape[0], g.shape[1]) for _, _, g in a), 0), 0) + 1
                last_group[:] = [list(a), b]  # the operands of this launch stay valid in the workspace: re-timed alone below
            elif kind == "mlp_dswiglu_recompute":  # (x, wp, dt, w2t, du): u tile recomputed (2 M 2F K) + dh tile (2 M F K)
                name, fl = "mlp_dswiglu_rc_k", 6.0 * a.shape[0] * rest[1].shape[0] * a.shape[1]
            elif kind.startswith("attn_"):  # (q, k, ..., B, H, N, dh, scale): 4 N^2 dh per head forward, 10 N^2 dh backward
                Bq, Hq, Nq, dq = a.shape
                name, fl = ("attn_bwd_k" if "bwd" in kind else "attn_fwd_k"), (10.0 if "bwd" in kind else 4.0) * Bq * Hq * Nq * Nq * dq
            else:
                M, N = kw.get("M") or a.shape[1], kw.get("N") or b.shape[1]
                name, fl = _tn_variant(a.shape[0], M, N), 2.0 * a.shape[0] * M * N
                tn_shapes[(name, a.shape[0], M, N)] = tn_shapes.get((name, a.shape[0], M, N), 0) + 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()  # current stream == the stream the launch goes to
            r = fn(a, b, *rest, **kw)
            e1.record()
            rec.append((name, e0, e1, fl))
            return r

        return wrapper

    # the HBM-bound row kernels of the same step: algorithmic bytes (every bf16 operand read or written once) / launch time
    hbm_rec: list[tuple[str, torch.cuda.Event, torch.cuda.Event, float]] = []

    def nbytes(*ts) -> float:
        return float(sum(t.numel() * t.element_size() for t in ts if t is not None))

    row_bytes = {
        "swiglu_bwd": lambda dh, u, du: nbytes(dh, u, du),
        "ln_modulate_fwd": lambda x, w, b, sc, sh, rpm, eps, out, mean, rstd, t=None, gate=None, x_out=None: nbytes(x, out, t, x_out),
        "ln_modulate_bwd": lambda dout, x, w, b, sc, rpm, mean, rstd, dres, dx, dsc, dsh, dwb, gate_t=None, gate=None, dt=None,
        dgate=None: nbytes(dout, x, dres, dx, gate_t, dt),
        # (v / dv None: V stays inside the qkv rows and only the q and k thirds of qkv / dqkv are touched)
        "qk_norm_rope_fwd": lambda qkv, sq, sk, cos, sin, q, k, v, *a, **kw: nbytes(q, k, v) + nbytes(qkv) * (1.0 if v is not None else 2 / 3),
        "qk_norm_rope_bwd": lambda dq, dk, dv, qkv, sq, sk, cos, sin, rrms, dqkv, *a, **kw: nbytes(dq, dk, dv)
        + nbytes(qkv, dqkv) * (1.0 if dv is not None else 2 / 3),
    }
    # (round 3: the in-place QK-norm backward reads the q, k thirds of qkv and of dqkv and writes the latter back)
    row_bytes["qk_norm_rope_bwd_inplace"] = lambda qkv, sq, sk, cos, sin, rrms, dqkv, *a, **kw: nbytes(qkv) * 2 / 3 + nbytes(dqkv) * 4 / 3
    orig_rows = {n: getattr(ops, n) for n in row_bytes}

    def timed_row(kind: str):
        fn, fb = orig_rows[kind], row_bytes[kind]

        def wrapper(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **kw)
            e1.record()
            hbm_rec.append((kind, e0, e1, fb(*a, **kw)))
            return r

        return wrapper

    for n in orig:
        setattr(ops, n, timed(n))
    for n in orig_rows:
        setattr(ops, n, timed_row(n))
    saved_reducer, model.engine.reducer = model.engine.reducer, None
    # the timed region issues each block through the native driver (dl_dit_block_fwd / _bwd); the replay issues the SAME launches in
    # the same order and on the same streams from Python so that every one can be bracketed with events here
    saved_native = os.environ.get("DL_NATIVE_BLOCK")
    os.environ["DL_NATIVE_BLOCK"] = "0"
    reps = 2
    try:
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
    finally:
        for n, f in list(orig.items()) + list(orig_rows.items()):
            setattr(ops, n, f)
        model.engine.reducer = saved_reducer
        if saved_native is None:
            del os.environ["DL_NATIVE_BLOCK"]
        else:
            os.environ["DL_NATIVE_BLOCK"] = saved_native
    per: dict[str, list[float]] = {}
    for name, e0, e1, fl in rec:
        d = per.setdefault(name, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1)
        d[2] += fl
    kernels = {k: {"launches_per_step": v[0] // reps, "avg_launch_us": round(v[1] * 1e3 / v[0], 2),
                   "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1)} for k, v in sorted(per.items())}
    hb: dict[str, list[float]] = {}
    for name, e0, e1, nb in hbm_rec:
        d = hb.setdefault(name, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1)
        d[2] += nb
    # dominant kernel = largest summed launch time over every instrumented kernel name (the MFMA kernels lead this workload;
    # the HBM-bound row kernels are reported beside it)
    dom = max(per, key=lambda k: per[k][1])
    n_l, ms, fl = per[dom]
    ach = fl / (ms * 1e-3) / 1e12
    traffic, tsrc = None, None
    tfiles = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json"))
    if tfiles:
        tsrc = tfiles[-1]
        t = json.load(open(os.path.join(ROOT, "profiles", tsrc)))
        key = dom.replace(",", ", ")  # rocprof prints template arguments with a space
        rows = [v for k, v in t.items() if k.startswith(key)]
        if rows:
            traffic = round(sum(v["launches"] * (v["fetch_MB"] + v["write_MB"]) for v in rows) * 1e6
                            / sum(v["launches"] for v in rows))
    if tfiles:  # both roofline fractions per kernel: HBM bytes per launch of the committed PMC pass over this run's launch time
        for k, v in kernels.items():
            rows = [r for n, r in t.items() if n.startswith(k.replace(",", ", "))]
            if rows:
                mb = sum(r["launches"] * (r["fetch_MB"] + r["write_MB"]) for r in rows) / sum(r["launches"] for r in rows)
                v["hbm_GBps"] = round(mb * 1e6 / (v["avg_launch_us"] * 1e-6) / 1e9, 1)
                v["hbm_frac_of_8TBps"] = round(v["hbm_GBps"] / 8000.0, 3)
                v["mfma_frac"] = round(v["tflops"] / PEAK_BF16_TFLOPS, 3)
    hbm_kernels = {k: {"launches_per_step": v[0] // reps, "avg_launch_us": round(v[1] * 1e3 / v[0], 2),
                       "achieved_GBps": round(v[2] / (v[1] * 1e-3) / 1e9, 1), "frac_of_8TBps": round(v[2] / (v[1] * 1e-3) / 8e12, 3)}
                   for k, v in sorted(hb.items())}
    # The weight-gradient kernel runs on the engine's SIDE stream, capped at half the CUs and beside the main chain (it is off the
    # critical path: the step time is the main chain's).  `achieved` above is that in-step figure; the same launches alone on the
    # whole chip (same shapes and counts, uncapped, nothing else resident) are reported next to it.
    alone = None
    if dom == "gemm_tn_group_k" and tn_shapes and last_group:
        # the operands of the step itself (a block's activations and their gradients, still in the workspace): the chip's clock --
        # and with it the rate -- depends on the data (random normal operands toggle more bits and run ~25 % slower)
        t_ms, t_fl = 0.0, 0.0
        for (name, R, shapes, _), cnt in tn_shapes.items():
            if name != dom:
                continue
            probs = [(dy, x, torch.zeros_like(g)) for dy, x, g in last_group[0]]
            slab = last_group[1]
            for _ in range(3):
                ops.gemm_tn_group(probs, slab)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                ops.gemm_tn_group(probs, slab)
            e1.record()
            torch.cuda.synchronize()
            t_ms += e0.elapsed_time(e1) / 10 * cnt
            t_fl += sum(2.0 * R * M * N for M, N in shapes) * cnt
        if t_ms > 0:
            alone = {"achieved": round(t_fl / (t_ms * 1e-3) / 1e12, 1), "frac": round(t_fl / (t_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                     "avg_launch_us": round(t_ms * 1e3 / sum(c for (n, *_), c in tn_shapes.items() if n == dom), 2),
                     "note": "the same launch (+ its fold) alone on the whole chip, on the step's own operands"}
    elif dom.startswith("gemm_tn") and tn_shapes:
        t_ms, t_fl = 0.0, 0.0
        for (name, R, M, N), cnt in tn_shapes.items():
            if name != dom:
                continue
            a = torch.randn(R, M, device="cuda").to(torch.bfloat16)
            b = torch.randn(R, N, device="cuda").to(torch.bfloat16)
            c = torch.zeros(M, N, device="cuda")
            for _ in range(3):
                ops.gemm_tn(a, b, c)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                ops.gemm_tn(a, b, c)
            e1.record()
            torch.cuda.synchronize()
            t_ms += e0.elapsed_time(e1) / 10 * cnt
            t_fl += 2.0 * R * M * N * cnt
        if t_ms > 0:
            alone = {"achieved": round(t_fl / (t_ms * 1e-3) / 1e12, 1), "frac": round(t_fl / (t_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                     "avg_launch_us": round(t_ms * 1e3 / sum(c for (n, *_), c in tn_shapes.items() if n == dom), 2),
                     "note": "same launches alone on the whole chip; in the step they share it with the main chain on a side stream"}
    step_ach = images_per_s_per_gpu * train_flops_per_image() / 1e12
    # The two floors of THIS dataflow and where the step stands against the larger one (the step is the SUM of its
    # kernels' MFMA and HBM phases; max(floors) is what a perfectly overlapped execution of the same bytes and FLOPs would take):
    #   hbm_floor_ms  = HBM bytes per step of the newest committed PMC pass (every kernel; the trace's step count comes from its
    #                   adamw_k launches; fills of the model build excluded) / 6.3 TB/s (achievable of the 8 TB/s peak)
    #   mfma_floor_ms = MFMA FLOPs the step EXECUTES (instrumented launches, recomputed MLP pre-activations included) / 2.5 PF/s
    floors = None
    if tfiles:
        steps_in_trace = max(1, sum(v["launches"] for k_, v in t.items() if k_.startswith("adamw_k")))
        gb = sum(v["launches"] * (v["fetch_MB"] + v["write_MB"]) for k_, v in t.items()
                 if not k_.startswith("at::native") and "FillFunctor" not in k_) / 1e3 / steps_in_trace
        exec_tf = sum(v[2] for v in per.values()) / reps / 1e12
        step_ms = 1e3 * model.engine.geo[0] / images_per_s_per_gpu
        hbm_ms, mfma_ms = gb / 6.3, exec_tf / PEAK_BF16_TFLOPS * 1e3
        floors = {"hbm_GB_per_step": round(gb, 1), "hbm_floor_ms": round(hbm_ms, 2), "executed_TFLOP_per_step": round(exec_tf, 2),
                  "mfma_floor_ms": round(mfma_ms, 2), "step_ms": round(step_ms, 2),
                  "frac_of_max_floor": round(max(hbm_ms, mfma_ms) / step_ms, 3),
                  "sum_of_floors_ms": round(hbm_ms + mfma_ms, 2)}
    label = dom + (" (+ tn_group_fold_k: one C call)" if dom == "gemm_tn_group_k" else "")
    return {"bound": "mfma", "kernel": label + " (largest summed launch time of the step's instrumented launches)",
            "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
            "traffic": traffic, "traffic_unit": f"bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/{tsrc})",
            "flops_per_launch": round(fl / n_l), "launches_per_step": n_l // reps, "avg_launch_us": round(ms * 1e3 / n_l, 2),
            "ms_per_step": round(ms / reps, 3), "kernels": kernels,
            "step_achieved": round(step_ach, 1), "step_frac": round(step_ach / PEAK_BF16_TFLOPS, 4),
            "hbm_bound_kernels": hbm_kernels, "dominant_kernel_alone": alone, "floors": floors}


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without an external launcher: the parent starts N fresh child processes of this script (one rank
    per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), relays their output -- rank 0 prints the JSON
    line -- and returns the worst exit status.  The parent has not initialised the HIP runtime (importing torch does not) and
    replaces no process image: the ranks are ordinary children (accelerate's `split_batches=True` launch, trainers/common.py:103-109,
    is one process per GPU as well)."""
    import socket
    import subprocess

    with socket.socket() as s:  # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    codes: list[int] = []
    try:
        # poll ALL ranks: whichever dies first (device init, out of memory) takes the others down at once -- they would sit in
        # init_process_group / a collective until the store timeout otherwise
        live = list(procs)
        while live:
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                codes.append(rc)
                if rc != 0:
                    for q in live:
                        q.terminate()
            if live:
                time.sleep(0.05)
    except KeyboardInterrupt:
        for q in procs:
            if q.poll() is None:
                q.terminate()
        raise
    return max((abs(c) for c in codes), default=1)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # 100 timed steps after 20 warm-up steps (2.2 s + 0.4 s)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dp-backend", choices=("nccl", "gloo"), default="nccl", help="nccl (= RCCL, one rank per GPU) is the product "
                    "path; gloo is the DRY MODE of the data-parallel branch for boxes with fewer GPUs than ranks: the ranks share the "
                    "visible GPUs and exchange through gloo, so the reducer, the overlap instrumentation and the `dp` object run "
                    "(tests/test_trainer_gpu.py); its throughput number means nothing")
    ap.add_argument("--trainer-mode", action="store_true", help="secondary number (N = 1): the timed step IS BaseTrainer.training_step "
                    "(loss meter, fused EMA) instead of the bare zero_grad / loss / backward / AdamW sequence")
    ap.add_argument("--launch-check", action="store_true", help="print this rank's launch environment and exit before any GPU "
                    "call (tests/test_host_logic.py checks the self-launch path on the CPU container with it)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))  # plain `python bench.py --gpus N`: this process becomes the launcher (never touches a GPU)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run or without WORLD_SIZE"
    if args.launch_check:
        print(json.dumps({"rank": rank, "local_rank": local, "world": world, "master": os.environ.get("MASTER_ADDR"),
                          "port": os.environ.get("MASTER_PORT"), "cuda_initialized": torch.cuda.is_initialized()}), flush=True)
        return
    if args.dp_backend == "gloo":
        local %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        from diffulab_amd.training.dp import configure_rccl_env

        configure_rccl_env()  # RCCL's channel workgroups fit into the CUs the persistent grids leave free
        if args.dp_backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from diffulab_amd import Diffuser, MMDiT, ops
    from diffulab_amd.training import FusedAdamW
    from diffulab_amd.training.dp import GradReducer, broadcast_arena

    torch.manual_seed(1234 + rank)
    model = MMDiT(**S2).to(dev)
    model.flatten_parameters()
    broadcast_arena(model._flat)
    diffuser = Diffuser(model, sampling_method="euler", model_type="rectified_flow", n_steps=50,
                        extra_args={"logits_normal": True})
    opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8)
    reducer = GradReducer(model._flat_grad)
    model.engine.reducer = reducer if world > 1 else None
    opt.grad_scale = reducer.grad_scale

    B = args.batch
    gen = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x_data = torch.randn(B, 4, 32, 32, generator=gen).to(dev)  # synthetic ImageNet-256 SD-VAE-f8 shaped latents
    y_data = torch.randint(0, 1000, (B,), generator=gen).to(dev)
    loss_host = torch.zeros((), pin_memory=True)

    def step() -> None:
        opt.zero_grad()
        t = diffuser.draw_timesteps(B).to(dev, non_blocking=True)  # CPU generator draw + H2D, as the reference
        losses = diffuser.compute_loss({"x": x_data, "y": y_data, "p": 0.1}, timesteps=t)
        loss_host.copy_(losses["loss"].detach(), non_blocking=True)  # tracker read-back without a per-step sync
        sum(losses.values()).backward()
        opt.step()

    ema = None
    if args.trainer_mode and world == 1:
        # the step as the drop-in trainer runs it: BaseTrainer.training_step itself (zero_grad -> CPU timestep draw -> loss heads ->
        # meter update -> backward -> optimizer step -> EMA), on a batch that is already on the device (a prefetching loader)
        import tempfile

        from diffulab_amd.training import EMA, AverageMeter, BaseTrainer

        trainer = BaseTrainer(n_epoch=1, precision_type="bf16", save_path=tempfile.mkdtemp(prefix="bench_trainer_"), use_ema=True)
        ema = EMA(model, beta=0.999, update_after_step=0, update_every=10)
        meter = AverageMeter()
        batch = {"model_inputs": {"x": x_data, "y": y_data}, "extra": {}}

        def step() -> None:  # noqa: F811
            trainer.training_step(diffuser=diffuser, optimizer=opt, batch=batch, tracker=meter, p_classifier_free_guidance=0.1,
                                  ema_denoiser=ema)

    def sync() -> None:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    extra_warmup = 0
    while world > 1 and reducer.tuning and extra_warmup < 32:
        # the reducer measures its two schedules on the first 14 steps (training/dp.py): with a shorter --warmup the decision would fall
        # into the timed region, so the untimed part is extended until it is taken (same count on every rank)
        step()
        extra_warmup += 1
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    ms = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    final_loss = float(loss_host)

    dp = None
    if world > 1:  # a few extra steps with the reducer's wait instrumented: how much of the all-reduce is NOT hidden under backward
        reducer.measure = True
        for _ in range(3):
            step()
        tail = reducer.exposed_ms()
        reducer.measure = False
        assert dist.get_world_size() == args.gpus
        dp = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "grad_bytes_per_step": model._flat_grad.numel() * 4,
              "exposed_allreduce_ms_per_step": None if tail is None else round(tail, 3),
              # how the exchange was scheduled in the timed region: decided by measurement during warm-up steps 2-13 (training/dp.py)
              # (a pinned DIFFULAB_DP_OVERLAP is honoured from the first step; DIFFULAB_DP_MEASURE=1 opts into timing both anyway)
              "exchange": reducer.tuned or {"mode": "overlapped" if reducer.overlap else "after_backward",
                                            "decided": "DIFFULAB_DP_OVERLAP pinned without DIFFULAB_DP_MEASURE=1 (or fewer warm-up steps than the "
                                                       "measurement needs)"},
              "untimed_steps_before_timing": args.warmup + extra_warmup}

    roof = None
    if rank == 0 and not args.no_roofline:
        roof = roofline_replay(step, model, value / world)

    cpu, curve, curve32, fp32_step = None, None, None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
        curve = loss_curve_rel_err(dev, "bf16")
        curve32 = loss_curve_rel_err(dev, "fp32")
        fp32_step = fp32_regime_step(dev, B, train_flops_per_image())

    if rank == 0:
        out = {
            "metric": "train images/sec/node (DiT-S/2 flow-matching, 256x256 latents)",
            "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            # (< 120 characters: the driver's parser cuts longer strings and drops the keys behind them)
            "config": {"workload": "DiT-S/2 384/6h/12L p2 39.9M rectified-flow train step, 4x32x32 latents, AdamW, p_drop 0.1",
                       "global_batch": world * B, "per_gpu_batch": B, "tokens_per_image": 256,
                       "parallelism": f"dp{world}", "flops_per_image": train_flops_per_image(), "final_loss": final_loss,
                       "loss_curve_rel_err": curve, "loss_curve_rel_err_fp32": curve32, "fp32_regime_step": fp32_step},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if dp is not None:
            out["dp"] = dp
            if args.dp_backend == "gloo":
                out["data"] = "synthetic (DRY MODE of the dp branch: ranks share GPUs, gloo exchange -- not a throughput measurement)"
        if args.trainer_mode:
            out["config"]["trainer_mode"] = "BaseTrainer.training_step itself (loss meter, EMA(update_every=10)) on a device-resident batch"
        print(json.dumps(out))
    if world > 1:
        dist.barrier()  # rank 0 was still replaying / printing: tear the communicator down together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
