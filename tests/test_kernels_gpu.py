"""GPU parity tests, kernel by kernel, THROUGH THE C ABI (diffulab_amd.ops -> libdiffulab_hip.so).

Each HIP kernel is compared with the CPU oracle (oracle/) or, for plain linear algebra, with an fp32 torch
evaluation of the same bf16-rounded inputs.  Tolerances (stated per test):
  * f32 elementwise heads / sampler steps: 1e-6 relative (fma contraction only);
  * bf16-output kernels: relative L2 <= 4e-3 (one bf16 rounding of the result, 2^-9 per element);
  * integer / index behaviour (label gather, timestep gather): exact.
"""

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import diffusion as od  # noqa: E402
from oracle import dit as odit  # noqa: E402
from oracle import synth  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    import ctypes

    from diffulab_amd import _lib, ops as _ops

    assert _lib.available(), "libdiffulab_hip.so missing on the GPU box"
    L = _lib.lib()
    arch = ctypes.create_string_buffer(64)
    cu, lds, hbm = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
    L.call("dl_device_info", 0, ctypes.byref(cu), ctypes.byref(lds), ctypes.byref(hbm), arch, 64)
    print("device:", arch.value.decode(), cu.value, "CUs", lds.value, "B LDS", hbm.value / 2**30, "GiB")
    assert arch.value.decode().startswith("gfx950")
    return _ops


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def bf(x):  # bf16-rounded fp32 copy on CPU
    return x.to(torch.bfloat16).to(torch.float32)


def dev_bf(x):
    return x.to(torch.bfloat16).to(DEV).contiguous()


# ------------------------------------------------------------------ hardware semantics the kernels rely on
def test_probe_tr16_lane_map(probe_lib):
    """ds_read_b64_tr_b16: inside each 16-lane group, lane i receives element (i%4) of the 8 bytes addressed
    by lane 4j + i/4, for j = 0..3 (a 4x16 transpose).  gemm_tn and attention are built on this."""
    out = torch.zeros(256, dtype=torch.int16, device=DEV)
    assert probe_lib.dl_probe_tr16(out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    got = out.cpu().numpy().astype(np.int64).reshape(64, 4)
    os.makedirs("gpurun_out", exist_ok=True)
    np.savetxt("gpurun_out/probe_tr16.txt", got, fmt="%d")
    exp = np.zeros((64, 4), dtype=np.int64)
    for lane in range(64):
        g, i = lane // 16, lane % 16
        for j in range(4):
            src_lane = g * 16 + 4 * j + i // 4
            exp[lane, j] = src_lane * 4 + (i % 4)
    assert np.array_equal(got, exp), got[:20]


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(256, 384, 384), (300, 1152, 384), (64, 16, 384), (2, 384, 256), (512, 384, 1536),
                                   (1024, 3072, 384), (130, 72, 64), (8192, 384, 1536), (16384, 1152, 384), (12288, 384, 384), (16384, 384, 768),
                                   (256, 384, 28416), (64, 136, 4096)])  # the last two: split-K path of the f32 output
def test_gemm_nt_plain(ops, M, N, K):
    a = synth.normal(f"nt.a{M}", (M, K))
    b = synth.normal(f"nt.b{N}", (N, K), std=K**-0.5)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm_nt(dev_bf(a), dev_bf(b), out)
    ref = bf(a) @ bf(b).t()
    assert rel(out.float(), ref) < 4e-3
    out32 = torch.empty(M, N, device=DEV, dtype=torch.float32)
    ops.gemm_nt(dev_bf(a), dev_bf(b), out32)
    assert rel(out32, ref) < 2e-5  # f32 accumulate, order differs only


@pytest.mark.parametrize("M", [512, 8192, 16384])  # 8192 rows: persistent 256x192 kernel; 16384 rows: 256x384 tiles
def test_gemm_nt_epilogues(ops, M):
    N, K, rows = 384, 384, 128
    a, b = synth.normal("ep.a", (M, K)), synth.normal("ep.b", (N, K), std=K**-0.5)
    bias = synth.normal("ep.bias", (N,))
    resid = synth.normal("ep.res", (M, N))
    gate = synth.normal("ep.gate", (M // rows, 2 * N))  # strided gate rows (ld = 2N)
    acc = bf(a) @ bf(b).t() + bias
    # bias + silu + pre_out
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    pre = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ops.gemm_nt(dev_bf(a), dev_bf(b), out, bias=bias.to(DEV), act=ops.ACT_SILU, pre_out=pre)
    assert rel(pre.float(), acc) < 4e-3
    assert rel(out.float(), odit.silu(acc)) < 4e-3
    # gated residual
    g_dev = dev_bf(gate)
    ops.gemm_nt(dev_bf(a), dev_bf(b), out, resid=dev_bf(resid), gate=g_dev[:, N:], rows_per_gate=rows)
    ref = bf(resid) + bf(gate)[:, N:].repeat_interleave(rows, 0) * (bf(a) @ bf(b).t())
    assert rel(out.float(), ref) < 4e-3
    # plain residual (no gate), f32 out
    o32 = torch.empty(M, N, device=DEV, dtype=torch.float32)
    ops.gemm_nt(dev_bf(a), dev_bf(b), o32, resid=dev_bf(resid))
    assert rel(o32, bf(resid) + bf(a) @ bf(b).t()) < 2e-5


@pytest.mark.parametrize("R,M,N", [(1024, 384, 1152), (128, 16, 384), (256, 2304, 64), (4096, 384, 384), (64, 136, 264), (8192, 1152, 384), (16384, 384, 1536),
                                   (8192, 512, 512), (4096, 4096, 512), (8192, 1024, 128), (4096, 640, 2560)])  # ragged last 384-row m-tile
def test_gemm_tn(ops, R, M, N):
    a = synth.normal(f"tn.a{R}{M}", (R, M))
    b = synth.normal(f"tn.b{R}{N}", (R, N))
    init = synth.normal(f"tn.c{M}{N}", (M, N))
    c = init.to(DEV).clone()
    ops.gemm_tn(dev_bf(a), dev_bf(b), c)
    ref = init + bf(a).t() @ bf(b)
    assert rel(c, ref) < 2e-5


@pytest.mark.parametrize("R,M,N,cap", [(8192, 1152, 384, 0), (8192, 384, 384, 128), (8192, 3072, 384, 128), (8192, 384, 1536, 0),
                                         (2048 + 7 * 64, 768, 192, 64), (4096 + 64, 384, 576, 256)])
def test_gemm_tn_ring_kernel(ops, R, M, N, cap):
    """gemm_tn_w4_k (csrc/gemm_w4.hip: 384 x 192 tiles, four-slot operand ring, asm transposing reads) at the four weight-gradient
    shapes of the headline DiT and at token counts that leave ragged last token ranges, with and without the workgroup cap;
    leading dimensions wider than the operands (the engine hands it column windows of wider activations)"""
    a = synth.normal(f"tnr.a{R}{M}", (R, M + 64))
    b = synth.normal(f"tnr.b{R}{N}", (R, N + 128))
    init = synth.normal(f"tnr.c{M}{N}", (M, N))
    c = init.to(DEV).clone()
    ops.gemm_tn(dev_bf(a)[:, 64:], dev_bf(b)[:, :N], c, max_wgs=cap)
    ref = init + bf(a)[:, 64:].t() @ bf(b)[:, :N]
    assert rel(c, ref) < 2e-5


@pytest.mark.parametrize("M,N,K,bias,resid", [(8192, 512, 512, True, True), (2048, 1024, 1024, False, False), (4096 + 40, 384, 256, True, False),
                                              (1024, 3072, 1024, True, False), (640, 1000, 448, False, True)])
def test_small_launch_gemm_on_the_four_slot_ring_equals_the_two_slot_form(ops, M, N, K, bias, resid):
    """round 6: launches of at most one workgroup per CU of the 128 x 128 GEMM kernel (the mid-size linears of the low-resolution
    UNet levels and of small-batch DiT steps) run `gemm_nt_k<CONV, 4>` -- a four-slot operand ring with counted waits: three stages in
    flight instead of one, because a workgroup alone on its CU has nobody to cover its memory latency.  Same products in the same
    order: bit-identical to the two-slot form (lab switch dl_lab_set_nt_deep), and against torch."""
    a = dev_bf(bf(synth.normal(f"d4.a{M}", (M, K))))
    w = dev_bf(bf(synth.normal(f"d4.w{N}", (N, K), std=K**-0.5)))
    b_ = synth.normal("d4.b", (N,), std=0.1).to(DEV) if bias else None
    r_ = dev_bf(bf(synth.normal(f"d4.r{M}", (M, N)))) if resid else None
    outs = []
    for mode in (0, 1, 1):
        ops.lib().cdll.dl_lab_set_nt_deep(mode)
        try:
            o = torch.full((M, N), 7.0, device=DEV, dtype=torch.bfloat16)
            ops.gemm_nt(a, w, o, bias=b_, resid=r_)
            torch.cuda.synchronize()
        finally:
            ops.lib().cdll.dl_lab_set_nt_deep(1)
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    ref = a.float() @ w.float().t() + (b_ if bias else 0) + (r_.float() if resid else 0)
    assert rel(outs[1].float(), ref) < 5e-3


@pytest.mark.parametrize("M,N0,N1,K0,K1,bias,resid", [(1024, 1024, 2048, 1024, 1024, True, False), (4096, 512, 1024, 512, 512, True, True),
                                                      (2048, 1024, 1024, 1024, 2048, False, False), (1000, 136, 264, 64, 192, True, True),
                                                      (16384, 256, 512, 256, 256, True, False), (64, 128, 128, 128, 64, False, True)])
def test_two_small_products_in_one_launch_equal_their_separate_launches(ops, M, N0, N1, K0, K1, bias, resid):
    """dl_gemm_nt_pair (round 6): two independent products -- the q and kv projections of a UNet AttentionBlock and their data gradients
    -- as ONE launch of the 128 x 128 kernel when both are small (workgroups from `tiles0` on compute the second); larger pairs fall
    back to the two dl_gemm_nt calls.  Either way bit-identical to those calls; outputs are column slices of wider buffers."""
    def mk(tag, N, K):
        a = dev_bf(bf(synth.normal(f"pr.a{tag}{M}", (M, K))))
        w = dev_bf(bf(synth.normal(f"pr.w{tag}{N}", (N, K), std=K**-0.5)))
        b_ = synth.normal(f"pr.b{tag}", (N,), std=0.1).to(DEV) if bias else None
        r_ = dev_bf(bf(synth.normal(f"pr.r{tag}{M}", (M, N)))) if resid else None
        return a, w, b_, r_
    p0, p1 = mk("0", N0, K0), mk("1", N1, K1)
    sep = []
    for a, w, b_, r_ in (p0, p1):
        o = torch.full((M, w.shape[0] + 8), 7.0, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(a, w, o[:, : w.shape[0]], bias=b_, resid=r_)
        sep.append(o)
    outs = [torch.full((M, N0 + 8), 7.0, device=DEV, dtype=torch.bfloat16), torch.full((M, N1 + 8), 7.0, device=DEV, dtype=torch.bfloat16)]
    ops.gemm_nt_pair([(p0[0], p0[1], outs[0][:, :N0], p0[2], p0[3], None, None, None), (p1[0], p1[1], outs[1][:, :N1], p1[2], p1[3], None, None, None)])
    torch.cuda.synchronize()
    assert torch.equal(outs[0], sep[0]) and torch.equal(outs[1], sep[1])
    ref = p1[0].float() @ p1[1].float().t() + (p1[2] if bias else 0) + (p1[3].float() if resid else 0)
    assert rel(outs[1][:, :N1].float(), ref) < 5e-3


@pytest.mark.parametrize("M,N,R", [(512, 512, 8192), (1024, 1024, 2048), (128, 256, 131072), (1024, 512, 8192), (256, 768, 32768), (2048, 1024, 2048),
                                   (200, 136, 4096 + 64), (28672, 512, 128), (64, 128, 16384)])
def test_atomic_weight_gradient_gemm_split_rules(ops, M, N, R):
    """round 6: dl_gemm_tn picks its split count from a cost model (k-steps per range + the ranges' f32 read-modify-writes at the rate
    measured for them) instead of a workgroup count.  Both rules (lab switch) and a forced count of 3 are the same products in
    another f32 summation order; accumulating on top of a non-zero C."""
    a = dev_bf(bf(synth.normal(f"ts.a{M}", (R, M))))
    b_ = dev_bf(bf(synth.normal(f"ts.b{N}", (R, N))))
    ref = a.float().t() @ b_.float()
    lib = ops.lib().cdll
    outs = []
    try:
        for model, force in ((0, 0), (1, 0), (1, 3)):
            lib.dl_lab_set_tn_split_model(model)
            lib.dl_lab_set_tn_force_splits(force)
            c = torch.full((M, N + 8), 0.5, device=DEV)
            ops.gemm_tn(a, b_, c[:, :N], M=M, N=N)
            assert bool((c[:, N:] == 0.5).all())
            outs.append(c[:, :N] - 0.5)
    finally:
        lib.dl_lab_set_tn_split_model(1)
        lib.dl_lab_set_tn_force_splits(0)
    scale = ref.abs().max()
    for o in outs:
        assert float((o - ref).abs().max() / scale) < 2e-5 * (1 + R / 8192)


@pytest.mark.parametrize("R,cap,ranges,D", [(4096 + 32 * 7, 0, 8, 384), (8192, 128, 8, 384), (2048, 64, 8, 384), (4096, 0, 1, 384),
                                            (16384, 0, 3, 384), (8192, 0, 8, 512), (2048 + 32 * 5, 192, 2, 512), (4096, 0, 8, 768),
                                            (4096, 0, 8, 256),
                                            # round 6, widths that are not whole tiles (the 640-wide joint DDT of ddt_txt.yaml; 320):
                                            # 256 x 256 tiles, the last tile of a row / column shifted back to end at the edge
                                            (4096, 0, 8, 640), (2048 + 32 * 3, 96, 4, 320)])
def test_gemm_tn_group_is_exact_and_bit_reproducible(ops, R, cap, ranges, D):
    """dl_gemm_tn_group (csrc/gemm_w4.hip): the four weight gradients of a DiT block (qkv, proj_out, MLP up / down) in ONE launch --
    32 tiles of 384 x 192 at D = 384 (768: 128 tiles), 64 tiles of 256 x 256 at D = 512 (the CIFAR / SPRINT / DDT width; 256: 16 tiles)
    -- partial tiles per token range in a slab, folded in a fixed order.  += semantics, column windows of wider operands, ragged last
    token range, a slab that only has room for `ranges` partial images, the workgroup cap -- and two runs are bit-identical (the
    one-problem launches meet in f32 atomics and are not)"""
    F = 4 * D
    shapes = [(3 * D, D), (D, D), (2 * F, D), (D, F)]
    probs, refs = [], []
    for i, (Mo, No) in enumerate(shapes):
        a = synth.normal(f"tng.a{i}{R}", (R, Mo + 64))
        b = synth.normal(f"tng.b{i}{R}", (R, No + 128))
        init = synth.normal(f"tng.c{i}", (Mo, No))
        probs.append((dev_bf(a)[:, 64:], dev_bf(b)[:, :No], init))
        refs.append(init + bf(a)[:, 64:].t() @ bf(b)[:, :No])
    total = sum(m * n for m, n in shapes)
    slab = torch.full((ranges * total,), float("nan"), device=DEV)  # contents on entry are irrelevant
    outs = []
    for _ in range(2):
        gs = [init.to(DEV).clone() for _, _, init in probs]
        assert ops.gemm_tn_group([(a, b, g) for (a, b, _), g in zip(probs, gs)], slab, max_wgs=cap)
        outs.append(gs)
    for g0, g1, ref in zip(outs[0], outs[1], refs):
        assert rel(g0, ref) < 2e-5
        assert torch.equal(g0, g1)
    # a shape below one tile: nothing is launched, the caller keeps the per-problem path
    g = torch.zeros(128, D, device=DEV)
    assert not ops.gemm_tn_group([(probs[0][0][:, :128], probs[0][1], g)], slab)
    assert float(g.abs().max()) == 0.0


def test_cast2d_column_window(ops):
    """dl_cast2d_f32_to_bf16 on a column window of wider rows (the data-parallel backward casts one block's slice of the f32
    modulation-gradient accumulator at a time): bit-equal to torch's rounding, nothing written outside the window"""
    src = torch.randn(37, 512, device=DEV)
    dst = torch.full((64, 512), 7.0, device=DEV, dtype=torch.bfloat16)
    ops.cast2d_f32_to_bf16(src[:, 128:320], dst[:37, 128:320])
    want = torch.full((64, 512), 7.0, device=DEV, dtype=torch.bfloat16)
    want[:37, 128:320] = src[:, 128:320].to(torch.bfloat16)
    assert torch.equal(dst, want)


# ------------------------------------------------------------------ adaLN
@pytest.mark.parametrize("D,affine", [(384, True), (768, True), (384, False), (128, True)])
def test_ln_modulate_fwd_bwd(ops, D, affine):
    B, N = 3, 64
    M = B * N
    x = bf(synth.normal("ln.x", (M, D))).requires_grad_(True)
    mod = bf(synth.normal("ln.mod", (B, 3 * D), std=0.3))
    sc, sh = mod[:, :D].clone().requires_grad_(True), mod[:, D : 2 * D].clone().requires_grad_(True)
    w = (1 + synth.normal("ln.w", (D,), std=0.1)).requires_grad_(affine)
    b = synth.normal("ln.b", (D,), std=0.1).requires_grad_(affine)
    eps = 1e-5 if affine else 1e-6
    xb = x.reshape(B, N, D)
    y = odit.layer_norm(xb, w if affine else None, b if affine else None, eps) * (1 + sc[:, None]) + sh[:, None]
    dy = bf(synth.normal("ln.dy", (M, D)))
    dres = bf(synth.normal("ln.dres", (M, D)))
    y.reshape(M, D).backward(dy)

    mod_d = dev_bf(mod)
    out = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    mean = torch.empty(M, device=DEV)
    rstd = torch.empty(M, device=DEV)
    wd, bd = (w.detach().to(DEV), b.detach().to(DEV)) if affine else (None, None)
    ops.ln_modulate_fwd(dev_bf(x.detach()), wd, bd, mod_d[:, :D], mod_d[:, D : 2 * D], N, eps, out, mean, rstd)
    assert rel(out.float(), y.reshape(M, D)) < 4e-3
    dx = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    dmod = torch.zeros(B, 3 * D, device=DEV)  # f32 accumulators (atomics from the workgroups that share a sample)
    dwb = torch.zeros(B, 2, D, device=DEV) if affine else None
    ops.ln_modulate_bwd(dev_bf(dy), dev_bf(x.detach()), wd, bd, mod_d[:, :D], N, mean, rstd, dev_bf(dres), dx,
                        dmod[:, :D], dmod[:, D : 2 * D], dwb)
    assert rel(dx.float(), x.grad + dres) < 5e-3
    assert rel(dmod[:, :D].float(), sc.grad) < 5e-3
    assert rel(dmod[:, D : 2 * D].float(), sh.grad) < 5e-3
    if affine:
        acc = torch.zeros(2 * D, device=DEV)
        ops.reduce_rows_f32(dwb, acc, B, 2 * D, clear=True)
        assert rel(acc[:D], w.grad) < 1e-4 and rel(acc[D:], b.grad) < 1e-4
        assert float(dwb.abs().sum()) == 0.0  # cleared while read
    # no residual input; fused backward of the gated residual that follows (dt = gate * dx, dgate += dx * t)
    tg = bf(synth.normal("ln.tg", (M, D)))
    dt = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    dmod.zero_()
    ops.ln_modulate_bwd(dev_bf(dy), dev_bf(x.detach()), wd, bd, mod_d[:, :D], N, mean, rstd, None, dx, dmod[:, :D],
                        dmod[:, D : 2 * D], dwb, gate_t=dev_bf(tg), gate=mod_d[:, 2 * D :], dt=dt, dgate=dmod[:, 2 * D :])
    assert rel(dx.float(), x.grad) < 5e-3
    dxr = dx.float().cpu()
    assert rel(dt.float(), dxr * mod[:, 2 * D :].repeat_interleave(N, 0)) < 4e-3
    assert rel(dmod[:, 2 * D :], (dxr * tg).reshape(B, N, D).sum(1)) < 1e-4


def test_ln_modulate_fwd_with_fused_gated_residual(ops):
    """x' = x + gate * t (mmdit.py:302,308) applied and stored by the LayerNorm kernel of the next sub-layer"""
    B, N, D = 3, 64, 384
    M = B * N
    x, t = bf(synth.normal("lg.x", (M, D))), bf(synth.normal("lg.t", (M, D)))
    mod = bf(synth.normal("lg.mod", (B, 3 * D), std=0.3))
    w, b = 1 + synth.normal("lg.w", (D,), std=0.1), synth.normal("lg.b", (D,), std=0.1)
    xn = bf(x + mod[:, 2 * D :].repeat_interleave(N, 0) * t)  # what the kernel stores (bf16) and normalises
    y = odit.layer_norm(xn.reshape(B, N, D), w, b, 1e-5) * (1 + mod[:, None, :D]) + mod[:, None, D : 2 * D]
    mod_d = dev_bf(mod)
    out, x_out = (torch.empty(M, D, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    ops.ln_modulate_fwd(dev_bf(x), w.to(DEV), b.to(DEV), mod_d[:, :D], mod_d[:, D : 2 * D], N, 1e-5, out, mean, rstd,
                        t=dev_bf(t), gate=mod_d[:, 2 * D :], x_out=x_out)
    assert rel(x_out.float(), xn) < 1e-3  # f32 fma + one bf16 rounding (vs mul, add, rounding on the CPU: <= 1 bf16 ulp)
    assert rel(out.float(), y.reshape(M, D)) < 4e-3
    assert rel(mean, xn.mean(-1)) < 1e-4


def test_gate_bwd(ops):
    B, N, D = 3, 64, 384
    M = B * N
    dout, t = bf(synth.normal("g.do", (M, D))), bf(synth.normal("g.t", (M, D)))
    gate = bf(synth.normal("g.g", (B, 2 * D)))
    dt = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    dg = torch.zeros(B, 2 * D, device=DEV)  # f32
    gd = dev_bf(gate)
    ops.gate_bwd(dev_bf(dout), dev_bf(t), gd[:, D:], N, dt, dg[:, D:])
    assert rel(dt.float(), dout * gate[:, D:].repeat_interleave(N, 0)) < 4e-3
    assert rel(dg[:, D:].float(), (dout * t).reshape(B, N, D).sum(1)) < 4e-3


# ------------------------------------------------------------------ QK norm + rope
@pytest.mark.parametrize("H,gh,gw", [(6, 8, 8), (2, 4, 16), (12, 8, 8)])
def test_qk_norm_rope_fwd_bwd(ops, H, gh, gw):
    B, dh = 2, 64
    N, D = gh * gw, H * dh
    M = B * N
    qkv = bf(synth.normal("qk.qkv", (M, 3 * D))).requires_grad_(True)
    sq = (1 + synth.normal("qk.sq", (D,), std=0.1)).requires_grad_(True)
    sk = (1 + synth.normal("qk.sk", (D,), std=0.1)).requires_grad_(True)
    cos, sin = odit.rope_tables(gh, gw, [32, 32], 10_000.0)
    q, k, v = qkv.reshape(B, N, 3 * D).split(D, dim=-1)
    qr = odit.apply_rope(odit.rms_norm(q, sq).reshape(B, N, H, dh), cos, sin).transpose(1, 2)
    kr = odit.apply_rope(odit.rms_norm(k, sk).reshape(B, N, H, dh), cos, sin).transpose(1, 2)
    vr = v.reshape(B, N, H, dh).transpose(1, 2)
    dq, dk, dv = (bf(synth.normal(f"qk.d{n}", (B, H, N, dh))) for n in "qkv")
    (qr * dq).sum().add((kr * dk).sum()).add((vr * dv).sum()).backward()

    qo, ko, vo = (torch.empty(B, H, N, dh, device=DEV, dtype=torch.bfloat16) for _ in range(3))
    rrms = torch.empty(M, 2, device=DEV)
    ops.qk_norm_rope_fwd(dev_bf(qkv.detach()), sq.detach().to(DEV), sk.detach().to(DEV), cos.to(DEV), sin.to(DEV), qo,
                         ko, vo, rrms, B, N, H, dh, 64)
    assert rel(qo.float(), qr) < 4e-3 and rel(ko.float(), kr) < 4e-3
    assert torch.equal(vo.float().cpu(), vr.detach())
    dqkv = torch.empty(M, 3 * D, device=DEV, dtype=torch.bfloat16)
    dscale = torch.zeros(2, D, device=DEV)
    ops.qk_norm_rope_bwd(dev_bf(dq), dev_bf(dk), dev_bf(dv), dev_bf(qkv.detach()), sq.detach().to(DEV),
                         sk.detach().to(DEV), cos.to(DEV), sin.to(DEV), rrms, dqkv, dscale, B, N, H, dh, 64)
    assert rel(dqkv.float(), qkv.grad) < 5e-3
    assert rel(dscale[0], sq.grad) < 1e-4 and rel(dscale[1], sk.grad) < 1e-4


# ------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,H,N", [(2, 6, 256), (1, 2, 64), (3, 3, 128), (1, 2, 512), (2, 1, 1024), (1, 1, 2048)])  # > 256 tokens: chunked (tiled) kernels
def test_attention_fwd_bwd(ops, B, H, N):
    dh = 64
    q, k, v = (bf(synth.normal(f"at.{n}{N}", (B, H, N, dh))).requires_grad_(True) for n in "qkv")
    scale = dh**-0.5
    o = odit.attention(q, k, v, scale)  # [B,H,N,dh]
    o_tok = o.transpose(1, 2).reshape(B, N, H * dh)
    do = bf(synth.normal(f"at.do{N}", (B, N, H * dh)))
    o_tok.backward(do)
    lse_ref = torch.logsumexp(q.detach() @ k.detach().transpose(-1, -2) * scale, dim=-1)

    out = torch.empty(B, N, H * dh, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(B, H, N, device=DEV)
    qd, kd, vd = dev_bf(q.detach()), dev_bf(k.detach()), dev_bf(v.detach())
    ops.attn_fwd(qd, kd, vd, out, lse, B, H, N, dh, scale)
    assert rel(out.float(), o_tok) < 5e-3
    assert rel(lse, lse_ref) < 1e-4
    dq, dk, dv = (torch.empty(B, H, N, dh, device=DEV, dtype=torch.bfloat16) for _ in range(3))
    ops.attn_bwd(qd, kd, vd, out, dev_bf(do), lse, dq, dk, dv, B, H, N, dh, scale)
    assert rel(dv.float(), v.grad) < 8e-3
    assert rel(dq.float(), q.grad) < 8e-3
    assert rel(dk.float(), k.grad) < 8e-3


@pytest.mark.parametrize("B,H,N", [(2, 6, 256), (3, 2, 64), (2, 3, 192)])
def test_attention_reads_v_and_writes_dv_inside_the_qkv_rows(ops, B, H, N):
    """dl_attn_{fwd,bwd}_sv: V addressed in place in the v third of token-major qkv rows [B*N, 3D], dV written into the v third of
    dqkv (whose q / k thirds must stay untouched) -- bit-identical to the head-major call on the same values."""
    dh, D = 64, H * 64
    scale = dh**-0.5
    q, k = (dev_bf(bf(synth.normal(f"sv.{n}{N}", (B, H, N, dh)))) for n in "qk")
    qkv = dev_bf(bf(synth.normal(f"sv.qkv{N}", (B * N, 3 * D))))
    v = qkv.view(B, N, 3, H, dh)[:, :, 2].permute(0, 2, 1, 3).contiguous()  # 'b n (h d) -> b h n d' of the v third
    do = dev_bf(bf(synth.normal(f"sv.do{N}", (B, N, D))))
    out0, out1 = (torch.empty(B, N, D, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    lse0, lse1 = (torch.empty(B, H, N, device=DEV) for _ in range(2))
    ops.attn_fwd(q, k, v, out0, lse0, B, H, N, dh, scale)
    ops.attn_fwd_qkv(q, k, qkv, out1, lse1, B, H, N, dh, scale)
    assert torch.equal(out0, out1) and torch.equal(lse0, lse1)
    dq0, dk0, dv0, dq1, dk1 = (torch.empty(B, H, N, dh, device=DEV, dtype=torch.bfloat16) for _ in range(5))
    dqkv = torch.full((B * N, 3 * D), 7.0, device=DEV, dtype=torch.bfloat16)
    ops.attn_bwd(q, k, v, out0, do, lse0, dq0, dk0, dv0, B, H, N, dh, scale)
    ops.attn_bwd_qkv(q, k, qkv, out1, do, lse1, dq1, dk1, dqkv, B, H, N, dh, scale)
    assert torch.equal(dq0, dq1) and torch.equal(dk0, dk1)
    got = dqkv.view(B, N, 3, H, dh)
    assert torch.equal(got[:, :, 2].permute(0, 2, 1, 3), dv0)
    assert bool((got[:, :, :2] == 7.0).all())


@pytest.mark.parametrize("B,H,N,gh,gw", [(2, 6, 256, 16, 16), (3, 2, 64, 8, 8), (2, 8, 128, 8, 16)])
def test_attention_backward_token_major_and_qk_norm_backward_in_place(ops, B, H, N, gh, gw):
    """dl_attn_bwd_tok + dl_qk_norm_rope_bwd_inplace (round 3): dQ / dK / dV leave the attention backward token-major inside the dqkv
    rows and the QK-norm + RoPE backward turns the q / k thirds into the pre-norm gradient in place.  The attention part is
    BIT-IDENTICAL to dl_attn_bwd_sv (only the addressing differs), the norm part equals dl_qk_norm_rope_bwd up to a bf16 ulp; the
    scale gradient is summed in a fixed order without atomics: equal to the atomic one to f32 rounding, += semantics, and both
    outputs are identical from run to run."""
    dh, D, M = 64, H * 64, B * N
    scale = dh**-0.5
    qkv = dev_bf(bf(synth.normal(f"ip.qkv{N}", (M, 3 * D))))
    sq, sk = (1 + synth.normal("ip.sq", (D,), std=0.1)).to(DEV), (1 + synth.normal("ip.sk", (D,), std=0.1)).to(DEV)
    cos, sin = (t.to(DEV) for t in odit.rope_tables(gh, gw, [32, 32], 10_000.0))
    q, k = (torch.empty(B, H, N, dh, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    rrms = torch.empty(M, 2, device=DEV)
    ops.qk_norm_rope_fwd(qkv, sq, sk, cos, sin, q, k, None, rrms, B, N, H, dh, 64)
    out, lse = torch.empty(B, N, D, device=DEV, dtype=torch.bfloat16), torch.empty(B, H, N, device=DEV)
    ops.attn_fwd_qkv(q, k, qkv, out, lse, B, H, N, dh, scale)
    do = dev_bf(bf(synth.normal(f"ip.do{N}", (B, N, D))))
    # head-major pair
    dq, dk = (torch.empty(B, H, N, dh, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    dqkv0 = torch.empty(M, 3 * D, device=DEV, dtype=torch.bfloat16)
    init = synth.normal("ip.ds0", (2, D)).to(DEV)
    ds0 = init.clone()
    ops.attn_bwd_qkv(q, k, qkv, out, do, lse, dq, dk, dqkv0, B, H, N, dh, scale)
    ops.qk_norm_rope_bwd(dq, dk, None, qkv, sq, sk, cos, sin, rrms, dqkv0, ds0, B, N, H, dh, 64)
    # token-major, in place (twice: bit-reproducible)
    part = torch.full((1024 * 2 * D,), float("nan"), device=DEV)
    res = []
    for _ in range(2):
        dqkv1 = torch.full((M, 3 * D), 7.0, device=DEV, dtype=torch.bfloat16)
        ds1 = init.clone()
        ops.attn_bwd_tok(q, k, qkv, out, do, lse, dqkv1, B, H, N, dh, scale)
        got = dqkv1.view(B, N, 3, H, dh)
        assert torch.equal(got[:, :, 0].permute(0, 2, 1, 3), dq) and torch.equal(got[:, :, 1].permute(0, 2, 1, 3), dk)
        assert ops.qk_norm_rope_bwd_inplace(qkv, sq, sk, cos, sin, rrms, dqkv1, ds1, part, B, N, H, dh, 64)
        assert torch.equal(dqkv1[:, 2 * D :], dqkv0[:, 2 * D :])  # dV: the same stores of the same kernel
        # q / k thirds: the same formulas in two separately compiled kernels (fma contraction differs): a bf16 ulp here and there
        assert rel(dqkv1[:, : 2 * D].float(), dqkv0[:, : 2 * D].float()) < 1e-3
        assert float((dqkv1[:, : 2 * D].float() - dqkv0[:, : 2 * D].float()).abs().max()) <= 2.0**-7 * float(dqkv0.float().abs().max())
        assert rel(ds1 - init, ds0 - init) < 1e-4
        res.append((dqkv1, ds1))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert not ops.qk_norm_rope_bwd_inplace(qkv, sq, sk, cos, sin, rrms, dqkv1, ds1, torch.empty(1024 * 2 * 1024, device=DEV), B, N, 16, 64, 64)


@pytest.mark.parametrize("B,H,Nq,Nk,valid", [(2, 2, 256, 512, (320, 512)), (2, 3, 512, 512, (300, 77)), (1, 1, 256, 256, (200,))])
def test_attention_cross_lengths_and_key_mask(ops, B, H, Nq, Nk, valid):
    """general form: Nq queries against Nk keys with a key-padding mask given as an additive bias (0 / -inf): cross-attention
    of a resampler (320 valid keys padded to 512) and joint text-image attention with padded text tokens"""
    dh, scale = 64, 64**-0.5
    q = bf(synth.normal("cx.q", (B, H, Nq, dh))).requires_grad_(True)
    k = bf(synth.normal("cx.k", (B, H, Nk, dh))).requires_grad_(True)
    v = bf(synth.normal("cx.v", (B, H, Nk, dh))).requires_grad_(True)
    keep = torch.zeros(B, Nk, dtype=torch.bool)
    for i in range(B):
        keep[i, : valid[i % len(valid)]] = True
    bias = torch.zeros(B, Nk).masked_fill(~keep, float("-inf"))
    att = torch.softmax(q @ k.transpose(-1, -2) * scale + bias[:, None, None, :], dim=-1) @ v
    o_tok = att.transpose(1, 2).reshape(B, Nq, H * dh)
    do = bf(synth.normal("cx.do", (B, Nq, H * dh)))
    o_tok.backward(do)
    out = torch.empty(B, Nq, H * dh, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(B, H, Nq, device=DEV)
    qd, kd, vd, bd = dev_bf(q.detach()), dev_bf(k.detach()), dev_bf(v.detach()), bias.to(DEV)
    ops.attn_fwd_ex(qd, kd, vd, out, lse, B, H, Nq, Nk, dh, scale, bd)
    assert rel(out.float(), o_tok) < 5e-3
    dq = torch.empty(B, H, Nq, dh, device=DEV, dtype=torch.bfloat16)
    dk, dv = (torch.empty(B, H, Nk, dh, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    ops.attn_bwd_ex(qd, kd, vd, out, dev_bf(do), lse, dq, dk, dv, B, H, Nq, Nk, dh, scale, bd)
    assert rel(dq.float(), q.grad) < 8e-3 and rel(dk.float(), k.grad) < 8e-3 and rel(dv.float(), v.grad) < 8e-3
    masked = (~keep)[:, None, :, None].expand(B, H, Nk, dh)
    assert float(dk.float().cpu()[masked].abs().max()) == 0.0 and float(dv.float().cpu()[masked].abs().max()) == 0.0


def test_attention_peaked_rows(ops):
    """online-softmax rescale branch: one key dominates per query at a late key tile (guide rule 26)."""
    B, H, N, dh = 1, 1, 256, 64
    q, k, v = (bf(synth.normal(f"pk.{n}", (B, H, N, dh))) for n in "qkv")
    k[0, 0, 200] = q[0, 0, 7] * 4.0  # score spike for query 7 at key 200 (4th key tile)
    k = bf(k)
    scale = dh**-0.5
    ref = odit.attention(q, k, v, scale).transpose(1, 2).reshape(B, N, dh)
    out = torch.empty(B, N, dh, device=DEV, dtype=torch.bfloat16)
    lse = torch.empty(B, H, N, device=DEV)
    ops.attn_fwd(dev_bf(q), dev_bf(k), dev_bf(v), out, lse, B, H, N, dh, scale)
    assert (out.float().cpu() - ref).abs().max() < 3e-2
    assert rel(out.float(), ref) < 5e-3


# ------------------------------------------------------------------ swiglu / small ops
def test_swiglu(ops):
    M, F = 192, 1536
    u = bf(synth.normal("sw.u", (M, 2 * F))).requires_grad_(True)
    h = odit.silu(u[:, :F]) * u[:, F:]
    dh = bf(synth.normal("sw.dh", (M, F)))
    h.backward(dh)
    ho = torch.empty(M, F, device=DEV, dtype=torch.bfloat16)
    ops.swiglu_fwd(dev_bf(u.detach()), ho)
    assert rel(ho.float(), h) < 4e-3
    du = torch.empty(M, 2 * F, device=DEV, dtype=torch.bfloat16)
    ops.swiglu_bwd(dev_bf(dh), dev_bf(u.detach()), du)
    assert rel(du.float(), u.grad) < 4e-3


def test_patchify_unpatchify(ops):
    cfg = odit.DiTConfig(input_channels=3, output_channels=3, patch_size=2)
    B, C, H, W, p = 2, 3, 8, 12, 2
    x = synth.normal("pt.x", (B, C, H, W))
    gh, gw = H // p, W // p
    tok = torch.full((B * gh * gw, 64), 7.0, device=DEV, dtype=torch.bfloat16)
    ops.patchify(x.to(DEV), tok, p, ops.PATCH_CPP)
    ref = x.reshape(B, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(B * gh * gw, C * p * p)
    assert torch.equal(tok[:, : C * p * p].float().cpu(), bf(ref))
    assert (tok[:, C * p * p :] == 0).all()
    t32 = synth.normal("pt.tok", (B * gh * gw, 16))
    img = torch.empty(B, C, H, W, device=DEV)
    ops.unpatchify(t32.to(DEV), img, p)
    assert torch.equal(img.cpu(), odit.unpatchify(t32[:, :12].reshape(B, gh * gw, 12), gh, gw, cfg))
    # PPC order == transpose of unpatchify
    ops.patchify(img, tok, p, ops.PATCH_PPC)
    assert torch.equal(tok[:, :12].float().cpu(), bf(t32[:, :12]))


def test_timestep_embedding_and_cond(ops):
    B, E = 5, 384
    t = synth.uniform("te.t", (B,))
    out = torch.empty(B, 256, device=DEV, dtype=torch.bfloat16)
    ops.timestep_embedding(t.to(DEV), out)
    assert (out.float().cpu() - odit.timestep_embedding(t, 256)).abs().max() < 4e-3  # one bf16 rounding of [-1,1]
    ti = torch.tensor([0.0, 1.0, 17.0, 500.0, 999.0])
    ops.timestep_embedding(ti.to(DEV), out)
    assert (out.float().cpu() - odit.timestep_embedding(ti, 256)).abs().max() < 4.5e-3  # + fp32 sincos argument error
    e = synth.normal("te.e", (B, E)).requires_grad_(True)
    table = synth.normal("te.tab", (11, E)).requires_grad_(True)
    idx = torch.tensor([3, 10, 3, 0, 7])
    emb_ref = e + table[idx]
    act_ref = odit.silu(emb_ref)
    dact = synth.normal("te.dact", (B, E))
    act_ref.backward(dact)
    emb = torch.empty(B, E, device=DEV)
    act = torch.empty(B, E, device=DEV, dtype=torch.bfloat16)
    ops.cond_combine_fwd(e.detach().to(DEV), table.detach().to(DEV), idx.to(DEV), emb, act)
    assert rel(emb, emb_ref) < 1e-6 and rel(act.float(), act_ref) < 4e-3
    demb = torch.empty(B, E, device=DEV)
    demb16 = torch.empty(B, E, device=DEV, dtype=torch.bfloat16)
    dtab = torch.zeros(11, E, device=DEV)
    ops.cond_combine_bwd(dact.to(DEV), emb, idx.to(DEV), demb, demb16, dtab)
    assert rel(demb, e.grad) < 1e-5 and rel(dtab, table.grad) < 1e-5
    # colsum (+= semantics) on both dtypes, strided
    xs = synth.normal("cs.x", (300, 2 * E))
    acc = torch.ones(E, device=DEV)
    ops.colsum(xs.to(DEV)[:, E:], acc, 300, E)
    assert rel(acc, 1 + xs[:, E:].sum(0)) < 1e-5
    acc.zero_()
    ops.colsum(dev_bf(xs)[:, :E], acc, 300, E)
    assert rel(acc, bf(xs)[:, :E].sum(0)) < 1e-5
    pre = bf(synth.normal("sb.pre", (B, E))).requires_grad_(True)
    odit.silu(pre).backward(dact)
    dx = torch.empty(B, E, device=DEV, dtype=torch.bfloat16)
    ops.silu_bwd(dact.to(DEV), dev_bf(pre.detach()), dx)
    assert rel(dx.float(), pre.grad) < 4e-3


# ------------------------------------------------------------------ diffusion heads & sampler steps
def test_noising_and_losses(ops):
    B, shp = 5, (5, 4, 32, 32)
    x, nz = synth.normal("h.x", shp), synth.normal("h.n", shp)
    t = synth.uniform("h.t", (B,))
    z = ops.flow_add_noise(x.to(DEV), nz.to(DEV), t.to(DEV))
    assert rel(z, od.flow_add_noise(x, t, nz)) < 1e-6
    T = od.GaussianTables(1000)
    ti = torch.tensor([0, 3, 500, 998, 999], dtype=torch.int32)
    xt = ops.ddpm_add_noise(x.to(DEV), nz.to(DEV), ti.to(DEV), T.sqrt_alphas_bar.float().to(DEV), T.alphas_bar.float().to(DEV))
    assert rel(xt, od.ddpm_add_noise(T, x, ti, nz)) < 1e-6
    pred = synth.normal("h.p", shp).requires_grad_(True)
    l_ref = od.flow_loss(pred, x, nz)
    l_ref.backward()
    loss = ops.mse_loss_fwd(pred.detach().to(DEV), nz.to(DEV), x.to(DEV), ops.LOSS_FLOW)
    assert abs(loss.item() - l_ref.item()) < 1e-6 * abs(l_ref.item())
    dp = ops.mse_loss_bwd(pred.detach().to(DEV), nz.to(DEV), x.to(DEV), 1.0, ops.LOSS_FLOW)
    assert rel(dp, pred.grad) < 1e-6
    pred.grad = None
    l2 = od.mse_loss(pred, nz)
    l2.backward()
    loss = ops.mse_loss_fwd(pred.detach().to(DEV), nz.to(DEV), None, ops.LOSS_EPS)
    assert abs(loss.item() - l2.item()) < 1e-6 * abs(l2.item())
    dp = ops.mse_loss_bwd(pred.detach().to(DEV), nz.to(DEV), None, 0.5, ops.LOSS_EPS)
    assert rel(dp, 0.5 * pred.grad) < 1e-6
    # odd sizes (not a multiple of 4) take the scalar path
    xo, no_, to_ = synth.normal("h.xo", (3, 1, 5, 7)), synth.normal("h.no", (3, 1, 5, 7)), synth.uniform("h.to", (3,))
    assert rel(ops.flow_add_noise(xo.to(DEV), no_.to(DEV), to_.to(DEV)), od.flow_add_noise(xo, to_, no_)) < 1e-6
    v = ops.flow_x_to_v(z, pred.detach().to(DEV), t.clamp(min=0.05).to(DEV))
    assert rel(v, od.flow_x_to_v(z.cpu(), pred.detach(), t.clamp(min=0.05))) < 1e-6


def test_sampler_steps_against_reference_fixtures(ops, golden):
    """the committed reference outputs (tests/golden/samplers.npz) are the expected values here."""
    g = golden("samplers")
    shp = (3, 4, 8, 8)
    xt, v, nz = (synth.normal(k, shp).to(DEV) for k in ("smp.xt", "smp.v", "smp.noise"))
    xp, x0 = ops.euler_step(xt, v, None, 0.0, 0.75, 0.5)
    assert rel(xp, torch.from_numpy(g["euler_x_prev"])) < 1e-6 and rel(x0, torch.from_numpy(g["euler_x0"])) < 1e-6
    # CFG fused into the step == combine then step
    vu = synth.normal("smp.vu", shp)
    xp, _ = ops.euler_step(xt, v, vu.to(DEV), 2.0, 0.75, 0.5)
    assert rel(xp, od.euler_step(xt.cpu(), od.cfg_combine(v.cpu(), vu, 2.0), 0.75, 0.5)["x_prev"]) < 1e-6
    tmax = od.flow_timesteps(10)[1]
    sigma = ((0.6 / (1 - min(0.6, tmax))) ** 0.5) * 0.7
    xp, mean, x0, lp, std = ops.euler_maruyama_step(xt, v, None, 0.0, torch.from_numpy(g["em_noise"]).to(DEV), None, 0.6,
                                                    0.5, sigma)
    for got, key in ((xp, "x_prev"), (mean, "x_prev_mean"), (x0, "estimated_x0"), (lp, "logprob")):
        assert rel(got, torch.from_numpy(g["em_" + key])) < 2e-6, key
    assert abs(std - float(g["em_x_prev_std"])) < 1e-7
    sigma2 = ((1.0 / (1 - min(1.0, tmax))) ** 0.5) * 0.7
    _, mean, _, lp, _ = ops.euler_maruyama_step(xt, v, None, 0.0, None, nz, 1.0, 0.9, sigma2)
    assert rel(lp, torch.from_numpy(g["em2_logprob"])) < 2e-6 and rel(mean, torch.from_numpy(g["em2_mean"])) < 2e-6
    T = od.GaussianTables(1000)
    tt = torch.tensor([0, 7, 999], dtype=torch.int32).to(DEV)
    dn = torch.from_numpy(g["ddpm_noise"]).to(DEV)
    for vt in ("fixed_small", "fixed_large"):
        if vt == "fixed_small":
            var, lv = T.posterior_variance, T.posterior_log_variance_clipped
        else:
            var = torch.cat([T.posterior_variance[1:2], T.betas[1:]])
            lv = torch.log(var)
        tab = torch.stack([T.sqrt_alphas_bar, T.alphas_bar, T.posterior_mean_coef1, T.posterior_mean_coef2, var, lv]).float().to(DEV)
        for mt in ("epsilon", "xstart", "xprev"):
            for clamp in (False, True):
                outs = ops.ddpm_step(v, None, 0.0, xt, dn, tt, tab, ops.MEAN_TYPES[mt], clamp)
                tag = f"ddpm_{mt}_{vt}_{int(clamp)}_"
                for got, key in zip(outs, ("x_prev", "estimated_x0", "x_prev_mean", "x_prev_std", "logprob")):
                    assert rel(got, torch.from_numpy(g[tag + key])) < 3e-6, tag + key
    tab3 = torch.stack([T.sqrt_alphas_bar, T.alphas_bar, T.alphas_bar_prev]).float().to(DEV)
    for eta in (0.0, 0.5):
        xp, x0, mean, std, lp = ops.ddim_step(v, None, 0.0, xt, torch.from_numpy(g["ddim_noise"]).to(DEV), tt, tab3, None,
                                              ops.MEAN_TYPES["epsilon"], False, eta)
        assert rel(xp, torch.from_numpy(g[f"ddim_eta{eta}_x_prev"])) < 3e-6
        assert rel(x0, torch.from_numpy(g[f"ddim_eta{eta}_estimated_x0"])) < 3e-6
        assert rel(mean, torch.from_numpy(g[f"ddim_eta{eta}_x_prev_mean"])) < 3e-6
        if eta > 0:
            assert rel(std, torch.from_numpy(g[f"ddim_eta{eta}_x_prev_std"])) < 3e-6
            ref_lp = np.nan_to_num(g[f"ddim_eta{eta}_logprob"], nan=0, posinf=0, neginf=0)
            assert rel(torch.nan_to_num(lp, 0, 0, 0), torch.from_numpy(ref_lp)) < 3e-6


# ------------------------------------------------------------------ optimizer side
def test_adamw_and_casts(ops):
    n = 10007
    p0, g = synth.normal("ad.p", (n,)), synth.normal("ad.g", (n,))
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-3, weight_decay=0.01, betas=(0.9, 0.999), eps=1e-8)
    pad = 4 - n % 4
    p = torch.cat([p0, torch.zeros(pad)]).to(DEV)[:n]
    m, v = torch.zeros(n + pad, device=DEV)[:n], torch.zeros(n + pad, device=DEV)[:n]
    for step in range(1, 4):
        pr.grad = g * step
        opt.step()
        ops.adamw_step(p, (g * step).to(DEV), m, v, 1e-3, 0.9, 0.999, 1e-8, 0.01, step)
        assert rel(p, pr.detach()) < 1e-6, step
    w = synth.normal("cw.w", (100, 72))
    d = torch.full((100, 128), 5.0, device=DEV, dtype=torch.bfloat16)
    dT = torch.full((72, 128), 5.0, device=DEV, dtype=torch.bfloat16)
    ops.cast_weight(w.to(DEV), d, dT)
    assert torch.equal(d[:, :72].float().cpu(), bf(w)) and (d[:, 72:] == 0).all()
    assert torch.equal(dT[:, :100].float().cpu(), bf(w).t()) and (dT[:, 100:] == 0).all()
    e = torch.zeros(n, device=DEV)
    ops.ema_update(e, p, 0.9)
    assert rel(e, 0.1 * p) < 1e-6


def test_fused_swiglu_gemms(ops):
    """MLP-up GEMM + SwiGLU forward fused into the GEMM epilogue, and the MLP backward that recomputes the pre-activations, == the
    unfused pairs."""
    M, D, F = 16384, 384, 1536
    x = synth.normal("fs.x", (M, D))
    w1 = synth.normal("fs.w1", (2 * F, D), std=D**-0.5)
    w2 = synth.normal("fs.w2", (D, F), std=F**-0.5)
    dt = synth.normal("fs.dt", (M, D))
    w1p = torch.empty(2 * F, D, device=DEV, dtype=torch.bfloat16)
    ops.cast_weight_swiglu(w1.to(DEV), w1p)
    u = torch.empty(M, 2 * F, device=DEV, dtype=torch.bfloat16)
    h = torch.empty(M, F, device=DEV, dtype=torch.bfloat16)
    assert ops.gemm_nt_swiglu(dev_bf(x), w1p, u, h)
    u_ref = torch.empty_like(u)
    h_ref = torch.empty_like(h)
    ops.gemm_nt(dev_bf(x), dev_bf(w1), u_ref)
    ops.swiglu_fwd(u_ref, h_ref)
    assert torch.equal(u, u_ref)  # same accumulation order, same rounding
    assert rel(h.float(), h_ref.float()) < 6e-3  # h is computed from the f32 accumulators here, from bf16 u there
    cpu_u = bf(x) @ bf(w1).t()
    assert rel(h.float(), odit.silu(cpu_u[:, :F]) * cpu_u[:, F:]) < 4e-3
    # backward: the dgrad GEMM + elementwise pair on the saved u ...
    w2t = dev_bf(w2.t().contiguous())  # [F, D]
    dh = torch.empty(M, F, device=DEV, dtype=torch.bfloat16)
    du_ref = torch.empty(M, 2 * F, device=DEV, dtype=torch.bfloat16)
    ops.gemm_nt(dev_bf(dt), w2t, dh)
    ops.swiglu_bwd(dh, u, du_ref)
    # ... and without any saved u: the backward recomputes the u tile it needs (bit-identical to the stored one: same accumulation
    # order, same rounding) next to the dh tile (kept in f32 here, rounded to bf16 there)
    du_rc = torch.full_like(du_ref, float("nan"))
    assert ops.mlp_dswiglu_recompute(dev_bf(x), w1p, dev_bf(dt), w2t, du_rc)
    assert rel(du_rc.float(), du_ref.float()) < 6e-3
    xw = torch.zeros(M, D + 64, device=DEV, dtype=torch.bfloat16)  # operands that are column windows of wider rows
    xw[:, 64:] = dev_bf(x)
    duw = torch.zeros(M, 2 * F + 128, device=DEV, dtype=torch.bfloat16)
    assert ops.mlp_dswiglu_recompute(xw[:, 64:], w1p, dev_bf(dt), w2t, duw[:, : 2 * F])
    assert torch.equal(duw[:, : 2 * F], du_rc) and float(duw[:, 2 * F :].abs().sum()) == 0.0
    assert not ops.mlp_dswiglu_recompute(dev_bf(x)[:512], w1p, dev_bf(dt)[:512], w2t, du_rc[:512])  # too few tiles
    # 2F = 4096 (the 512-wide configurations: 256- / 128-wide tiles) through the same three entry points
    D2, F2, M2 = 512, 2048, 8192
    x2, dt2 = synth.normal("fs.x2", (M2, D2)), synth.normal("fs.dt2", (M2, D2))
    w12 = synth.normal("fs.w12", (2 * F2, D2), std=D2**-0.5)
    w22t = dev_bf(synth.normal("fs.w22", (F2, D2), std=F2**-0.5))
    w12p = torch.empty(2 * F2, D2, device=DEV, dtype=torch.bfloat16)
    ops.cast_weight_swiglu(w12.to(DEV), w12p)
    u2, h2 = torch.empty(M2, 2 * F2, device=DEV, dtype=torch.bfloat16), torch.empty(M2, F2, device=DEV, dtype=torch.bfloat16)
    assert ops.gemm_nt_swiglu(dev_bf(x2), w12p, u2, h2)
    u2_ref = torch.empty_like(u2)
    ops.gemm_nt(dev_bf(x2), dev_bf(w12), u2_ref)
    assert torch.equal(u2, u2_ref)
    dh2, du2_ref, du2 = torch.empty(M2, F2, device=DEV, dtype=torch.bfloat16), torch.empty_like(u2), torch.empty_like(u2)
    ops.gemm_nt(dev_bf(dt2), w22t, dh2)
    ops.swiglu_bwd(dh2, u2, du2_ref)
    assert ops.mlp_dswiglu_recompute(dev_bf(x2), w12p, dev_bf(dt2), w22t, du2)
    assert rel(du2.float(), du2_ref.float()) < 6e-3
    # small shapes have no fused kernel: the wrappers say so instead of computing something else
    xs = torch.zeros(256, D, device=DEV, dtype=torch.bfloat16)
    assert not ops.gemm_nt_swiglu(xs, w1p, u[:256], h[:256])


def test_small_gemms_and_column_sums_have_bit_reproducible_forms(ops):
    """dl_gemm_tn_det / dl_gemm_nt_f32_det / dl_colsum_det (round 3): the skinny problems of the head and the conditioning path
    split their contraction over workgroups; with a scratch every split keeps its partial image and a fold adds them in a fixed
    order -- same values as the atomic forms (f32 rounding), += / = semantics kept, and two runs are bit-identical."""
    scr = torch.full((1 << 22,), float("nan"), device=DEV)
    for R, M, N in [(65536, 16, 384), (65536, 384, 16), (256, 384, 384), (8192, 1152, 384), (4096, 136, 264)]:  # incl. the group path
        a, b = synth.normal(f"det.a{R}{M}", (R, M)), synth.normal(f"det.b{R}{N}", (R, N))
        init = synth.normal(f"det.c{M}{N}", (M, N))
        ref = init + bf(a).t() @ bf(b)
        outs = []
        for _ in range(2):
            c = init.to(DEV).clone()
            ops.gemm_tn(dev_bf(a), dev_bf(b), c, scratch=scr)
            outs.append(c)
        assert rel(outs[0], ref) < 2e-5 and torch.equal(outs[0], outs[1]), (R, M, N)
    # a scratch with room for two partial images only: fewer splits, same result
    a, b = synth.normal("det.a6553616", (65536, 16)), synth.normal("det.b65536384", (65536, 384))
    c = torch.zeros(16, 384, device=DEV)
    ops.gemm_tn(dev_bf(a), dev_bf(b), c, scratch=scr[: 2 * 16 * 384])
    assert rel(c, bf(a).t() @ bf(b)) < 2e-5
    for M, N, K in [(256, 384, 28416), (64, 136, 4096), (256, 384, 384)]:  # split-K, split-K ragged, no split
        a, b = synth.normal(f"dnt.a{M}{K}", (M, K)), synth.normal(f"dnt.b{N}{K}", (N, K), std=K**-0.5)
        outs = []
        for _ in range(2):
            c = torch.full((M, N), float("nan"), device=DEV)  # '=' semantics: the previous content is irrelevant
            ops.gemm_nt(dev_bf(a), dev_bf(b), c, scratch=scr)
            outs.append(c)
        assert rel(outs[0], bf(a) @ bf(b).t()) < 2e-5 and torch.equal(outs[0], outs[1]), (M, N, K)
    for R, C, dt in [(65536, 16, torch.bfloat16), (256, 28416, torch.bfloat16), (300, 384, torch.float32)]:
        x = synth.normal(f"dcs.{R}{C}", (R, C)).to(dt)
        init = synth.normal(f"dcs.o{C}", (C,))
        outs = []
        for _ in range(2):
            o = init.to(DEV).clone()
            ops.colsum(x.to(DEV), o, scratch=scr)
            outs.append(o)
        assert rel(outs[0], init + x.float().sum(0)) < 1e-5 and torch.equal(outs[0], outs[1]), (R, C)


@pytest.mark.parametrize("R,C,ld", [(4096 + 37, 13312, 13312), (8192, 512, 1536), (65536, 16, 16), (1000, 24, 32), (300, 8, 8),
                                    (2048, 520, 528), (5000, 12, 16)])  # the last: C % 8 != 0 -> the one-element-per-lane kernel
def test_colsum_bf16_vector_path(ops, R, C, ld):
    """dl_colsum / dl_colsum_det on bf16 rows: 16 bytes per lane, VL lanes across the columns and 256 / VL thread rows down the slab
    (colsum_vec_k); += semantics, column windows of wider rows, ragged last slab, widths from one lane to 1664 lanes; the scratch form
    is identical from run to run"""
    x = synth.normal(f"csv.{R}{C}", (R, ld))
    init = synth.normal(f"csv.o{C}", (C,))
    ref = init + bf(x)[:, :C].sum(0)
    xd = dev_bf(x)[:, :C]
    o = init.to(DEV).clone()
    ops.colsum(xd, o, R, C)
    assert rel(o, ref) < 1e-5
    scr = torch.full((1 << 22,), float("nan"), device=DEV)
    outs = []
    for _ in range(2):
        o = init.to(DEV).clone()
        ops.colsum(xd, o, R, C, scratch=scr)
        outs.append(o)
    assert rel(outs[0], ref) < 1e-5 and torch.equal(outs[0], outs[1])


def test_label_table_gradient_without_atomics(ops):
    """dl_cond_combine_bwd: nn.Embedding's backward (repeated labels add up) with one writer per table row and a fixed order:
    equal to index_add_ and identical from run to run"""
    B, E, n_cls = 64, 384, 10
    dact, emb = synth.normal("lt.dact", (B, E)).to(DEV), synth.normal("lt.emb", (B, E)).to(DEV)
    idx = synth.integers("lt.idx", (B,), n_cls).to(DEV)  # 64 samples over 10 classes: every row is hit several times
    g = dact * (torch.sigmoid(emb) * (1 + emb * (1 - torch.sigmoid(emb))))
    outs = []
    for _ in range(2):
        table = torch.ones(n_cls + 1, E, device=DEV)
        demb, demb16 = torch.empty(B, E, device=DEV), torch.empty(B, E, device=DEV, dtype=torch.bfloat16)
        ops.cond_combine_bwd(dact, emb, idx, demb, demb16, table)
        outs.append(table)
        assert rel(demb, g) < 1e-5
    ref = torch.ones(n_cls + 1, E, device=DEV).index_add_(0, idx, g)
    assert rel(outs[0], ref) < 1e-5 and torch.equal(outs[0], outs[1]) and torch.equal(outs[0][n_cls], torch.ones(E, device=DEV))
