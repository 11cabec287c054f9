"""Row-complete GEMMs (csrc/gemm_ln.hip) against the launch pairs they replace, THROUGH THE C ABI.

The fused kernels run the LayerNorm-modulate / QK-norm row kernels as the epilogue of the GEMM that produces their input
(reference chain: DiTBlock._forward mmdit.py:288-309, DiTAttention.forward mmdit.py:81-91).  The unfused pair
(dl_gemm_nt + dl_ln_modulate_* / dl_qk_norm_rope_fwd) is itself pinned against the oracle in tests/test_kernels_gpu.py, so the
bar here is stronger than a tolerance:
  * the GEMM result, the residual stream and everything that is pure data movement (t, x, qkv) must EQUAL the unfused pair bit for
    bit; row statistics agree to 1e-6 (the compiler contracts the same expressions into different fma chains in the two kernels)
    and the bf16 rows computed from them (xm, dx, dt, q, k) differ in at most a few elements per thousand, by one bf16 ulp;
  * per-sample column sums (dscale, dshift, dgate, LayerNorm-affine partials) are summed in another order (fixed, no atomics):
    1e-5 relative against the unfused pair, and two runs must give identical bits (the unfused pair's atomics do not).
"""

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dit as odit  # noqa: E402
from oracle import synth  # noqa: E402

DEV = "cuda"
D, N = 384, 256


@pytest.fixture(scope="module")
def ops():
    from diffulab_amd import _lib, ops as _ops

    assert _lib.available(), "libdiffulab_hip.so missing on the GPU box"
    return _ops


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def dev_bf(x):
    return x.to(torch.bfloat16).to(DEV).contiguous()


def bits(t):
    return t.view(torch.int16) if t.dtype == torch.bfloat16 else t.view(torch.int32)


def same(a, b):
    return torch.equal(bits(a), bits(b))


def close_f32(a, b, tol=1e-6):
    return bool(((a - b).abs() <= tol * b.abs().clamp_min(1e-3)).all())


def same_to_an_ulp(a, b, frac=5e-3):
    """bf16 tensors: equal, except that at most `frac` of the elements may differ by one bf16 ulp (2^-8 relative: two roundings of
    f32 values that differ in their last bits)"""
    a32, b32 = a.float(), b.float()
    bad = a32 != b32
    if not bool(bad.any()):
        return True
    ulp = (a32 - b32).abs() <= b32.abs().clamp_min(1e-30) * 2.0**-6
    return bool(ulp[bad].all()) and float(bad.float().mean()) <= frac


@pytest.mark.parametrize("B,K,resid,gate,affine", [(8, 384, True, True, True), (16, 1536, True, True, True), (8, 64, False, False, True),
                                                   (8, 1536, True, True, False), (24, 384, True, False, True)])
def test_ln_modulate_gemm_fwd_equals_gemm_then_row_kernel(ops, B, K, resid, gate, affine):
    M = B * N
    a = dev_bf(synth.normal("rg.a", (M, K), std=1.0))
    w = dev_bf(synth.normal("rg.w", (D, K), std=K**-0.5))
    x = dev_bf(synth.normal("rg.x", (M, D)))
    mod = dev_bf(synth.normal("rg.mod", (B, 3 * D), std=0.3))
    lw = (1 + synth.normal("rg.lw", (D,), std=0.1)).to(DEV) if affine else None
    lb = synth.normal("rg.lb", (D,), std=0.1).to(DEV) if affine else None
    eps = 1e-5 if affine else 1e-6
    sc, sh, gt = mod[:, :D], mod[:, D : 2 * D], mod[:, 2 * D :]

    def bufs():
        return [torch.full((M, D), 7.0, device=DEV, dtype=torch.bfloat16) for _ in range(3)] + [torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)]

    # unfused: GEMM -> t, then the row kernel (with the gated residual folded in, as the engine issues it)
    t0, x0, xm0, mu0, rs0 = bufs()
    ops.gemm_nt(a, w, t0)
    if resid:
        g_rows = gt if gate else torch.ones_like(gt)
        ops.ln_modulate_fwd(x, lw, lb, sc, sh, N, eps, xm0, mu0, rs0, t=t0, gate=g_rows, x_out=x0)
    else:
        ops.ln_modulate_fwd(t0, lw, lb, sc, sh, N, eps, xm0, mu0, rs0)
        x0.copy_(t0)
    t1, x1, xm1, mu1, rs1 = bufs()
    assert ops.ln_modulate_gemm_fwd(a, w, x if resid else None, gt if (resid and gate) else None, lw, lb, sc, sh, N, eps, t1, x1, xm1,
                                    mu1, rs1)
    torch.cuda.synchronize()
    assert same(t1, t0), rel(t1.float(), t0.float())
    assert same(x1, x0), rel(x1.float(), x0.float())
    assert close_f32(mu1, mu0) and close_f32(rs1, rs0)
    assert same_to_an_ulp(xm1, xm0), rel(xm1.float(), xm0.float())


@pytest.mark.parametrize("B,K,dres,gated,affine", [(8, 1152, True, True, True), (16, 3072, True, True, True), (8, 64, False, True, False),
                                                   (8, 1152, True, False, True)])
def test_ln_modulate_gemm_bwd_equals_gemm_then_row_kernel(ops, B, K, dres, gated, affine):
    M = B * N
    a = dev_bf(synth.normal("rb.a", (M, K), std=1.0))
    wt = dev_bf(synth.normal("rb.w", (D, K), std=K**-0.5))
    x = dev_bf(synth.normal("rb.x", (M, D)))
    dr = dev_bf(synth.normal("rb.dres", (M, D))) if dres else None
    tg = dev_bf(synth.normal("rb.tg", (M, D))) if gated else None
    mod = dev_bf(synth.normal("rb.mod", (B, 3 * D), std=0.3))
    lw = (1 + synth.normal("rb.lw", (D,), std=0.1)).to(DEV) if affine else None
    lb = synth.normal("rb.lb", (D,), std=0.1).to(DEV) if affine else None
    mean = x.float().mean(-1).contiguous()
    rstd = (x.float().var(-1, unbiased=False) + 1e-5).rsqrt().contiguous()

    def run(fused):
        dx = torch.full((M, D), 7.0, device=DEV, dtype=torch.bfloat16)
        dt = torch.full((M, D), 7.0, device=DEV, dtype=torch.bfloat16)
        dmod = torch.zeros(B, 3 * D, device=DEV)
        dwb = torch.zeros(B, 2, D, device=DEV) if affine else None
        kw = dict(gate_t=tg, gate=mod[:, 2 * D :], dt=dt, dgate=dmod[:, 2 * D :]) if gated else {}
        if fused:
            assert ops.ln_modulate_gemm_bwd(a, wt, x, lw, lb, mod[:, :D], N, mean, rstd, dr, dx, dmod[:, :D], dmod[:, D : 2 * D], dwb, **kw)
        else:
            dxm = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
            ops.gemm_nt(a, wt, dxm)
            ops.ln_modulate_bwd(dxm, x, lw, lb, mod[:, :D], N, mean, rstd, dr, dx, dmod[:, :D], dmod[:, D : 2 * D], dwb, **kw)
        torch.cuda.synchronize()
        return dx, dt, dmod, dwb

    dx0, dt0, dmod0, dwb0 = run(False)
    dx1, dt1, dmod1, dwb1 = run(True)
    assert same_to_an_ulp(dx1, dx0), rel(dx1.float(), dx0.float())
    if gated:
        assert same_to_an_ulp(dt1, dt0), rel(dt1.float(), dt0.float())
    assert rel(dmod1, dmod0) < 1e-5
    if affine:
        assert rel(dwb1, dwb0) < 1e-5
    # deterministic: a second fused run reproduces every bit of the per-sample sums
    _, _, dmod2, dwb2 = run(True)
    assert same(dmod2, dmod1) and (not affine or same(dwb2, dwb1))


@pytest.mark.parametrize("B,gh,gw", [(8, 16, 16), (32, 8, 8), (12, 16, 16)])
def test_gemm_nt_qk_norm_rope_equals_gemm_then_row_kernel(ops, B, gh, gw):
    H, dh = 6, 64
    Nt = gh * gw
    M = B * Nt
    a = dev_bf(synth.normal("rq.a", (M, D)))
    w = dev_bf(synth.normal("rq.w", (3 * D, D), std=D**-0.5))
    sq = (1 + synth.normal("rq.sq", (D,), std=0.1)).to(DEV)
    sk = (1 + synth.normal("rq.sk", (D,), std=0.1)).to(DEV)
    cos, sin = (t.to(DEV) for t in odit.rope_tables(gh, gw, [32, 32], 10_000.0))

    qkv0 = torch.empty(M, 3 * D, device=DEV, dtype=torch.bfloat16)
    q0, k0 = (torch.full((B, H, Nt, dh), 7.0, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    r0 = torch.zeros(M, 2, device=DEV)
    ops.gemm_nt(a, w, qkv0)
    ops.qk_norm_rope_fwd(qkv0, sq, sk, cos, sin, q0, k0, None, r0, B, Nt, H, dh, 64)
    qkv1 = torch.full((M, 3 * D), 7.0, device=DEV, dtype=torch.bfloat16)
    q1, k1 = (torch.full((B, H, Nt, dh), 7.0, device=DEV, dtype=torch.bfloat16) for _ in range(2))
    r1 = torch.zeros(M, 2, device=DEV)
    assert ops.gemm_nt_qk_norm_rope(a, w, sq, sk, cos, sin, qkv1, q1, k1, r1, B, Nt, H, dh, 64)
    torch.cuda.synchronize()
    assert same(qkv1, qkv0), rel(qkv1.float(), qkv0.float())
    assert close_f32(r1, r0)
    assert same_to_an_ulp(q1, q0), rel(q1.float(), q0.float())
    assert same_to_an_ulp(k1, k0), rel(k1.float(), k0.float())


def test_row_gemms_decline_other_shapes(ops):
    """D != 384, a modulation group that is not one tile, a ragged row count: DL_ERR_UNSUPPORTED (the engine then issues the pair)"""
    M = 8 * N
    a, w = dev_bf(torch.zeros(M, 512)), dev_bf(torch.zeros(512, 512))
    mod = dev_bf(torch.zeros(8, 1024))
    o = torch.empty(M, 512, device=DEV, dtype=torch.bfloat16)
    mu = torch.empty(M, device=DEV)
    assert not ops.ln_modulate_gemm_fwd(a, w, None, None, None, None, mod[:, :512], mod[:, 512:], N, 1e-5, None, o, o, mu, mu)
    a, w = dev_bf(torch.zeros(M, D)), dev_bf(torch.zeros(D, D))
    o = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    assert not ops.ln_modulate_gemm_fwd(a, w, None, None, None, None, mod[:, :D], mod[:, D : 2 * D], 64, 1e-5, None, o, o, mu, mu)


@pytest.mark.parametrize("B,gh,gw,axes", [(24, 16, 16, [32, 32]), (40, 16, 16, [32, 32]), (24, 16, 16, [16, 16]), (24, 16, 16, None)])
def test_qkv_gemm_with_row_statistics_and_attention_with_qk_norm_on_load(ops, B, gh, gw, axes):
    """round 4: dl_gemm_nt_ssq + dl_attn_fwd_qkn (QK-RMSNorm statistics from the qkv GEMM's epilogue, norm + RoPE applied as the
    attention stages q and k) against the sequence they replace, dl_gemm_nt -> dl_qk_norm_rope_fwd -> dl_attn_fwd_sv: qkv bit for
    bit, rrms to 1e-6, the normalised q / k to a bf16 ulp on a few elements per thousand, the attention output and lse to the
    accuracy those ulps allow; two runs give identical bits (two addends per statistics word).  axes = rope_axes_dim: [16, 16]
    rotates 32 of the 64 channels of a head (tables [N, 16]: the loads of the un-rotated channels must not touch them), None = no
    rotary embedding at all (NULL tables)"""
    H, dh = 6, 64
    Nt = gh * gw
    M = B * Nt
    a = dev_bf(synth.normal("qn.a", (M, D)))
    w = dev_bf(synth.normal("qn.w", (3 * D, D), std=D**-0.5))
    sq = (1 + synth.normal("qn.sq", (D,), std=0.1)).to(DEV)
    sk = (1 + synth.normal("qn.sk", (D,), std=0.1)).to(DEV)
    rot = sum(axes) if axes else 0
    # (the tables are allocated exactly [N, rot / 2]: a read past the end of the last row faults or meets the guard value)
    cos, sin = (t.to(DEV).contiguous() for t in odit.rope_tables(gh, gw, axes, 10_000.0)) if axes else (None, None)
    bf = dict(device=DEV, dtype=torch.bfloat16)

    qkv0 = torch.empty(M, 3 * D, **bf)
    q0, k0 = (torch.empty(B, H, Nt, dh, **bf) for _ in range(2))
    r0, o0, l0 = torch.zeros(M, 2, device=DEV), torch.empty(M, D, **bf), torch.empty(B, H, Nt, device=DEV)
    ops.gemm_nt(a, w, qkv0)
    dummy = torch.zeros(8, device=DEV)  # (the unfused pass wants non-NULL tables even when it rotates nothing)
    ops.qk_norm_rope_fwd(qkv0, sq, sk, cos if axes else dummy, sin if axes else dummy, q0, k0, None, r0, B, Nt, H, dh, rot)
    ops.attn_fwd_qkv(q0, k0, qkv0, o0, l0, B, H, Nt, dh, dh**-0.5)

    def fused():
        qkv1 = torch.full((M, 3 * D), 7.0, **bf)
        q1, k1 = (torch.full((B, H, Nt, dh), 7.0, **bf) for _ in range(2))
        ssq = torch.zeros(M, 2, device=DEV)
        r1, o1, l1 = torch.zeros(M, 2, device=DEV), torch.empty(M, D, **bf), torch.empty(B, H, Nt, device=DEV)
        assert ops.gemm_nt_ssq(a, w, qkv1, ssq)
        ops.attn_fwd_qkn(qkv1, ssq, sq, sk, cos, sin, q1, k1, r1, o1, l1, B, H, Nt, dh, rot, dh**-0.5)
        torch.cuda.synchronize()
        return qkv1, ssq, q1, k1, r1, o1, l1

    qkv1, ssq, q1, k1, r1, o1, l1 = fused()
    assert same(qkv1, qkv0)
    ref_ssq = torch.stack([qkv0[:, :D].float().square().sum(1), qkv0[:, D : 2 * D].float().square().sum(1)], 1)
    assert rel(ssq, ref_ssq) < 1e-6
    assert close_f32(r1, r0, 2e-6)
    # (the statistics are summed in another order than the row kernel's wave reduction: r differs in its last bits, so a few per cent
    # of the normalised elements may land on the neighbouring bf16)
    for got, want in ((q1, q0), (k1, k0)):
        g32, w32 = got.float(), want.float()
        diff = (g32 - w32).abs()
        # at most a few per cent of the elements on the neighbouring bf16, by one ulp of the element or -- where the rotation
        # a c - b s cancels -- of the operands' magnitude
        assert float((diff > 0).float().mean()) < 5e-2 and bool((diff <= 2.0**-6 * w32.abs().clamp_min(0.05 * float(w32.abs().mean()))).all())
        assert rel(g32, w32) < 1e-4
    assert rel(o1.float(), o0.float()) < 2e-3 and float((l1 - l0).abs().max()) < 2e-3
    again = fused()
    assert all(same(x, y) for x, y in zip((qkv1, ssq, q1, k1, r1, o1, l1), again))
    # shapes without the persistent 384-wide tiling decline (the engine keeps the launch pair)
    small = dev_bf(torch.zeros(8 * 256, D))
    assert not ops.gemm_nt_ssq(small, w, torch.empty(8 * 256, 3 * D, **bf), torch.zeros(8 * 256, 2, device=DEV))


@pytest.mark.parametrize("B,H,axes,train", [(256, 6, [32, 32], True), (130, 6, [32, 32], True), (131, 6, [16, 16], False), (128, 6, None, True),
                                            (100, 8, [32, 32], True), (65, 12, [32, 32], False)])  # (512- and 768-wide models)
def test_pipelined_attention_forward_equals_the_chain_form_bit_for_bit(ops, B, H, axes, train):
    """round 6: from three (sample, head) items per CU dl_attn_fwd_qkn runs `attn_fwd_qkn_pipe_k` -- one persistent workgroup per CU,
    K / V tiles double-buffered, the next item's loads in flight under the current item's MFMAs, O stored one item late -- with the
    arithmetic of the chain form `attn_fwd_qkn_k` in the same order: every output (O, lse, the normalised q / k, rrms) bit for bit
    equal to the chain form (the lab switch dl_lab_set_attn_pipe selects it).  B = 130 / 131: item runs that end inside a sample and
    workgroups with a shorter last run; train = False: the inference form (no q / k / rrms outputs)."""
    dh, Nt = 64, 256
    D = H * dh  # noqa: N806 (shadows the module's 384 for the wider models)
    M = B * Nt
    bf = dict(device=DEV, dtype=torch.bfloat16)
    qkv = dev_bf(synth.normal("pp.qkv", (M, 3 * D)))
    ssq = torch.stack([qkv[:, :D].float().square().sum(1), qkv[:, D : 2 * D].float().square().sum(1)], 1).contiguous()
    sq = (1 + synth.normal("pp.sq", (D,), std=0.1)).to(DEV)
    sk = (1 + synth.normal("pp.sk", (D,), std=0.1)).to(DEV)
    rot = sum(axes) if axes else 0
    cos, sin = (t.to(DEV).contiguous() for t in odit.rope_tables(16, 16, axes, 10_000.0)) if axes else (None, None)

    def run(mode):
        ops.lib().cdll.dl_lab_set_attn_pipe(mode)
        try:
            q1, k1 = (torch.full((B, H, Nt, dh), 7.0, **bf) for _ in range(2)) if train else (None, None)
            r1 = torch.full((M, 2), 7.0, device=DEV) if train else None
            o1, l1 = torch.full((M, D), 7.0, **bf), torch.full((B, H, Nt), 7.0, device=DEV)
            ops.attn_fwd_qkn(qkv, ssq, sq, sk, cos, sin, q1, k1, r1, o1, l1, B, H, Nt, dh, rot, dh**-0.5)
            torch.cuda.synchronize()
        finally:
            ops.lib().cdll.dl_lab_set_attn_pipe(1)
        return [t for t in (q1, k1, r1, o1, l1) if t is not None]

    chain, pipe, again = run(0), run(1), run(1)
    assert all(same(x, y) for x, y in zip(chain, pipe))
    assert all(same(x, y) for x, y in zip(pipe, again))
    assert float(pipe[-2].float().abs().max()) < 7.0  # (every O row was written)
