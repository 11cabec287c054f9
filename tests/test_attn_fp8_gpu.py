"""LAB kernel (libdiffulab_probe.so since round 4: no engine uses it, the product's config-5 attention is the bf16 kernel -- the fp8
forward broke even end to end once its quantisation pre-pass was counted and has no fp8 backward).
fp8 (e4m3) MFMA attention forward for the long joint sequences (BASELINE config 5; mmdit.py:172-190): semantics of the CDNA4
block-scaled matrix instruction, the quantisation pre-pass, and the kernel against an fp64 softmax attention on the same bf16
inputs.  Tolerance: e4m3 keeps 3 mantissa bits, so per-element products carry ~3 % noise that averages over 64-wide dot products and
hundreds of keys.  Stated tolerance: 6e-2 relative L2 of the attention output on WHITE-NOISE q, k, v (measured 5.3e-2: the worst
case -- the output is then an average of ~N independent values and every quantisation error is as large as the signal's own spread),
1e-1 on peaked attention patterns (scores with 3x the spread: the ~4 % rms error of an e4m3 dot product is multiplied by the score
scale before the exponential; measured 8.1e-2), lse within 6e-2 / 0.5 absolute."""

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def test_mfma_scale_f8_operand_layout_probe(probe_lib):
    g = torch.Generator().manual_seed(0)
    vals = torch.tensor([-4.0, -2.0, -1.5, -1.0, -0.5, 0.0, 0.5, 1.0, 1.5, 2.0, 3.0])
    a = vals[torch.randint(0, len(vals), (32, 64), generator=g)]
    b = vals[torch.randint(0, len(vals), (32, 64), generator=g)]
    a8, b8 = a.to(torch.float8_e4m3fn).to(DEV), b.to(torch.float8_e4m3fn).to(DEV)
    d = torch.empty(32, 32, device=DEV)
    assert probe_lib.dl_probe_mfma_f8(a8.data_ptr(), b8.data_ptr(), d.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(d.cpu(), a @ b.T)  # exact: every product and partial sum is representable


def _fp8(probe_lib):
    import ctypes

    v, q, f = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float
    probe_lib.dl_probe_attn_fp8_quantize.argtypes = [v] * 7 + [q] * 5 + [v]
    probe_lib.dl_probe_attn_fwd_fp8.argtypes = [v] * 6 + [q] * 5 + [f, v, v]
    st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
    p = lambda t: None if t is None else t.data_ptr()  # noqa: E731

    def quant(q_, k_, v_, q8, k8, v8t, sc, B, H, Nq, Nk):
        assert probe_lib.dl_probe_attn_fp8_quantize(p(q_), p(k_), p(v_), p(q8), p(k8), p(v8t), p(sc), B, H, Nq, Nk, 64, st()) == 0

    def fwd(q8, k8, v8t, sc, out, lse, B, H, Nq, Nk, dh, scale, bias):
        assert probe_lib.dl_probe_attn_fwd_fp8(p(q8), p(k8), p(v8t), p(sc), p(out), p(lse), B, H, Nq, Nk, dh, scale, p(bias), st()) == 0

    return quant, fwd


def _inputs(B, H, Nq, Nk, seed):
    g = torch.Generator().manual_seed(seed)
    mk = lambda n: (torch.randn(B, H, n, 64, generator=g)).to(torch.bfloat16).to(DEV)  # noqa: E731
    return mk(Nq), mk(Nk), mk(Nk)


def test_quantize_layouts_and_scales(probe_lib):
    quant, _ = _fp8(probe_lib)
    B, H, N = 2, 3, 256
    q, k, v = _inputs(B, H, N, N, 1)
    v = v * 3.0
    q8, k8 = (torch.empty(B, H, N, 64, device=DEV, dtype=torch.uint8) for _ in range(2))
    v8t = torch.empty(B, H, 64, N, device=DEV, dtype=torch.uint8)
    sc = torch.empty(B, H, 3, device=DEV)
    quant(q, k, v, q8, k8, v8t, sc, B, H, N, N)
    torch.cuda.synchronize()
    for i, t in enumerate((q, k, v)):
        amax = t.float().abs().amax(dim=(2, 3))
        assert torch.allclose(sc[:, :, i], amax / 448.0, rtol=1e-6)
    deq = lambda t8, s: t8.view(torch.float8_e4m3fn).float() * s[:, :, None, None]  # noqa: E731
    assert rel(deq(q8, sc[:, :, 0]), q.float()) < 4e-2 and rel(deq(k8, sc[:, :, 1]), k.float()) < 4e-2
    # v8t[b, h, c, blk*64 + p] = V[blk*64 + key(p), c]:  p = 32 hi + 16 t + r  <->  key = 32 t + (r & 3) + 8 (r >> 2) + 4 hi
    p = torch.arange(64)
    hi, t, r = p // 32, (p // 16) % 2, p % 16
    key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi
    perm = (torch.arange(N // 64)[:, None] * 64 + key[None, :]).flatten().to(DEV)
    want = v.float().transpose(2, 3)[:, :, :, perm]
    assert rel(deq(v8t, sc[:, :, 2]), want) < 4e-2


@pytest.mark.parametrize("shape", [(2, 3, 256, 512, False, 1.0), (2, 12, 1280, 1280, True, 1.0), (1, 2, 512, 256, True, 1.0),
                                   (2, 4, 512, 512, True, 3.0)])
def test_fp8_attention_forward_against_fp64_softmax(shape, probe_lib):
    from diffulab_amd import ops

    quant, fwd8 = _fp8(probe_lib)
    B, H, Nq, Nk, masked, sharp = shape
    q, k, v = _inputs(B, H, Nq, Nk, 7)
    q = (q.float() * sharp).to(torch.bfloat16)  # sharp > 1: peaked attention (a few keys carry each query)
    bias = None
    if masked:  # key-padding mask of the joint sequence: a ragged number of valid keys per sample
        valid = torch.tensor([Nk - 37 * (i + 1) for i in range(B)])
        bias = torch.where(torch.arange(Nk)[None, :] < valid[:, None], 0.0, float("-inf")).to(DEV)
    scale = 64**-0.5
    s = torch.einsum("bhqd,bhkd->bhqk", q.double(), k.double()) * scale
    if bias is not None:
        s = s + bias[:, None, None, :].double()
    ref = torch.einsum("bhqk,bhkd->bhqd", torch.softmax(s, -1), v.double()).transpose(1, 2).reshape(B, Nq, H * 64)
    ref_lse = torch.logsumexp(s, -1)
    q8, k8 = torch.empty(B, H, Nq, 64, device=DEV, dtype=torch.uint8), torch.empty(B, H, Nk, 64, device=DEV, dtype=torch.uint8)
    v8t, sc = torch.empty(B, H, 64, Nk, device=DEV, dtype=torch.uint8), torch.empty(B, H, 3, device=DEV)
    out, lse = torch.empty(B, Nq, H * 64, device=DEV, dtype=torch.bfloat16), torch.empty(B, H, Nq, device=DEV)
    quant(q, k, v, q8, k8, v8t, sc, B, H, Nq, Nk)
    fwd8(q8, k8, v8t, sc, out, lse, B, H, Nq, Nk, 64, scale, bias)
    out16, lse16 = torch.empty_like(out), torch.empty_like(lse)
    ops.attn_fwd_ex(q, k, v, out16, lse16, B, H, Nq, Nk, 64, scale, bias)
    torch.cuda.synchronize()
    e8, e16 = rel(out.float(), ref), rel(out16.float(), ref)
    print(f"fp8 attention {shape}: rel-L2 {e8:.3e} (bf16 kernel {e16:.3e}); max |lse - ref| {float((lse.double() - ref_lse).abs().max()):.3e}")
    assert e8 < (6e-2 if sharp == 1.0 else 1e-1)
    assert float((lse.double() - ref_lse).abs().max()) < (6e-2 if sharp == 1.0 else 0.5)
    assert bool(torch.isfinite(out.float()).all())


@pytest.mark.parametrize("shape", [(2, 12, 1280, 1280, True, 1.0), (2, 4, 512, 512, True, 3.0), (8, 12, 1280, 1280, True, 1.0)])
def test_fp8_qk_only_attention_forward_error_and_time(shape, probe_lib):
    """round 6 (the narrower attempt): ONLY QK^T on the fp8 matrix instruction, probabilities and V in bf16
    (`dl_probe_attn_fwd_fp8qk`).  Measured here, stated in DESIGN.md section 7: the error against an fp64 softmax attention on the
    same bf16 inputs -- white-noise and peaked scores -- next to the bf16 product kernel's and the all-fp8 kernel's, and the kernel time
    against `dl_attn_fwd_ex` at the joint sequence of BASELINE config 5 (1152 -> 1280 padded tokens, 12 heads).  MEASURED (round 6,
    profiles/r06_f_fp8_qk.txt): 4.0e-2 on white-noise scores -- AT the 4e-2 per-tensor bound the joint engines hold against the
    reference fixtures, with nothing left for the rest of the network --, 7.7e-2 on peaked ones (all-fp8: 5.4e-2 / 8.1e-2; the bf16
    kernel: 2.2e-3); kernel time 0.79x the bf16 kernel's at 24 (sample, head) pairs, 1.09x at 96 -- before the quantisation pre-pass
    (another 0.4x).  Neither the error nor the time supports it: the row stays closed with these numbers (asserted here as recorded
    bounds of the lab kernel, 4.5e-2 / 1e-1)."""
    import ctypes

    from diffulab_amd import ops

    quant, fwd8 = _fp8(probe_lib)
    v_, q_, f_ = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float
    probe_lib.dl_probe_attn_fwd_fp8qk.argtypes = [v_] * 6 + [q_] * 5 + [f_, v_, v_]
    B, H, Nq, Nk, masked, sharp = shape
    q, k, v = _inputs(B, H, Nq, Nk, 7)
    q = (q.float() * sharp).to(torch.bfloat16)
    valid = torch.tensor([Nk - 37 * (i + 1) for i in range(B)])
    bias = torch.where(torch.arange(Nk)[None, :] < valid[:, None], 0.0, float("-inf")).to(DEV)
    scale = 64**-0.5
    s = torch.einsum("bhqd,bhkd->bhqk", q.double(), k.double()) * scale + bias[:, None, None, :].double()
    ref = torch.einsum("bhqk,bhkd->bhqd", torch.softmax(s, -1), v.double()).transpose(1, 2).reshape(B, Nq, H * 64)
    q8, k8 = torch.empty(B, H, Nq, 64, device=DEV, dtype=torch.uint8), torch.empty(B, H, Nk, 64, device=DEV, dtype=torch.uint8)
    v8t, sc = torch.empty(B, H, 64, Nk, device=DEV, dtype=torch.uint8), torch.empty(B, H, 3, device=DEV)
    out8, outqk, out16 = (torch.empty(B, Nq, H * 64, device=DEV, dtype=torch.bfloat16) for _ in range(3))
    lse = torch.empty(B, H, Nq, device=DEV)
    st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
    quant(q, k, v, q8, k8, v8t, sc, B, H, Nq, Nk)
    fwd8(q8, k8, v8t, sc, out8, lse, B, H, Nq, Nk, 64, scale, bias)

    def fqk():
        assert probe_lib.dl_probe_attn_fwd_fp8qk(q8.data_ptr(), k8.data_ptr(), v.data_ptr(), sc.data_ptr(), outqk.data_ptr(), lse.data_ptr(),
                                                 B, H, Nq, Nk, 64, scale, bias.data_ptr(), st()) == 0

    def f16():
        ops.attn_fwd_ex(q, k, v, out16, lse, B, H, Nq, Nk, 64, scale, bias)

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    t_qk, t_16 = timeit(fqk), timeit(f16)
    t_quant = timeit(lambda: quant(q, k, v, q8, k8, v8t, sc, B, H, Nq, Nk))
    t_8 = timeit(lambda: fwd8(q8, k8, v8t, sc, out8, lse, B, H, Nq, Nk, 64, scale, bias))
    e_qk, e_8, e_16 = rel(outqk.float(), ref), rel(out8.float(), ref), rel(out16.float(), ref)
    print(f"fp8-QK-only attention {shape}: rel-L2 vs fp64 {e_qk:.3e} (all-fp8 {e_8:.3e}, bf16 kernel {e_16:.3e}); kernel {t_qk:.1f} us vs bf16 "
          f"{t_16:.1f} us (all-fp8 {t_8:.1f} us; quantisation pre-pass of q, k, v {t_quant:.1f} us)")
    assert bool(torch.isfinite(outqk.float()).all())
    assert e_qk < (4.5e-2 if sharp == 1.0 else 1e-1)
    assert e_qk <= e_8 * 1.05  # never worse than quantising P and V as well
