"""Standalone timing of the row-complete GEMMs (csrc/gemm_ln.hip) against the launch pairs they replace, at the DiT-S/2
training shapes (B = 256 samples x 256 tokens, D = 384).  Prints one line per site: unfused pair (GEMM + row kernel), fused launch.

    python scripts/row_gemm_bench.py [B]
"""

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffulab_amd import ops  # noqa: E402

DEV = "cuda"
D, N, H = 384, 256, 6


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    M = B * N
    bf = torch.bfloat16
    rn = lambda *s, std=1.0: (torch.randn(*s, device=DEV) * std).to(bf)  # noqa: E731
    mod = rn(B, 6 * D, std=0.3)
    lw, lb = 1 + 0.1 * torch.randn(D, device=DEV), 0.1 * torch.randn(D, device=DEV)
    x, t = rn(M, D), rn(M, D)
    o1, o2, o3, o4 = (torch.empty(M, D, device=DEV, dtype=bf) for _ in range(4))
    mu, rs = torch.zeros(M, device=DEV), torch.ones(M, device=DEV)
    dmod = torch.zeros(B, 6 * D, device=DEV)
    dwb = torch.zeros(B, 2, D, device=DEV)
    print(f"B={B} M={M}")
    for name, K in (("proj  -> LN2 (K=384)", 384), ("mlp2  -> LN1 (K=1536)", 1536)):
        a, w = rn(M, K), rn(D, K, std=K**-0.5)

        def pair():
            ops.gemm_nt(a, w, o1)
            ops.ln_modulate_fwd(x, lw, lb, mod[:, :D], mod[:, D : 2 * D], N, 1e-5, o3, mu, rs, t=o1, gate=mod[:, 2 * D : 3 * D], x_out=o2)

        def gemm_only():
            ops.gemm_nt(a, w, o1)

        def fused():
            ops.ln_modulate_gemm_fwd(a, w, x, mod[:, 2 * D : 3 * D], lw, lb, mod[:, :D], mod[:, D : 2 * D], N, 1e-5, o1, o2, o3, mu, rs)

        print(f"fwd {name}: gemm {timeit(gemm_only):7.1f}  pair {timeit(pair):7.1f}  fused {timeit(fused):7.1f} us")
    for name, K in (("dqkv -> LN1 bwd (K=1152)", 1152), ("du   -> LN2 bwd (K=3072)", 3072), ("dO   -> LNf bwd (K=64)", 64)):
        a, w = rn(M, K), rn(D, K, std=K**-0.5)

        def pair():
            ops.gemm_nt(a, w, o1)
            ops.ln_modulate_bwd(o1, x, lw, lb, mod[:, :D], N, mu, rs, o4, o2, dmod[:, :D], dmod[:, D : 2 * D], dwb, gate_t=t,
                                gate=mod[:, 2 * D : 3 * D], dt=o3, dgate=dmod[:, 2 * D : 3 * D])

        def gemm_only():
            ops.gemm_nt(a, w, o1)

        def fused():
            ops.ln_modulate_gemm_bwd(a, w, x, lw, lb, mod[:, :D], N, mu, rs, o4, o2, dmod[:, :D], dmod[:, D : 2 * D], dwb, gate_t=t,
                                     gate=mod[:, 2 * D : 3 * D], dt=o3, dgate=dmod[:, 2 * D : 3 * D])

        print(f"bwd {name}: gemm {timeit(gemm_only):7.1f}  pair {timeit(pair):7.1f}  fused {timeit(fused):7.1f} us")
    a, w = rn(M, D), rn(3 * D, D, std=D**-0.5)
    qkv = torch.empty(M, 3 * D, device=DEV, dtype=bf)
    q, k = (torch.empty(B, H, N, 64, device=DEV, dtype=bf) for _ in range(2))
    rr = torch.zeros(M, 2, device=DEV)
    sq, sk = torch.ones(D, device=DEV), torch.ones(D, device=DEV)
    from diffulab_amd.engine import rope_grid_tables

    cos, sin = (z.to(DEV) for z in rope_grid_tables(16, 16, [32, 32], 10_000.0))

    def pair():
        ops.gemm_nt(a, w, qkv)
        ops.qk_norm_rope_fwd(qkv, sq, sk, cos, sin, q, k, None, rr, B, N, H, 64, 64)

    def gemm_only():
        ops.gemm_nt(a, w, qkv)

    def fused():
        ops.gemm_nt_qk_norm_rope(a, w, sq, sk, cos, sin, qkv, q, k, rr, B, N, H, 64, 64)

    print(f"fwd qkv   -> QK-norm + RoPE: gemm {timeit(gemm_only):7.1f}  pair {timeit(pair):7.1f}  fused {timeit(fused):7.1f} us")


if __name__ == "__main__":
    main()
