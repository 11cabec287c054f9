"""LAB: what the vendor library (hipBLASLt / rocBLAS behind torch.matmul) reaches on the step's GEMM shapes -- a calibration of the
headroom of the hand-written kernels, not part of the product (the product never calls a library GEMM).
    python scripts/lab/blaslt_calib.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffulab_amd import ops  # noqa: E402

dev, BF = "cuda", torch.bfloat16
M = 65536


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, N, K in (("qkv", 1152, 384), ("proj", 384, 384), ("mlp_up", 3072, 384), ("mlp_down", 384, 1536), ("d_xm2", 384, 3072),
                   ("d_xm1", 384, 1152), ("d_h", 1536, 384)):
    a = (torch.randn(M, K, device=dev) * 0.5).to(BF)
    w = (torch.randn(N, K, device=dev) * K**-0.5).to(BF)
    out = torch.empty(M, N, device=dev, dtype=BF)
    t_lib = timeit(lambda: torch.matmul(a, w.t(), out=out))
    t_own = timeit(lambda: ops.gemm_nt(a, w, out))
    fl = 2.0 * M * N * K
    print(f"NT {name:9s} N={N:5d} K={K:5d}: library {t_lib:7.1f} us {fl / t_lib / 1e6:7.1f} TF/s | dl_gemm_nt {t_own:7.1f} us {fl / t_own / 1e6:7.1f} TF/s")
for name, Mo, No in (("w_qkv", 1152, 384), ("w_proj", 384, 384), ("w_up", 3072, 384), ("w_down", 384, 1536)):
    a = (torch.randn(M, Mo, device=dev) * 0.5).to(BF)
    b = (torch.randn(M, No, device=dev) * 0.5).to(BF)
    out = torch.empty(Mo, No, device=dev, dtype=torch.float32)
    outb = torch.empty(Mo, No, device=dev, dtype=BF)
    t_lib = timeit(lambda: torch.matmul(a.t(), b, out=outb))
    t_own = timeit(lambda: ops.gemm_tn(a, b, out))
    fl = 2.0 * M * Mo * No
    print(f"TN {name:9s} [{Mo:4d} x {No:4d}]: library {t_lib:7.1f} us {fl / t_lib / 1e6:7.1f} TF/s (bf16 out) | dl_gemm_tn {t_own:7.1f} us {fl / t_own / 1e6:7.1f} TF/s")
