"""LAB: NT GEMMs of the CIFAR DiT configuration (B = 32: 8192 token rows, D = 512): which kernel should take them?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffulab_amd import ops
dev, BF = "cuda", torch.bfloat16
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
tag = os.environ.get("DIFFULAB_HIP_LIB", "product").split("_hip_")[-1]
for M in (8192, 16384):
    for N, K in ((512, 512), (512, 1536), (512, 2048), (512, 4096), (1536, 512), (2048, 512)):
        a = (torch.randn(M, K, device=dev) * 0.5).to(BF); b = (torch.randn(N, K, device=dev) * K**-0.5).to(BF)
        c = torch.empty(M, N, device=dev, dtype=BF); r = torch.randn(M, N, device=dev).to(BF)
        us = timeit(lambda: ops.gemm_nt(a, b, c))
        us_r = timeit(lambda: ops.gemm_nt(a, b, c, resid=r))
        print(f"{tag:10s} M={M:5d} N={N:4d} K={K:4d}: plain {us:6.1f} us {2.0*M*N*K/us/1e6:6.1f} TF/s | +resid {us_r:6.1f} us")
