"""plain NT GEMMs at mid-size row counts (M = 8192: CIFAR DiT at batch 32, DiT-B/REPA at batch 128 x 64 tokens) through the tile
dispatch of dl_gemm_nt (widest persistent tile that still fills >= 70 % of the CU rounds, else the 128x128 kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffulab_amd import ops

dev = "cuda"
def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M in (8192, 16384):
    for D in (512, 768):
        for name, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("mlp2", D, 4 * D), ("d_h", 4 * D, D), ("d_xm2", D, 8 * D), ("d_xm1", D, 3 * D)):
            a, b, out = rnd(M, K), rnd(N, K), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            us = timeit(lambda: ops.gemm_nt(a, b, out))
            print(f"M={M:6d} D={D} {name:6s} N={N:5d} K={K:5d}: {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s", flush=True)
