"""3x3 implicit-GEMM convolution forward alone at the MNIST-DDPM UNet's high-resolution shapes (B=128):
    DL_CONV_BIG=0|1 python scripts/conv_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffulab_amd import ops

dev, BF = "cuda", torch.bfloat16
B = 128
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
r64 = lambda v: (v + 63) // 64 * 64
zero = torch.zeros(64, device=dev, dtype=BF)
for H, ci, co in ((32, 128, 128), (32, 256, 128), (16, 256, 256), (16, 128, 256), (16, 512, 256), (8, 512, 512)):
    M = B * H * H
    x = torch.randn(M, ci, device=dev).to(BF)
    wf = (torch.randn(co, r64(9 * ci), device=dev) * 0.02).to(BF)
    out = torch.empty(M, co, device=dev, dtype=BF)
    bias = torch.randn(co, device=dev)
    res = torch.randn(M, co, device=dev).to(BF)
    us = timeit(lambda: ops.conv3x3_nt(x, B, H, H, ci, wf, out, co, bias, res, zero))
    print(f"{H}x{H} Ci={ci:4d} Co={co:4d}: {us:7.1f} us  {2.0 * M * co * 9 * ci / us / 1e6:7.1f} TF/s")
