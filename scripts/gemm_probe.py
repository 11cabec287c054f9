#!/usr/bin/env python
"""component-stripping probe of the persistent GEMM kernels (dl_probe_gemm_set): which of {epilogue stores / atomics, MFMA,
operand DMA, LDS fragment reads} a launch is waiting for.  HIP-event timing, random operands, DiT-S/2 B=256 shapes.
    python scripts/gemm_probe.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffulab_amd import ops
from diffulab_amd._lib import lib

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M = 65536
dev = "cuda"
torch.manual_seed(0)
L = lib()


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


def timeit(fn):
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


NAMES = {0: "full", 1: "-stores", 2: "-mfma", 4: "-dma", 8: "-ldsread", 5: "-stores-dma", 10: "-mfma-ldsread", 16: "+pf", 17: "+pf-stores", 26: "+pf-mfma-ldsread", 32: "xcd-slabs"}
for name, N, K in [][:0] or [("qkv", 1152, 384), ("d_h", 1536, 384), ("d_xm2", 384, 3072), ("d_xm1", 384, 1152), ("d_a", 384, 384)]:
    a, b = rnd(M, K), rnd(N, K)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    row = []
    for fl in (0, 1, 2, 4, 8, 5, 10):
        L.call("dl_probe_gemm_set", fl)
        row.append(f"{NAMES[fl]} {timeit(lambda: ops.gemm_nt(a, b, out)):7.1f}")
    L.call("dl_probe_gemm_set", 0)
    print(f"nt {name:6s} N={N:5d} K={K:5d} us: " + " | ".join(row), flush=True)
for name, Mo, No in [("w_mlp1", 3072, 384), ("w_mlp2", 384, 1536)]:
    a, b = rnd(M, Mo), rnd(M, No)
    c = torch.zeros(Mo, No, device=dev)
    ws = torch.zeros(8 * Mo * No, device=dev)
    row = []
    for fl, nm in ((0, "wide full"), (2, "wide -fold"), (3, "wide -fold-atomics")):
        L.call("dl_probe_gemm_set", fl)
        row.append(f"{nm} {timeit(lambda: ops.gemm_tn(a, b, c, ws=ws)):7.1f}")
    L.call("dl_probe_gemm_set", 0)
    print(f"tn {name:6s} M={Mo:5d} N={No:5d} us: " + " | ".join(row), flush=True)
