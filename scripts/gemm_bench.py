#!/usr/bin/env python
"""micro-benchmark of the two MFMA GEMM kernels at the DiT-S/2 (B=256) shapes; HIP-event timing, random operands.
    python scripts/gemm_bench.py [nt|tn|all] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffulab_amd import ops

which = sys.argv[1] if len(sys.argv) > 1 else "all"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
M = 65536
dev = "cuda"
torch.manual_seed(0)

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
    return e0.elapsed_time(e1) / iters * 1e3  # us

if which in ("nt", "all"):
    for name, N, K, kw in [("qkv", 1152, 384, {}), ("proj+gate", 384, 384, "gate"), ("mlp1", 3072, 384, {}),
                           ("mlp2+gate", 384, 1536, "gate"), ("d_h", 1536, 384, {}), ("d_xm2", 384, 3072, {}),
                           ("d_xm1", 384, 1152, {}), ("d_a", 384, 384, {})]:
        a, b = rnd(M, K), rnd(N, K)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        if kw == "gate":
            res, gate, pre = rnd(M, N), rnd(256, N), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            fn = lambda: ops.gemm_nt(a, b, out, pre_out=pre, resid=res, gate=gate, rows_per_gate=256)
        else:
            fn = lambda: ops.gemm_nt(a, b, out)
        us = timeit(fn)
        err = ""
        if kw != "gate" and os.environ.get("GEMM_BENCH_CHECK"):
            ref = a.float() @ b.float().T
            err = f"  max|err| {(out.float() - ref).abs().max().item():.3e} (ref max {ref.abs().max().item():.1f})"
            del ref
        print(f"nt {name:10s} M={M} N={N:5d} K={K:5d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s{err}")
if which in ("nt", "all", "swiglu"):
    a, wp = rnd(M, 384), rnd(3072, 384)
    u, h = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16), torch.empty(M, 1536, device=dev, dtype=torch.bfloat16)
    us = timeit(lambda: ops.gemm_nt_swiglu(a, wp, u, h))
    print(f"nt mlp1+swiglu (U and H stored)      : {us:8.1f} us  {2.0*M*3072*384/us/1e6:7.1f} TF/s")
    us = timeit(lambda: ops.gemm_nt_swiglu(a, wp, None, h))
    print(f"nt mlp1+swiglu (H only, inference)   : {us:8.1f} us  {2.0*M*3072*384/us/1e6:7.1f} TF/s")
if which in ("nt", "all", "swiglu"):
    dt2, w2t = rnd(M, 384), rnd(1536, 384)
    uu, du = rnd(M, 3072), torch.empty(M, 3072, device=dev, dtype=torch.bfloat16)
    dh = torch.empty(M, 1536, device=dev, dtype=torch.bfloat16)
    us_g = timeit(lambda: ops.gemm_nt(dt2, w2t, dh))
    us_s = timeit(lambda: ops.swiglu_bwd(dh, uu, du))
    print(f"nt d_h GEMM {us_g:.1f} + swiglu_bwd {us_s:.1f} = {us_g + us_s:.1f} us")
if which in ("tn", "all"):
    for name, Mo, No in [("w_qkv", 1152, 384), ("w_proj", 384, 384), ("w_mlp1", 3072, 384), ("w_mlp2", 384, 1536)]:
        a, b = rnd(M, Mo), rnd(M, No)
        c = torch.zeros(Mo, No, device=dev)
        us = timeit(lambda: ops.gemm_tn(a, b, c))
        print(f"tn {name:10s} R={M} M={Mo:5d} N={No:5d}: {us:8.1f} us  {2.0*M*Mo*No/us/1e6:7.1f} TF/s")
