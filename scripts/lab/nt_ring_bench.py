"""LAB: the wide persistent NT GEMM (gemm_nt_big_k) at the headline step's shapes; run once per library variant
(DIFFULAB_HIP_LIB=diffulab_amd/csrc/build/libdiffulab_hip_<variant>.so), checks the result against torch"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffulab_amd import ops
dev, BF = "cuda", torch.bfloat16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
tag = os.environ.get("DIFFULAB_HIP_LIB", "product").split("_hip_")[-1]
M = 65536
for N, K in ((1152, 384), (384, 384), (3072, 384), (768, 1536), (1536, 1536), (384, 1536)):
    a = (torch.randn(M, K, device=dev) * 0.5).to(BF); b = (torch.randn(N, K, device=dev) * K**-0.5).to(BF)
    c = torch.empty(M, N, device=dev, dtype=BF)
    us = timeit(lambda: ops.gemm_nt(a, b, c))
    ref = (a[:4096].float() @ b.float().t())
    err = ((c[:4096].float() - ref).norm() / ref.norm()).item()
    ref2 = (a[-256:].float() @ b.float().t()); err2 = ((c[-256:].float() - ref2).norm() / ref2.norm()).item()
    print(f"{tag:12s} NT  M={M} N={N:4d} K={K:4d}: {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF/s  rel err {err:.2e} {err2:.2e}")
D, F = 384, 1536
x = (torch.randn(M, D, device=dev) * 0.5).to(BF)
w1 = torch.randn(2 * F, D, device=dev) * D**-0.5
w1p = torch.empty(2 * F, D, device=dev, dtype=BF); ops.cast_weight_swiglu(w1, w1p)
h = torch.empty(M, F, device=dev, dtype=BF); u = torch.empty(M, 2 * F, device=dev, dtype=BF)
us = timeit(lambda: ops.gemm_nt_swiglu(x, w1p, None, h))
uu = x[:2048].float() @ w1.to(BF).float().t()
href = torch.nn.functional.silu(uu[:, :F]) * uu[:, F:]
err = ((h[:2048].float() - href).norm() / href.norm()).item()
print(f"{tag:12s} SwiGLU up (h only): {us:7.1f} us {2.0*M*2*F*D/us/1e6:7.1f} TF/s  rel err {err:.2e}")
us = timeit(lambda: ops.gemm_nt_swiglu(x, w1p, u, h))
print(f"{tag:12s} SwiGLU up (u + h):  {us:7.1f} us")
