"""MLP backward with recomputed pre-activations vs the stored-u paths at the headline shape (M = 65536, D = 384, F = 1536).
    python scripts/mlp_rc_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffulab_amd import ops

dev, BF = "cuda", torch.bfloat16
M, D, F = 65536, 384, 1536
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
x = (torch.randn(M, D, device=dev) * 0.5).to(BF)
dt = (torch.randn(M, D, device=dev) * 0.5).to(BF)
w1 = torch.randn(2 * F, D, device=dev) * D**-0.5
w1p = torch.empty(2 * F, D, device=dev, dtype=BF)
ops.cast_weight_swiglu(w1, w1p)
w2t = (torch.randn(F, D, device=dev) * F**-0.5).to(BF)
u = torch.empty(M, 2 * F, device=dev, dtype=BF)
h = torch.empty(M, F, device=dev, dtype=BF)
dh = torch.empty(M, F, device=dev, dtype=BF)
du = torch.empty(M, 2 * F, device=dev, dtype=BF)
f_u = timeit(lambda: ops.gemm_nt_swiglu(x, w1p, u, h))
f_h = timeit(lambda: ops.gemm_nt_swiglu(x, w1p, None, h))
b_g = timeit(lambda: ops.gemm_nt(dt, w2t, dh))
b_s = timeit(lambda: ops.swiglu_bwd(dh, u, du))
b_r = timeit(lambda: ops.mlp_dswiglu_recompute(x, w1p, dt, w2t, du))
print(f"forward  : u and h stored {f_u:6.1f} us | h only {f_h:6.1f} us")
print(f"backward : dgrad GEMM {b_g:6.1f} + swiglu_bwd {b_s:6.1f} = {b_g + b_s:6.1f} us | "
      f"recompute {b_r:6.1f} us ({(2.0 * M * F * D + 4.0 * M * F * D) / b_r / 1e6:6.1f} TF/s over both GEMMs)")
print(f"fwd + bwd: stored {f_u + b_g + b_s:6.1f} us | recompute {f_h + b_r:6.1f} us")
