"""LAB: mlp_dswiglu_rc_k alone at the headline shape (run once per library variant built by scripts/lab/build_variant.sh)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
x = (torch.randn(M, D, device=dev) * 0.5).to(BF); dt = (torch.randn(M, D, device=dev) * 0.5).to(BF)
w1p = (torch.randn(2 * F, D, device=dev) * D**-0.5).to(BF); w2t = (torch.randn(F, D, device=dev) * F**-0.5).to(BF)
du = torch.empty(M, 2 * F, device=dev, dtype=BF); h = torch.empty(M, F, device=dev, dtype=BF)
print(f"{os.environ.get('DIFFULAB_HIP_LIB', 'product').split('_hip_')[-1]:24s} rc {timeit(lambda: ops.mlp_dswiglu_recompute(x, w1p, dt, w2t, du)):6.1f} us"
      f"   swiglu fwd (h only) {timeit(lambda: ops.gemm_nt_swiglu(x, w1p, None, h)):6.1f} us")
