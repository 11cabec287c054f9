"""A/B of the weight-gradient kernels at the headline shapes (R = 65536 tokens): DL_GEMM_TN_VARIANT=1 (384x128 persistent) vs 2
(384x192 four-slot ring, gemm_w4.hip), each in its own process because the variant is latched at first use.
    python scripts/tn_w4_bench.py            # runs both variants (and both wave counts of variant 2) as child processes
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from diffulab_amd import ops
dev = "cuda"
R = 65536
def rnd(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, Mo, No in [("w_qkv", 1152, 384), ("w_proj", 384, 384), ("w_mlp1", 3072, 384), ("w_mlp2", 384, 1536)]:
    a, b = rnd(R, Mo), rnd(R, No)
    c = torch.zeros(Mo, No, device=dev)
    ops.gemm_tn(a[:8192], b[:8192], c)
    ref = a[:8192].float().T @ b[:8192].float()
    err = ((c - ref).norm() / ref.norm()).item()
    c.zero_()
    us = timeit(lambda: ops.gemm_tn(a, b, c))
    us192 = timeit(lambda: ops.gemm_tn(a, b, c, max_wgs=192))
    print(f"  {name:7s} [{Mo:4d} x {No:4d}]: {us:7.1f} us {2.0*R*Mo*No/us/1e6:7.1f} TF/s | capped at 192 WGs {us192:7.1f} us | rel err (8192 rows) {err:.2e}")
''' % ROOT
ENVS = [{"DL_GEMM_TN_VARIANT": "1"}, {"DL_GEMM_TN_VARIANT": "2", "DL_GEMM_TN_W4_XCD": "0"}, {"DL_GEMM_TN_VARIANT": "2"},
        {"DL_GEMM_TN_VARIANT": "2", "DL_GEMM_TN_W4_MINSTEPS": "64"}, {"DL_GEMM_TN_VARIANT": "2", "DL_GEMM_TN_W4_MINSTEPS": "16"}]
if len(sys.argv) > 1 and sys.argv[1] == "probe":
    ENVS = [{"DL_GEMM_TN_VARIANT": "2", "DL_GEMM_TN_W4_PROBE": str(p)} for p in (0, 2, 4, 8, 10, 12, 6)]
for env in ENVS:
    print(env, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env={**os.environ, **env}, check=False)
