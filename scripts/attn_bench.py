"""micro-benchmark of the attention kernels at the DiT-S/2 (B=256) shape: 1536 heads of 256 tokens x 64.
    python scripts/attn_bench.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffulab_amd import ops

dev, bf = "cuda", torch.bfloat16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B, H, dh = 256 * 256 // N, 6, 64
q, k, v, do = (torch.randn(B, H, N, dh, device=dev).to(bf) for _ in range(4))
out = torch.empty(B, N, H * dh, device=dev, dtype=bf)
do = do.view(B, N, H * dh)
lse = torch.empty(B, H, N, device=dev)
dq, dk, dv = (torch.empty_like(q) for _ in range(3))
scale = dh ** -0.5


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


fl = 4 * B * H * N * N * dh
us = timeit(lambda: ops.attn_fwd(q, k, v, out, lse, B, H, N, dh, scale))
print(f"attn_fwd N={N}: {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s")
us = timeit(lambda: ops.attn_bwd(q, k, v, out, do, lse, dq, dk, dv, B, H, N, dh, scale))
print(f"attn_bwd N={N}: {us:7.1f} us  {3.5 * fl / us / 1e6:6.1f} TFLOP/s (7 matmuls incl. the recomputed S, twice)")
if N <= 256:  # V / dV in place inside token-major qkv / dqkv rows
    D = H * dh
    qkv = torch.randn(B * N, 3 * D, device=dev).to(bf)
    dqkv = torch.empty_like(qkv)
    us = timeit(lambda: ops.attn_fwd_qkv(q, k, qkv, out, lse, B, H, N, dh, scale))
    print(f"attn_fwd_qkv N={N}: {us:7.1f} us")
    us = timeit(lambda: ops.attn_bwd_qkv(q, k, qkv, out, do, lse, dq, dk, dqkv, B, H, N, dh, scale))
    print(f"attn_bwd_qkv N={N}: {us:7.1f} us")

if N >= 512:  # long joint sequences: the fp8 MFMA forward (quantisation pre-pass timed separately) against the bf16 kernel
    q8, k8 = torch.empty(B, H, N, dh, device=dev, dtype=torch.uint8), torch.empty(B, H, N, dh, device=dev, dtype=torch.uint8)
    v8t, sc = torch.empty(B, H, dh, N, device=dev, dtype=torch.uint8), torch.empty(B, H, 3, device=dev)
    us_q = timeit(lambda: ops.attn_fp8_quantize(q, k, v, q8, k8, v8t, sc, B, H, N, N))
    us_f = timeit(lambda: ops.attn_fwd_fp8(q8, k8, v8t, sc, out, lse, B, H, N, N, dh, scale))
    us_b = timeit(lambda: ops.attn_fwd_ex(q, k, v, out, lse, B, H, N, N, dh, scale))
    print(f"attn_fwd_ex  bf16 N={N}: {us_b:7.1f} us  {fl / us_b / 1e6:6.1f} TFLOP/s")
    print(f"attn_fwd_fp8      N={N}: {us_f:7.1f} us  {fl / us_f / 1e6:6.1f} TFLOP/s ({fl / us_f / 1e6 / 5000:.3f} of the 5 PF fp8 peak)"
          f"  + quantise {us_q:6.1f} us  -> {fl / (us_f + us_q) / 1e6:6.1f} TFLOP/s end to end")
