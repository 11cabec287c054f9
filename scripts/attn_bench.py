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

if N <= 256:  # the training step's pair: QK-norm + RoPE on load (forward), token-major dQ / dK / dV (backward)
    from diffulab_amd.engine import rope_grid_tables
    D = H * dh
    ssq = (torch.rand(B * N, 2, device=dev) + 0.5) * D
    sq, sk = torch.ones(D, device=dev), torch.ones(D, device=dev)
    side = int(N ** 0.5)
    cos, sin = (z.to(dev) for z in rope_grid_tables(side, side, [32, 32], 10_000.0))
    rr = torch.empty(B * N, 2, device=dev)
    for mode in (0, 1, 0, 1):  # LAB switch: 0 = chain form (one workgroup per head), 1 = persistent + pipelined where it applies
        ops.lib().cdll.dl_lab_set_attn_pipe(mode)
        us = timeit(lambda: ops.attn_fwd_qkn(qkv, ssq, sq, sk, cos, sin, q, k, rr, out, lse, B, H, N, dh, 64, scale))
        us_i = timeit(lambda: ops.attn_fwd_qkn(qkv, ssq, sq, sk, cos, sin, None, None, None, out, lse, B, H, N, dh, 64, scale))
        print(f"attn_fwd_qkn N={N} pipe={mode}: training {us:7.1f} us, inference (no q / k / rrms outputs) {us_i:7.1f} us")
    ops.lib().cdll.dl_lab_set_attn_pipe(1)
