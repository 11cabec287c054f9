"""attention backward + QK-norm/RoPE backward at the headline shape (B=256, 6 heads, 256 tokens): the head-major pair (dl_attn_bwd_sv
+ dl_qk_norm_rope_bwd) against the token-major in-place pair (dl_attn_bwd_tok + dl_qk_norm_rope_bwd_inplace).
    python scripts/qk_inplace_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffulab_amd import ops  # noqa: E402
from diffulab_amd.engine import rope_grid_tables  # noqa: E402

dev, BF = "cuda", torch.bfloat16
B, H, N, dh = 256, 6, 256, 64
D, M = H * dh, B * N


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


qkv = (torch.randn(M, 3 * D, device=dev) * 0.5).to(BF)
sq, sk = torch.ones(D, device=dev), torch.ones(D, device=dev)
cos, sin = (t.to(dev) for t in rope_grid_tables(16, 16, [32, 32], 10_000.0))
q, k = (torch.empty(B, H, N, dh, device=dev, dtype=BF) for _ in range(2))
rrms = torch.empty(M, 2, device=dev)
ops.qk_norm_rope_fwd(qkv, sq, sk, cos, sin, q, k, None, rrms, B, N, H, dh, 64)
out, lse = torch.empty(B, N, D, device=dev, dtype=BF), torch.empty(B, H, N, device=dev)
ops.attn_fwd_qkv(q, k, qkv, out, lse, B, H, N, dh, 0.125)
do = (torch.randn(B, N, D, device=dev) * 0.5).to(BF)
dq, dk = (torch.empty(B, H, N, dh, device=dev, dtype=BF) for _ in range(2))
dqkv, ds = torch.empty(M, 3 * D, device=dev, dtype=BF), torch.zeros(2, D, device=dev)
part = torch.empty(1024 * 2 * D, device=dev)
a0 = timeit(lambda: ops.attn_bwd_qkv(q, k, qkv, out, do, lse, dq, dk, dqkv, B, H, N, dh, 0.125))
n0 = timeit(lambda: ops.qk_norm_rope_bwd(dq, dk, None, qkv, sq, sk, cos, sin, rrms, dqkv, ds, B, N, H, dh, 64))
a1 = timeit(lambda: ops.attn_bwd_tok(q, k, qkv, out, do, lse, dqkv, B, H, N, dh, 0.125))
n1 = timeit(lambda: ops.qk_norm_rope_bwd_inplace(qkv, sq, sk, cos, sin, rrms, dqkv, ds, part, B, N, H, dh, 64))
gb = 6.0 * M * D * 2 / 1e3  # q, k rows of qkv + dq, dk in + dq, dk out (bytes / 1e3 -> us * GB/s)
print(f"head-major : attn_bwd {a0:6.1f} us + qk_norm_rope_bwd {n0:6.1f} us ({gb / n0 / 1e3:.2f} TB/s) = {a0 + n0:6.1f} us")
print(f"token-major: attn_bwd {a1:6.1f} us + in-place (incl. fold) {n1:6.1f} us ({gb / n1 / 1e3:.2f} TB/s) = {a1 + n1:6.1f} us")
