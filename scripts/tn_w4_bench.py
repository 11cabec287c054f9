"""The weight-gradient kernels of csrc/gemm_w4.hip alone on the chip at the headline shapes (R = 65536 tokens, DiT-S block):
the four one-problem launches (f32 atomics) against the grouped launch (dl_gemm_tn_group: one launch, partial slabs + fold),
with and without the side-stream workgroup cap.
    python scripts/tn_w4_bench.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffulab_amd import ops  # noqa: E402

dev = "cuda"
R = 65536
SHAPES = [("w_qkv", 1152, 384), ("w_proj", 384, 384), ("w_mlp1", 3072, 384), ("w_mlp2", 384, 1536)]


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


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


ops_ab = [(rnd(R, Mo), rnd(R, No), torch.zeros(Mo, No, device=dev)) for _, Mo, No in SHAPES]
flop = sum(2.0 * R * Mo * No for _, Mo, No in SHAPES)
for cap in (0, 192, 128, 64):
    tot = 0.0
    line = []
    for (name, Mo, No), (a, b, c) in zip(SHAPES, ops_ab):
        us = timeit(lambda: ops.gemm_tn(a, b, c, max_wgs=cap))
        tot += us
        line.append(f"{name} {us:6.1f}")
    print(f"cap {cap:3d}  four atomic launches: {tot:7.1f} us = {flop / tot / 1e6:6.1f} TF/s   ({', '.join(line)})")
total = sum(Mo * No for _, Mo, No in SHAPES)
for ranges in (8, 4, 2):
    slab = torch.empty(ranges * total, device=dev)
    for cap in (0, 192, 128, 64):
        us = timeit(lambda: ops.gemm_tn_group(ops_ab, slab, max_wgs=cap))
        print(f"cap {cap:3d}  grouped launch + fold, slab for {ranges} ranges: {us:7.1f} us = {flop / us / 1e6:6.1f} TF/s")
