"""GroupNorm32 forward / backward alone at the MNIST-DDPM UNet's shapes (B=128): python scripts/gn_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffulab_amd import ops

dev, BF = "cuda", torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for HW, C in ((1024, 128), (1024, 256), (1024, 384), (256, 128), (256, 256), (256, 512), (256, 768), (64, 256), (64, 512), (64, 1024), (16, 512), (16, 1024), (16, 2048)):
    x = torch.randn(B * HW, C, device=dev).to(BF)
    dy = torch.randn(B * HW, C, device=dev).to(BF)
    dres = torch.randn(B * HW, C, device=dev).to(BF)
    fs = torch.randn(B, 2 * C, device=dev).to(BF) * 0.1
    w, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    st = torch.empty(B, 32, 2, device=dev)
    out, dx = torch.empty_like(x), torch.empty_like(x)
    dw, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dfs = torch.zeros(B, 2 * C, device=dev, dtype=BF)
    scr = torch.empty(8 * B * 4 * C + B * 64, device=dev)
    t_s = timeit(lambda: ops.gn_stats(x, st, B, HW, C))
    t_f = timeit(lambda: ops.gn_apply_fwd(x, st, w, b, fs[:, :C], fs[:, C:], True, out, B, HW, C))
    t_ff = timeit(lambda: ops.gn_fwd(x, st, w, b, fs[:, :C], fs[:, C:], True, out, B, HW, C))
    t_b = timeit(lambda: ops.gn_bwd(dy, x, st, w, b, fs[:, :C], fs[:, C:], True, dres, dx, dw, db, dfs[:, :C], dfs[:, C:], scr, B, HW, C))
    mb = B * HW * C * 2 / 1e6
    print(f"HW={HW:5d} C={C:5d} ({mb:6.1f} MB/tensor): stats {t_s:6.1f} us ({mb/t_s:5.2f} TB/s)  apply_fwd {t_f:6.1f} us ({2*mb/t_f:5.2f} TB/s)  "
          f"fwd fused {t_ff:6.1f} us ({2*mb/t_ff:5.2f} TB/s)  bwd {t_b:6.1f} us ({4*mb/t_b:5.2f} TB/s on x, dy, dres in + dx out)")
