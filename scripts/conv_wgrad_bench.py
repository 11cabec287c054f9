"""3x3 convolution weight gradient alone at the MNIST-DDPM UNet's shapes (B = 128 by default), beside the forward convolution of
the same shape (same FLOPs):  python scripts/conv_wgrad_bench.py [batch]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffulab_amd import ops

dev, BF = "cuda", torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lib = ops.lib().cdll
if os.environ.get("DL_LAB_WGRAD_HALO"):
    lib.dl_lab_set_wgrad_halo(int(os.environ["DL_LAB_WGRAD_HALO"]))
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
r64 = lambda v: (v + 63) // 64 * 64
zero = torch.zeros(64, device=dev, dtype=BF)
tot_w = tot_f = tot_fl = tot_p = tot_fp = tot_f1 = 0.0
for H, ci, co in ((32, 128, 128), (32, 256, 128), (16, 128, 256), (16, 256, 256), (16, 512, 256), (16, 384, 256), (8, 256, 512), (8, 512, 512),
                  (8, 1024, 512), (8, 768, 512), (4, 512, 1024), (4, 1024, 1024), (4, 2048, 1024), (4, 1536, 1024)):
    M = B * H * H
    x = torch.randn(M, ci, device=dev).to(BF)
    dy = torch.randn(M, co, device=dev).to(BF)
    g = torch.zeros(r64(9 * ci), co, device=dev)
    wf = (torch.randn(co, r64(9 * ci), device=dev) * 0.02).to(BF)
    out = torch.empty(M, co, device=dev, dtype=BF)
    scr = torch.empty(8 * M * co, device=dev)
    ok = ops.conv3x3_wgrad_tn(x, B, H, H, ci, dy, co, g, zero)
    tw = timeit(lambda: ops.conv3x3_wgrad_tn(x, B, H, H, ci, dy, co, g, zero)) if ok else float("nan")
    tf = timeit(lambda: ops.conv3x3_nt(x, B, H, H, ci, wf, out, co, None, None, zero, scr))
    # the partial-image form + what its fold costs beyond the fold of ONE image (the atomic form's)
    n = ops.conv3x3_wgrad_nparts(H, H, ci, co, M)
    parts = torch.empty(n, r64(9 * ci), co, device=dev)
    dw = torch.zeros(co, ci, 3, 3, device=dev)
    tp = timeit(lambda: ops.conv3x3_wgrad_tn_parts(x, B, H, H, ci, dy, co, parts, zero))
    tabp, tab1 = ops.ConvFoldTable([(parts, dw)]), ops.ConvFoldTable([(g, dw)])
    tfp, tf1 = timeit(lambda: tabp.run(clear=False)), timeit(lambda: tab1.run(clear=True))
    fl = 2.0 * M * co * 9 * ci
    tot_w += tw; tot_f += tf; tot_fl += fl; tot_p += tp; tot_fp += tfp; tot_f1 += tf1
    print(f"{H:2d}x{H:<2d} Ci={ci:4d} Co={co:4d}  R={M:6d}: wgrad {tw:7.1f} us {fl / tw / 1e6:6.1f} TF/s  fold {tf1:6.1f} us | {n:3d} parts {tp:7.1f} us {fl / tp / 1e6:6.1f} TF/s  fold {tfp:6.1f} us"
          f" | forward {tf:7.1f} us {fl / tf / 1e6:6.1f} TF/s")
print(f"sum: wgrad {tot_w:.0f} + fold {tot_f1:.0f} us ({tot_fl / tot_w / 1e6:.0f} TF/s)   parts {tot_p:.0f} + fold {tot_fp:.0f} us ({tot_fl / tot_p / 1e6:.0f} TF/s)   forward {tot_f:.0f} us ({tot_fl / tot_f / 1e6:.0f} TF/s)")
