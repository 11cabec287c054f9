"""3x3 implicit-GEMM convolution forward alone at the MNIST-DDPM UNet's high-resolution shapes (B=128):
    DL_CONV_BIG=0|1 python scripts/conv_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffulab_amd import ops

dev, BF = "cuda", torch.bfloat16
B = 128
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
lib = ops.lib().cdll
if os.environ.get("DL_LAB_CONV_SPLIT128"):
    lib.dl_lab_set_conv_split128(int(os.environ["DL_LAB_CONV_SPLIT128"]))
if os.environ.get("CONV_BENCH_B"):
    B = int(os.environ["CONV_BENCH_B"])
if os.environ.get("HALO_NST"):
    lib.dl_lab_set_conv_halo_nst(int(os.environ["HALO_NST"]))
rows = []
# every 3x3 convolution shape of the MNIST-DDPM UNet (model_channels 128, channel_mult 1-2-4-8, 32x32 .. 4x4) incl. the decoder's
# concatenated inputs; old = the 128x128 kernel (+ split-K), big = conv3x3_big_k (persistent 256-row tiles)
EXTRA = ((32, 128, 256), (16, 256, 512), (16, 128, 512), (32, 128, 384)) if os.environ.get("CONV_BENCH_DGRAD") else ()  # data-gradient shapes: 256-wide tiles
for H, ci, co in EXTRA + ((32, 128, 128), (32, 256, 128), (16, 128, 256), (16, 256, 256), (16, 512, 256), (16, 384, 256), (8, 256, 512), (8, 512, 512),
                  (8, 1024, 512), (8, 768, 512), (4, 512, 1024), (4, 1024, 1024), (4, 2048, 1024), (4, 1536, 1024)):
    M = B * H * H
    x = torch.randn(M, ci, device=dev).to(BF)
    wf = (torch.randn(co, r64(9 * ci), device=dev) * 0.02).to(BF)
    bias = torch.randn(co, device=dev)
    res = torch.randn(M, co, device=dev).to(BF)
    scr = torch.empty(8 * M * co, device=dev)
    outs, us = {}, {}
    for mode, halo, name in ((0, 0, "old"), (1, 0, "big"), (1, 1, "halo")):
        lib.dl_lab_set_conv_big(mode)
        lib.dl_lab_set_conv_halo(halo)
        out = torch.empty(M, co, device=dev, dtype=BF)
        us[name] = timeit(lambda: ops.conv3x3_nt(x, B, H, H, ci, wf, out, co, bias, res, zero, scr))
        outs[name] = out.float()
    err = float((outs["big"] - outs["old"]).norm() / outs["old"].norm())
    fl = 2.0 * M * co * 9 * ci
    errh = float((outs["halo"] - outs["big"]).norm() / outs["big"].norm())
    print(f"{H:2d}x{H:<2d} Ci={ci:4d} Co={co:4d}: old {us['old']:7.1f} us {fl / us['old'] / 1e6:7.1f} TF/s   big {us['big']:7.1f} us {fl / us['big'] / 1e6:7.1f} TF/s   "
          f"halo {us['halo']:7.1f} us {fl / us['halo'] / 1e6:7.1f} TF/s   rel diff big/old {err:.1e} halo/big {errh:.1e}")
lib.dl_lab_set_conv_big(1)
lib.dl_lab_set_conv_halo(1)
