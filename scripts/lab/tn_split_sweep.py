"""LAB: dl_gemm_tn at forced split counts for the UNet's linear weight-gradient shapes (dl_lab_set_tn_force_splits)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffulab_amd import ops
dev = "cuda"
lib = ops.lib().cdll
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
shapes = [(64, 128, 131072), (128, 256, 131072), (128, 384, 131072), (256, 128, 32768), (256, 384, 32768), (256, 512, 32768), (256, 768, 32768),
          (512, 256, 8192), (512, 512, 8192), (512, 768, 8192), (512, 1024, 8192), (512, 1536, 8192), (1024, 512, 8192), (1024, 512, 2048),
          (1024, 1024, 2048), (1024, 1536, 2048), (1024, 2048, 2048), (2048, 1024, 2048), (28672, 512, 128),
          (256, 256, 16384), (512, 512, 4096), (1024, 1024, 1024), (2048, 1024, 1024), (128, 256, 65536)]
if True:
    for M, N, R in shapes:
        a = torch.randn(R, M, device=dev).to(torch.bfloat16); b = torch.randn(R, N, device=dev).to(torch.bfloat16)
        out = torch.zeros(M, N, device=dev)
        lib.dl_lab_set_tn_force_splits(0)
        t_model = timeit(lambda: ops.gemm_tn(a, b, out, M=M, N=N))
        row = []
        for S in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 256):
            if S > R // 64: break
            lib.dl_lab_set_tn_force_splits(S)
            row.append((timeit(lambda: ops.gemm_tn(a, b, out, M=M, N=N)), S))
        lib.dl_lab_set_tn_force_splits(0)
        best = min(row)
        ntile = ((M + 127) // 128) * ((N + 127) // 128)
        print(f"M={M:5d} N={N:5d} R={R:6d} tiles {ntile:4d}: model {t_model:6.1f} us | best S={best[1]:3d} {best[0]:6.1f} us | " + " ".join(f"{S}:{t:.0f}" for t, S in row))
