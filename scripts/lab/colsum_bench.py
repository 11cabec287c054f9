import os, sys, torch
sys.path.insert(0, os.getcwd())
from diffulab_amd import ops
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for R, C in ((65536, 13312), (131072, 256), (65536, 384), (8192, 512)):
    x = torch.randn(R, C, device="cuda").to(torch.bfloat16); o = torch.zeros(C, device="cuda")
    us = timeit(lambda: ops.colsum(x, o, R, C))
    print(f"colsum [{R} x {C}] bf16: {us:8.1f} us  {R*C*2/us/1e6:6.2f} TB/s")
