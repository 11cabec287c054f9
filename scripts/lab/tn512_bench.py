"""LAB: the weight-gradient shapes of a 512-wide block (CIFAR DiT / SPRINT / DDT configs) on the kernels dl_gemm_tn dispatches to"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffulab_amd import ops
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for R in (65536, 16384, 8192):
    tot, fl = 0.0, 0.0
    probs = []
    for name, Mo, No in (("qkv", 1536, 512), ("proj", 512, 512), ("up", 4096, 512), ("down", 512, 2048)):
        a = (torch.randn(R, Mo, device=dev) * 0.5).to(torch.bfloat16); b = (torch.randn(R, No, device=dev) * 0.5).to(torch.bfloat16)
        c = torch.zeros(Mo, No, device=dev); probs.append((a, b, c))
        us = timeit(lambda: ops.gemm_tn(a, b, c)); tot += us; fl += 2.0 * R * Mo * No
        print(f"R={R:6d} {name:5s} [{Mo:4d} x {No:4d}]: {us:7.1f} us {2.0*R*Mo*No/us/1e6:7.1f} TF/s")
    print(f"R={R:6d} block total {tot:7.1f} us = {fl/tot/1e6:6.1f} TF/s")

    slab = torch.empty(ops.WgradGroups.slab_floats(512, 2048), device=dev)
    us = timeit(lambda: ops.gemm_tn_group(probs, slab))
    print(f"R={R:6d} grouped 256x256 tiles (one launch + fold) {us:7.1f} us = {fl/us/1e6:6.1f} TF/s")
