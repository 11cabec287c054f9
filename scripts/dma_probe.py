"""L2/HBM -> LDS operand stream of the NT GEMM tile walk alone: 128-byte row segments with one stage in flight (today's kernels) vs
64-byte segments with three stages in flight (what a 32-deep four-slot ring would issue).  python scripts/dma_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _probe_lib import lib  # libdiffulab_probe.so (lab code, include/diffulab_probe.h)

L = lib()
dev = "cuda"
M = 65536
out = torch.zeros(256 * 512, device=dev)
for K in (384, 1152, 1536, 3072):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(384, K, device=dev).to(torch.bfloat16)
    for kb in (128, 64):
        def go():
            L.call("dl_probe_dma", kb, a.data_ptr(), w.data_ptr(), M, K, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        for _ in range(3): go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): go()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        nbytes = (M // 256) * 640 * K * 2
        print(f"K={K:5d} segment {kb:3d} B: {us:7.1f} us  {nbytes/us/1e6:6.2f} TB/s of DMA ({M*K*2/1e6:.0f} MB of activations from HBM)")
