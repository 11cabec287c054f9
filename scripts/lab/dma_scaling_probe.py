"""LAB (round 4): does a CU's L2 -> LDS operand rate scale with the number of resident waves / workgroups?  DMA-only tile walks
(dl_probe_dma2) in several geometries at the headline shape.   python scripts/lab/dma_scaling_probe.py"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from _probe_lib import lib

L = lib()
i, v, q = ctypes.c_int, ctypes.c_void_p, ctypes.c_int64
L.cdll.dl_probe_dma2.argtypes = [i, i, i, i, i, v, v, q, q, v, v]
dev = "cuda"
M = 65536
out = torch.zeros(256 * 512, device=dev)
geos = [  # (workgroups per CU, threads, activation rows, weight rows, ring slots)
    (1, 512, 256, 384, 2), (1, 1024, 256, 384, 2), (2, 512, 128, 384, 1), (2, 256, 128, 384, 1), (2, 512, 128, 192, 2),
    (4, 256, 64, 192, 1), (4, 256, 128, 128, 1), (2, 512, 256, 128, 1), (1, 512, 256, 384, 1), (1, 256, 256, 384, 2),
]
for K in (384, 3072):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(384, K, device=dev).to(torch.bfloat16)
    for wpc, th, ra, rb, ns in geos:
        def go():
            L.call("dl_probe_dma2", wpc, th, ra, rb, ns, a.data_ptr(), w.data_ptr(), M, K, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        for _ in range(3): go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): go()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        nbytes = (M // ra) * (ra + rb) * K * 2
        print(f"K={K:5d}  {wpc} WG/CU x {th // 64:2d} waves, tile {ra:3d}+{rb:3d} rows, {ns} slot(s): {us:7.1f} us  "
              f"{nbytes / us / 1e6:6.2f} TB/s staged ({nbytes / 1e9:.2f} GB), {wpc * th // 64:2d} waves/CU")
