"""LAB (round 4): the 256 + 384-row operand stream of the NT GEMM tile walk as PLAIN 16-byte loads into registers (dl_probe_ld) beside
its direct-to-LDS form (dl_probe_dma2): is the L2 -> CU path itself faster than `global_load_lds`?   python scripts/lab/ld_vs_dma_probe.py"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
from _probe_lib import lib

L = lib()
i, v, q = ctypes.c_int, ctypes.c_void_p, ctypes.c_int64
L.cdll.dl_probe_dma2.argtypes = [i, i, i, i, i, v, v, q, q, v, v]
L.cdll.dl_probe_ld.argtypes = [i, i, v, v, q, q, v, v]
dev = "cuda"
M = 65536
out = torch.zeros(256 * 512, device=dev)


def timed(go):
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 20


for K in (384, 1536, 3072):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(384, K, device=dev).to(torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    nbytes = (M // 256) * 640 * K * 2
    us = timed(lambda: L.call("dl_probe_dma2", 1, 512, 256, 384, 2, a.data_ptr(), w.data_ptr(), M, K, out.data_ptr(), st))
    print(f"K={K:5d}  direct-to-LDS, 2 slots        : {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s ({nbytes / 1e9:.2f} GB)")
    for pat, name in ((0, "8 rows x 128 B / instr "), (1, "32 rows x 32 B / instr ")):
        for depth in (2, 3):
            us = timed(lambda: L.call("dl_probe_ld", pat, depth, a.data_ptr(), w.data_ptr(), M, K, out.data_ptr(), st))
            print(f"K={K:5d}  plain loads {name} depth {depth}: {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s")

# ---- is a hot L2 channel (all workgroups of an XCD reading the same weight lines at the same moment) what bounds the stream?
L.cdll.dl_probe_dma2_rot.argtypes = [i, i, i, i, i, i, v, v, q, q, v, v]
for K in (384, 1536, 3072):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(384, K, device=dev).to(torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    for ra, rb in ((256, 384), (8, 384), (256, 8)):
        nbytes = 256 * (ra + rb) * K * 2
        for rot in (0, 1, 5, 7):
            us = timed(lambda: L.call("dl_probe_dma2_rot", 1, 512, ra, rb, 2, rot, a.data_ptr(), w.data_ptr(), 256 * ra, K, out.data_ptr(), st))
            print(f"K={K:5d}  DMA {ra:3d} activation + {rb:3d} weight rows, k rotated by {rot} per workgroup: {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s")
