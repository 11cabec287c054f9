"""Write bandwidth of the GEMM register epilogue's store pattern (32 rows x 32 B per wave instruction) against full 128-byte lines,
on a [65536, 1152] bf16 output (dl_probe_mfma modes 8 / 9):  python scripts/store_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _probe_lib import lib  # libdiffulab_probe.so (lab code, include/diffulab_probe.h)
L = lib()

out = torch.empty(65536, 1152, device="cuda", dtype=torch.bfloat16)
s = torch.cuda.current_stream().cuda_stream
iters = 20
for mode, name in [(8, "32 rows x 32 B per instruction"), (9, "8 rows x 128 B per instruction")]:
    L.call("dl_probe_mfma", mode, 2, 0, out.data_ptr(), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.call("dl_probe_mfma", mode, iters, 0, out.data_ptr(), s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name}: {us:7.1f} us per 151 MB  {out.numel() * 2 / us / 1e6:5.2f} TB/s")
t = torch.empty_like(out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    t.fill_(1.0)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / iters
print(f"torch fill_: {us:7.1f} us  {out.numel() * 2 / us / 1e6:5.2f} TB/s")
