import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _probe_lib import lib  # libdiffulab_probe.so (lab code, include/diffulab_probe.h)
L = lib()
src = (torch.randn(256 * 57344 // 2, device="cuda") * 0.1).to(torch.bfloat16)
out = torch.zeros(256 * 512, device="cuda")
iters = 2000
s = torch.cuda.current_stream().cuda_stream
for mode, name in ((5, "LDS-DMA only"), (6, "global_load + ds_write_b128"), (7, "global_load only")):
    for _ in range(2):
        L.call("dl_probe_mfma", mode, iters, src.data_ptr(), out.data_ptr(), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        L.call("dl_probe_mfma", mode, iters, src.data_ptr(), out.data_ptr(), s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    byts = 256.0 * iters * 57344
    print(f"mode {mode} {name:30s}: {ms*1e3:8.1f} us  {byts/ms/1e9:7.2f} TB/s aggregate  = {byts/256/(ms*1e-3*2.2e9):5.1f} B/clk/CU @2.2GHz")
