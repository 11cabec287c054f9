"""LAB: dl_gn_bwd on the GroupNorm shapes of the MNIST UNet's training step (B = 128), microseconds per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffulab_amd import ops  # noqa: E402

DEV, BF = "cuda", torch.bfloat16
B = int(os.environ.get("B", "128"))
if os.environ.get("DL_LAB_GN_FUSED"):
    ops.lib().cdll.dl_lab_set_gn_fused(int(os.environ["DL_LAB_GN_FUSED"]))
print("library:", os.environ.get("DIFFULAB_HIP_LIB", "(product)"))
for HW, C, film in ((1024, 128, True), (1024, 256, False), (256, 256, True), (256, 384, False), (256, 512, False), (64, 512, True),
                    (64, 1024, False), (64, 768, False), (64, 1536, False), (16, 1024, True), (16, 2048, False), (16, 1536, False)):
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(B * HW, C, device=DEV, generator=g).to(BF)
    dy = torch.randn(B * HW, C, device=DEV, generator=g).to(BF)
    dres = torch.randn(B * HW, C, device=DEV, generator=g).to(BF)
    w, b = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
    fs = (torch.randn(B, 2 * C, device=DEV) * 0.3).to(BF)
    st = torch.empty(B, 32, 2, device=DEV)
    ops.gn_stats(x, st, B, HW, C)
    dx = torch.empty_like(x)
    dw, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dfs = torch.zeros(B, 2 * C, device=DEV, dtype=BF)
    scr = torch.empty(8 * B * 4 * C + B * 64, device=DEV)
    sc, sh = (fs[:, :C], fs[:, C:]) if film else (None, None)
    run = lambda: ops.gn_bwd(dy, x, st, w, b, sc, sh, True, dres, dx, dw, db, dfs[:, :C] if film else None, dfs[:, C:] if film else None, scr, B, HW, C)  # noqa: E731
    out = torch.empty_like(x)
    if os.environ.get("FWD"):
        if os.environ["FWD"] == "2":
            run = lambda: (ops.gn_stats(x, st, B, HW, C), ops.gn_apply_fwd(x, st, w, b, sc, sh, True, out, B, HW, C))  # noqa: E731
        else:
            run = lambda: ops.gn_fwd(x, st, w, b, sc, sh, True, out, B, HW, C)  # noqa: E731
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    mb = B * HW * C * 2 * (2 if os.environ.get("FWD") else 4) / 1e6  # fwd: x in, out; bwd: x, dout, dres in, dx out (once each)
    print(f"HW={HW:5d} C={C:5d} film={int(film)}: {us:7.1f} us   {mb:7.1f} MB algorithmic -> {mb / us * 1e-6:6.2f} TB/s")
