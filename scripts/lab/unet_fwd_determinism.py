"""LAB: is the bf16 UNet inference forward bit-reproducible?  Two runs of the same forward, per-block activation differences."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import diffulab_amd as da  # noqa: E402
from diffulab_amd import unet_engine  # noqa: E402

DEV = "cuda"
TRACE = []


def wrap(name):
    orig = getattr(unet_engine.UNetEngine, name)

    def f(self, b, x, *a, **k):
        r = orig(self, b, x, *a, **k)
        out = r[0] if isinstance(r, tuple) else r
        TRACE.append((f"{name[1:-4]} {b.prefix} {b.cin}->{b.cout}", out.float().clone()))
        return r

    setattr(unet_engine.UNetEngine, name, f)


for n in ("_res_fwd", "_attn_fwd", "_resample_fwd"):
    wrap(n)
os.environ["DL_HIPGRAPH"] = "0"
torch.manual_seed(0)
m = da.UNetModel(image_size=[32, 32], in_channels=3, model_channels=64, out_channels=3, num_res_blocks=1, attention_resolutions=[4],
                 channel_mult="1, 2, 2", num_heads=4, use_scale_shift_norm=True, resblock_updown=True, n_classes=10, classifier_free=True)
with torch.no_grad():
    for q in m.parameters():
        if float(q.abs().sum()) == 0:
            q.normal_(0, 0.05)
m = m.to(DEV).eval()
x = torch.randn(4, 3, 32, 32, device=DEV)
t = torch.tensor([10.0, 500.0, 900.0, 3.0], device=DEV)
y = torch.arange(4, device=DEV)
runs = []
for _ in range(3):
    TRACE.clear()
    with torch.no_grad():
        out = m(x=x, timesteps=t, y=y, p=0.0)["x"].float()
    runs.append(list(TRACE) + [("prediction", out.clone())])
for (n, a), (_, b), (_, c) in zip(*runs):
    d1 = (a - b).abs().max().item()
    d2 = (a - c).abs().max().item()
    print(f"{n:50s} max|run1-run2| {d1:.3e}  max|run1-run3| {d2:.3e}  scale {a.abs().max().item():.2e}")
