"""LAB: per-block activation error of the bf16 UNet engine against the fp32 engine of the same build at configs/model/unet.yaml dims."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import synth  # noqa: E402
from oracle import unet as ounet  # noqa: E402

from diffulab_amd import unet_engine  # noqa: E402
from diffulab_amd.networks.denoisers import UNetModel  # noqa: E402

DEV = "cuda"
TRACE: list = []


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def wrap(name):
    orig = getattr(unet_engine.UNetEngine, name)

    def f(self, b, x, *a, **k):
        r = orig(self, b, x, *a, **k)
        out = r[0] if isinstance(r, tuple) else r
        TRACE.append((f"{name[1:-4]} {b.prefix} {b.cin}->{b.cout}" + (" up" if b.up else " down" if b.down else ""), out.float().clone()))
        return r

    setattr(unet_engine.UNetEngine, name, f)


for n in ("_res_fwd", "_attn_fwd", "_resample_fwd"):
    wrap(n)
orig_gn = {}

cfg = ounet.UNetConfig()
P = synth.generic_params(ounet.param_shapes(cfg), seed=41)
B = 2
x = synth.normal("fd.x0", (B, 1, 32, 32)).to(DEV)
y = synth.integers("fd.y", (B,), 10).to(DEV)
t = torch.tensor([17, 940], dtype=torch.int32).to(DEV)
tr = {}
for prec in ("fp32", "bf16"):
    m = UNetModel(image_size=[32, 32], in_channels=1, model_channels=128, out_channels=1, num_res_blocks=2, attention_resolutions=[4, 8, 16],
                  num_heads=2, resblock_updown=True, n_classes=10, use_scale_shift_norm=True, classifier_free=False)
    m.load_state_dict(P)
    m = m.set_precision(prec).to(DEV).eval()
    TRACE.clear()
    with torch.no_grad():
        out = m(x=x, timesteps=t, y=y, p=0.0)["x"].float()
    tr[prec] = list(TRACE) + [("prediction", out)]
    del m
for (n, a), (_, b) in zip(tr["bf16"], tr["fp32"]):
    print(f"{n:55s} rel {rel(a, b):.3e}   |ref| rms {b.pow(2).mean().sqrt().item():.3e}", flush=True)
