"""LAB: where the bf16 UNet's prediction error at configs/model/unet.yaml dims comes from -- bf16 regime against the fp32 regime of
the same build (itself within 3e-6 of the oracle) over structural variants of the model."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import synth  # noqa: E402
from oracle import unet as ounet  # noqa: E402

from diffulab_amd.networks.denoisers import UNetModel  # noqa: E402

DEV = "cuda"


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def run(tag, **over):
    kw = dict(image_size=(32, 32), in_channels=1, model_channels=128, out_channels=1, num_res_blocks=2, attention_resolutions=(4, 8, 16),
              channel_mult=(1, 2, 4, 8), num_heads=2, use_scale_shift_norm=True, resblock_updown=True, n_classes=10, classifier_free=False)
    kw.update(over)
    cfg = ounet.UNetConfig(**kw)
    P = synth.generic_params(ounet.param_shapes(cfg), seed=41)
    mk = dict(kw, image_size=list(kw["image_size"]), attention_resolutions=list(kw["attention_resolutions"]),
              channel_mult=", ".join(map(str, kw["channel_mult"])))
    B = int(os.environ.get("B", "2"))
    x = synth.normal("fd.x0", (B, 1, 32, 32)).to(DEV)
    y = synth.integers("fd.y", (B,), 10).to(DEV)
    t = torch.tensor([17, 940] * (B // 2), dtype=torch.int32).to(DEV)
    outs = {}
    for prec in ("fp32", "bf16"):
        m = UNetModel(**mk)
        m.load_state_dict(P)
        m = m.set_precision(prec).to(DEV).eval()
        with torch.no_grad():
            outs[prec] = m(x=x, timesteps=t, y=y, p=0.0)["x"].float()
        del m
    print(f"{tag:40s} rel {rel(outs['bf16'], outs['fp32']):.3e}   per sample", [round(rel(outs['bf16'][i], outs['fp32'][i]), 4) for i in range(B)], flush=True)


run("full")
run("no attention", attention_resolutions=())
run("attention at 16 only (1024 ch)", attention_resolutions=(16,))
run("attention at 4 only", attention_resolutions=(4,))
run("mult 1,2", channel_mult=(1, 2), attention_resolutions=())
run("mult 1,2,4", channel_mult=(1, 2, 4), attention_resolutions=())
run("1 res block", num_res_blocks=1)
run("1 res block no attention", num_res_blocks=1, attention_resolutions=())
run("no updown resblocks", resblock_updown=False)
run("additive conditioning", use_scale_shift_norm=False)
run("mc 64", model_channels=64)
run("mc 32", model_channels=32)
