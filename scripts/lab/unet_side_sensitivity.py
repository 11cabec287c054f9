"""LAB: how much of the UNet step is the side stream's -- step time with parts of the side-stream work stubbed out (WRONG gradients;
timing only)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffulab_amd import Diffuser, ops  # noqa: E402
from diffulab_amd.networks.denoisers import UNetModel  # noqa: E402
from diffulab_amd.training.optim import FusedAdamW  # noqa: E402

mode = os.environ.get("STUB", "")
if "colsum" in mode:
    ops.colsum = lambda *a, **k: None
if "fold" in mode:
    ops.conv3x3_wgrad_fold = lambda *a, **k: None
if "wgrad" in mode:
    ops.conv3x3_wgrad_tn = lambda *a, **k: True
if "gn" in mode:
    ops.gn_bwd = lambda *a, **k: None
dev = "cuda"
torch.manual_seed(0)
m = UNetModel(image_size=[32, 32], in_channels=1, model_channels=128, out_channels=1, num_res_blocks=2, attention_resolutions=[4, 8, 16],
              num_heads=2, resblock_updown=True, n_classes=10, use_scale_shift_norm=True, classifier_free=False).to(dev)
gd = Diffuser(m, sampling_method="ddpm", model_type="gaussian_diffusion", n_steps=1000)
opt = FusedAdamW(m.parameters(), lr=1e-4)
B = 128
x0 = torch.randn(B, 1, 32, 32, device=dev)
y = torch.randint(0, 10, (B,), device=dev)


def step():
    opt.zero_grad()
    loss = gd.compute_loss({"x": x0, "y": y, "p": 0.0}, timesteps=gd.draw_timesteps(B))["loss"]
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print(f"STUB={mode or '(none)':24s} {1e3 * (time.perf_counter() - t0) / 20:.2f} ms/step")
