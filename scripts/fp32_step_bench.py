"""throughput of the fp32-class regime (precision_type="no") at the headline shape, for the record: DiT-S/2 train step on the f32 MFMA.
    python scripts/fp32_step_bench.py [batch] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import S2, train_flops_per_image  # noqa: E402
from diffulab_amd import Diffuser, MMDiT  # noqa: E402
from diffulab_amd.training import FusedAdamW  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
m = MMDiT(**S2).set_precision("fp32").to("cuda")
d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=50, extra_args={"logits_normal": True})
opt = FusedAdamW(m.parameters(), lr=1e-4, weight_decay=0.01)
x, y = torch.randn(B, 4, 32, 32, device="cuda"), torch.randint(0, 1000, (B,), device="cuda")


def step():
    opt.zero_grad()
    t = d.draw_timesteps(B).to("cuda")
    d.compute_loss({"x": x, "y": y, "p": 0.1}, timesteps=t)["loss"].backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"fp32 regime B={B}: {dt * 1e3:.1f} ms/step = {B / dt:.0f} img/s = {B / dt * train_flops_per_image() / 1e12:.1f} TFLOP/s "
      f"({B / dt * train_flops_per_image() / 157.3e12:.2f} of the 157.3 TF f32 MFMA peak)")
