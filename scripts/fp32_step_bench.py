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
if len(sys.argv) > 3 and sys.argv[3] == "unet":  # configuration 1 (MNIST-DDPM UNet, 276.7 M parameters) in the fp32 regime
    from diffulab_amd.config import instantiate, load_config

    cfg = load_config(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs"), "train_mnist_ddpm")
    m = instantiate(cfg.model).set_precision("fp32").to("cuda")
    d = Diffuser(m, sampling_method="ddpm", model_type="gaussian_diffusion", n_steps=1000)
    opt = FusedAdamW(m.parameters(), lr=1e-4, weight_decay=0.01)
    x, y = torch.randn(B, 1, 32, 32, device="cuda"), torch.randint(0, 10, (B,), device="cuda")

    def ustep():
        opt.zero_grad()
        d.compute_loss({"x": x.clone(), "y": y, "p": 0.0}, timesteps=d.draw_timesteps(B).to("cuda"))["loss"].backward()
        opt.step()

    for _ in range(2):
        ustep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ustep()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"fp32 regime, MNIST-DDPM UNet B={B}: {dt * 1e3:.1f} ms/step = {B / dt:.0f} img/s ({B / dt * 3 * 27.1e9 / 1e12:.1f} TFLOP/s of conv work)")
    sys.exit(0)
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
