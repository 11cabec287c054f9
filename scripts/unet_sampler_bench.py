"""DDPM sampling throughput of the MNIST UNet (configs/train_mnist_ddpm.yaml): images/s of a respaced `--steps`-step DDPM chain."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffulab_amd import Diffuser  # noqa: E402
from diffulab_amd.networks.denoisers import UNetModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=50)
a = ap.parse_args()
dev = "cuda"
torch.manual_seed(0)
m = UNetModel(image_size=[32, 32], in_channels=1, model_channels=128, out_channels=1, num_res_blocks=2,
              attention_resolutions=[4, 8, 16], num_heads=2, resblock_updown=True, n_classes=10, use_scale_shift_norm=True,
              classifier_free=False).to(dev).eval()
d = Diffuser(m, sampling_method="ddpm", model_type="gaussian_diffusion", n_steps=1000)
d.set_steps(a.steps)
y = torch.randint(0, 10, (a.batch,), device=dev)
run = lambda: d.generate({"y": y}, data_shape=(a.batch, 1, 32, 32), use_tqdm=False)["x"]  # noqa: E731
run()
torch.cuda.synchronize()
t0 = time.perf_counter()
out = run()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"workload": f"UNet MNIST {a.steps}-step DDPM sampling", "batch": a.batch, "s_per_batch": round(dt, 4),
                  "images_per_s": round(a.batch / dt, 1), "ms_per_forward": round(dt / a.steps * 1e3, 3),
                  "graphs": {str(k[1]): (v is not False) for k, v in (m.__dict__.get("_graphs") or {}).items()},
                  "finite": bool(torch.isfinite(out).all())}))
