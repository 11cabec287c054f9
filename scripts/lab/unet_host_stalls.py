"""LAB: which host call stalls while the GPU is running a UNet training step -- wall-clock time of every ops.* call, allocation and
stream / event call of the step, per function: total, mean, max (steady state, no synchronisation inside the loop)."""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffulab_amd import Diffuser, ops, unet_engine  # noqa: E402
from diffulab_amd.networks.denoisers import UNetModel  # noqa: E402
from diffulab_amd.training.optim import FusedAdamW  # noqa: E402

STAT = collections.defaultdict(lambda: [0, 0.0, 0.0])
ON = [False]


def wrap(owner, name, label=None):
    fn = getattr(owner, name)
    label = label or name

    def f(*a, **k):
        if not ON[0]:
            return fn(*a, **k)
        t0 = time.perf_counter()
        r = fn(*a, **k)
        dt = time.perf_counter() - t0
        s = STAT[label]
        s[0] += 1
        s[1] += dt
        s[2] = max(s[2], dt)
        return r

    setattr(owner, name, f)


for n in ("colsum", "conv3x3_wgrad_tn", "conv3x3_wgrad_fold", "conv3x3_nt", "gn_bwd", "gn_fwd", "gn_apply_fwd", "gn_stats", "gemm_nt", "gemm_tn",
          "copy2d_bf16", "add_bf16", "attn_small_fwd", "attn_small_bwd", "expand2x2", "reduce2x2"):
    wrap(ops, n, "ops." + n)
wrap(torch, "zeros", "torch.zeros")
wrap(torch, "empty", "torch.empty")
wrap(torch.cuda.Stream, "record_event", "Stream.record_event")
wrap(torch.cuda.Stream, "wait_event", "Stream.wait_event")
wrap(torch.Tensor, "record_stream", "Tensor.record_stream")
wrap(unet_engine.UNetEngine, "_off_chain", "[_off_chain total]")
wrap(unet_engine.UNetEngine, "_conv3_bwd", "[_conv3_bwd total]")
wrap(unet_engine.UNetEngine, "_res_bwd", "[_res_bwd total]")
wrap(unet_engine.UNetEngine, "_gn_bwd", "[_gn_bwd total]")
wrap(unet_engine.UNetEngine, "backward", "[backward total]")
wrap(unet_engine.UNetEngine, "forward", "[forward total]")

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


for _ in range(4):
    step()
torch.cuda.synchronize()
ON[0] = True
n = 10
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
ON[0] = False
print(f"issue {1e3 * (t1 - t0) / n:.2f} ms/step, wall {1e3 * (t2 - t0) / n:.2f} ms/step")
for k, (c, tot, mx) in sorted(STAT.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:28s} calls/step {c / n:7.1f}  total {1e3 * tot / n:7.3f} ms/step  mean {1e6 * tot / c:7.1f} us  max {1e6 * mx:8.1f} us")
