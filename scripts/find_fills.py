"""Which Python lines launch fill kernels inside one DiT-S/2 training step (torch profiler, with_stack)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import torch
from torch.profiler import ProfilerActivity, profile
from diffulab_amd import Diffuser, MMDiT
from diffulab_amd.training import FusedAdamW

dev = "cuda"
kw = dict(input_channels=4, output_channels=4, inner_dim=384, embedding_dim=384, num_heads=6, mlp_ratio=4, patch_size=2, depth=12,
          n_classes=1000, classifier_free=True)
m = MMDiT(simple_dit=True, **kw).to(dev)
d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=50, extra_args={"logits_normal": True})
opt = FusedAdamW(m.parameters(), lr=1e-4, weight_decay=0.01)
B = 256
x0 = torch.randn(B, 4, 32, 32, device=dev)
y = torch.randint(0, 1000, (B,), device=dev)


def step():
    opt.zero_grad()
    t = d.draw_timesteps(B).to(dev, non_blocking=True)
    losses = d.compute_loss({"x": x0, "y": y, "p": 0.1}, timesteps=t)
    sum(losses.values()).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like", "aten::full", "aten::ones"):
        st = [s for s in (e.stack or []) if "diffulab_amd" in s or "scripts/" in s]
        cnt[(e.name, st[0] if st else (e.stack[0] if e.stack else "?"))] += 1
for (n, s), c in cnt.most_common(30):
    print(f"{c:4d} {n:18s} {s}")
