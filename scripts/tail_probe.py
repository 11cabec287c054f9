"""How much of the weight-gradient work is left on the side stream when the main chain finishes the last block's backward
(DL_TAIL_PROBE=1 makes the engine record one event per stream at the join):  python scripts/tail_probe.py"""
import os, sys
os.environ["DL_TAIL_PROBE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
tails = []
for i in range(20):
    opt.zero_grad()
    t = d.draw_timesteps(B).to(dev, non_blocking=True)
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    losses = d.compute_loss({"x": x0, "y": y, "p": 0.1}, timesteps=t)
    sum(losses.values()).backward()
    opt.step()
    torch.cuda.synchronize()
    em, es = m.engine._tail_probe
    if i >= 5:
        tails.append((e0.elapsed_time(em), em.elapsed_time(es)))
print("start -> main chain at the join (ms), join -> side stream drained (ms):")
for a, b in tails[:6]:
    print(f"  {a:7.2f}  {b:7.2f}")
print("mean tail", sum(b for _, b in tails) / len(tails))
