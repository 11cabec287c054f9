"""Would two independent half-batch chains on two streams fill each other's bubbles?  Two DiT-S/2 replicas at B=128 stepping
concurrently (one Python thread and one HIP stream each) against one replica at B=256 / B=128.  python scripts/two_chain_probe.py"""
import os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffulab_amd import Diffuser, MMDiT  # noqa: E402
from diffulab_amd.training import FusedAdamW  # noqa: E402

KW = dict(input_channels=4, output_channels=4, inner_dim=384, embedding_dim=384, num_heads=6, mlp_ratio=4, patch_size=2,
          depth=12, n_classes=1000, classifier_free=True)
dev = "cuda"


def make(batch):
    m = MMDiT(simple_dit=True, **KW).to(dev)
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=50, extra_args={"logits_normal": True})
    opt = FusedAdamW(list(m.parameters()), lr=1e-4, weight_decay=0.01)
    x0 = torch.randn(batch, 4, 32, 32, device=dev)
    y = torch.randint(0, 1000, (batch,), device=dev)

    def step():
        opt.zero_grad()
        t = d.draw_timesteps(batch).to(dev, non_blocking=True)
        losses = d.compute_loss({"x": x0, "y": y, "p": 0.1}, timesteps=t)
        sum(losses.values()).backward()
        opt.step()
    return step


def run(steps_fns, n=20, warm=5):
    streams = [torch.cuda.Stream() for _ in steps_fns]

    def loop(fn, s, k):
        with torch.cuda.stream(s):
            for _ in range(k):
                fn()
    for phase, k in (("warm", warm), ("timed", n)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=loop, args=(f, s, k)) for f, s in zip(steps_fns, streams)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt / n


for name, batches in (("1 x B=256", [256]), ("1 x B=128", [128]), ("2 x B=128 concurrent", [128, 128])):
    fns = [make(b) for b in batches]
    ms = run(fns) * 1e3
    print(f"{name:24s}: {ms:7.2f} ms per round, {sum(batches) / ms * 1e3:8.0f} img/s", flush=True)
