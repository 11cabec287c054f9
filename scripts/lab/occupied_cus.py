"""LAB: what a communication kernel's resident workgroups cost the training step.  Every heavy kernel of the step is one workgroup per
CU with the CU's whole register file, so a foreign workgroup on c CUs leaves 256 - c CUs for a launch sized for 256.  A spinner
(dl_probe_spin: n workgroups that only hold their slots) runs on its own stream during the step; persistent grids are capped with
DL_MAX_WGS-style arguments where the engine exposes them.   python scripts/lab/occupied_cus.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import _probe_lib
from diffulab_amd import Diffuser, MMDiT
from diffulab_amd.training import FusedAdamW

dev = "cuda"
kw = dict(input_channels=4, output_channels=4, inner_dim=384, embedding_dim=384, num_heads=6, mlp_ratio=4, patch_size=2, depth=12,
          n_classes=1000, classifier_free=True)
m = MMDiT(simple_dit=True, **kw).to(dev)
d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=50, extra_args={"logits_normal": True})
opt = FusedAdamW(m.parameters(), lr=1e-4, weight_decay=0.01)
B = 256
x0 = torch.randn(B, 4, 32, 32, device=dev); y = torch.randint(0, 1000, (B,), device=dev)
P = _probe_lib.lib()
P.cdll.dl_probe_spin.argtypes = [__import__("ctypes").c_int] * 3 + [__import__("ctypes").c_void_p]
spin = torch.cuda.Stream()

def step():
    opt.zero_grad()
    t = d.draw_timesteps(B).to(dev, non_blocking=True)
    losses = d.compute_loss({"x": x0, "y": y, "p": 0.1}, timesteps=t)
    sum(losses.values()).backward()
    opt.step()

def run(n_wgs, threads, frac, steps=10):
    """spinner of n_wgs workgroups alive for `frac` of every step (started with the step, like a bucket's all-reduce)"""
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        if n_wgs:
            spin.wait_stream(torch.cuda.current_stream())
            P.call("dl_probe_spin", n_wgs, threads, int(21000 * frac), spin.cuda_stream)
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

cap = os.environ.get("DL_MAIN_WGS", "")
base = run(0, 0, 0)
print(f"persistent main-chain grids: {cap or 'all CUs'};  no spinner: {base:.2f} ms/step")
for n, th, fr in ((8, 256, 0.5), (16, 256, 0.5), (32, 256, 0.5), (32, 512, 0.5), (64, 256, 0.5), (32, 256, 0.1), (32, 256, 1.0)):
    ms = run(n, th, fr)
    print(f"spinner {n:3d} WGs x {th} threads for {fr:.0%} of the step: {ms:.2f} ms/step ({(ms / base - 1) * 100:+.1f} %)")
