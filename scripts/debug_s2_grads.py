"""one-off diagnostic (GPU box): full per-tensor gradient error of the HIP DiT-S/2 path vs the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import diffusion as od, dit as odit, synth
from diffulab_amd import MMDiT, Diffuser

cfgk = dict(input_channels=4, output_channels=4, inner_dim=384, embedding_dim=384, num_heads=6, mlp_ratio=4,
            patch_size=2, depth=12, n_classes=1000, classifier_free=True)
cfg = odit.DiTConfig(**cfgk)
P = synth.dit_params(odit.param_shapes(cfg), seed=7)
m = MMDiT(simple_dit=True, **cfgk); m.load_state_dict(P); m = m.cuda()
B = 2
x0 = synth.normal("s2.x0", (B, 4, 32, 32)); noise = synth.normal("s2.noise", (B, 4, 32, 32))
t = synth.uniform("s2.t", (B,), lo=0.05, hi=0.95); y = synth.integers("s2.y", (B,), 1000)
d = Diffuser(m, "euler", n_steps=50)
loss = d.compute_loss({"x": x0.cuda(), "y": y.cuda(), "p": 0.0}, timesteps=t, noise=noise.cuda())["loss"]
loss.backward()
Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
pred = odit.dit_forward(Pr, od.flow_add_noise(x0, t, noise), t, y, cfg)
ref = od.flow_loss(pred, x0, noise); ref.backward()
print("loss", loss.item(), ref.item())
errs = []
for n, p in m.named_parameters():
    a, b = p.grad.double().cpu(), Pr[n].grad.double()
    errs.append(((a - b).norm() / b.norm()).item())
    if "layers.10.mlp_input.2" in n:
        s = slice(None, None, max(1, a.numel() // 512))
        aa, bb = a.flatten()[s][:512], b.flatten()[s][:512]
        print("sample rel", ((aa - bb).norm() / bb.norm()).item(), "sample norm", bb.norm().item(), "full norm", b.norm().item())
        idx = (aa - bb).abs().argsort(descending=True)[:5]
        print("worst sample entries", [(int(i), aa[i].item(), bb[i].item()) for i in idx])
names = [n for n, _ in m.named_parameters()]
top = sorted(zip(errs, names), reverse=True)[:10]
print("worst full-tensor rel-L2:", top)
print("median", sorted(errs)[len(errs) // 2])
