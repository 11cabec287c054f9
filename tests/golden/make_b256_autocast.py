"""Fixture `dit_b256_autocast.npz`: what the oracle's bf16-autocast leg (``oracle.dit.bf16_autocast()``: bf16 matmuls and residual
stream, f32 statistics -- pinned to the REFERENCE under ``torch.autocast("cpu", bfloat16)`` by dit_autocast.npz /
tests/test_oracle_golden.py, which show it is the stricter yardstick) loses against the fp32 oracle at the BENCHED shape, DiT-S/2 at
B = 256, per parameter tensor and on the loss.  tests/test_parity_bf16_gpu.py::test_full_training_step_at_the_benched_shape_b256 used to
run this leg on the GPU box's host in every run of the suite (~2 of its 3 minutes); the numbers depend on the seeded inputs only.

    python tests/golden/make_b256_autocast.py [out.npz]        (needs ~100 GB of host memory: run where the GPU suite runs)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import diffusion as od  # noqa: E402
from oracle import dit as odit  # noqa: E402
from oracle import synth  # noqa: E402

S2 = dict(input_channels=4, output_channels=4, inner_dim=384, embedding_dim=384, num_heads=6, mlp_ratio=4, patch_size=2, depth=12,
          n_classes=1000, classifier_free=True)


def main() -> None:
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "dit_b256_autocast.npz")
    tag, B, H, seed = "pb.b256", 256, 32, 7
    cfg = odit.DiTConfig(**S2)
    P = synth.dit_params(odit.param_shapes(cfg), seed=seed)
    x0, noise = synth.normal(f"{tag}.x0", (B, 4, H, H)), synth.normal(f"{tag}.noise", (B, 4, H, H))
    t = synth.uniform(f"{tag}.t", (B,), lo=0.05, hi=0.95)
    y = synth.integers(f"{tag}.y", (B,), 1000)
    legs = {}
    for leg in ("fp32", "bf16"):
        Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        t0 = time.time()
        if leg == "bf16":
            with odit.bf16_autocast():
                loss = od.flow_loss(odit.dit_forward(Pr, od.flow_add_noise(x0, t, noise), t, y, cfg), x0, noise)
        else:
            loss = od.flow_loss(odit.dit_forward(Pr, od.flow_add_noise(x0, t, noise), t, y, cfg), x0, noise)
        loss.backward()
        legs[leg] = (loss.item(), {k: v.grad.double() for k, v in Pr.items()})
        print(f"{leg} leg: {time.time() - t0:.1f} s, loss {loss.item():.6f}", flush=True)
    names = list(legs["fp32"][1])
    g32, gbf = legs["fp32"][1], legs["bf16"][1]
    err = np.array([((gbf[k] - g32[k]).norm() / g32[k].norm().clamp_min(1e-30)).item() for k in names])
    flat = lambda g: torch.cat([g[k].flatten() for k in sorted(g)])  # noqa: E731
    whole = ((flat(gbf) - flat(g32)).norm() / flat(g32).norm()).item()
    np.savez(out_path, names=np.array(names), err=err, loss_fp32=legs["fp32"][0], loss_bf16=legs["bf16"][0], whole_grad_err=whole,
             grad_norms_fp32=np.array([g32[k].norm().item() for k in names]))
    print(f"wrote {out_path}: per-tensor error median {np.median(err):.3e} max {err.max():.3e}; whole gradient {whole:.3e}")


if __name__ == "__main__":
    main()
