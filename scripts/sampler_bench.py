"""Secondary metric: sampler throughput of DiT-S/2 -- images/s of the 50-step Euler loop with classifier-free
guidance 4 (two denoiser forwards per step, `Flow.denoise` flow.py:410-524) on 4x32x32 latents, one MI355X."""

import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from diffulab_amd import Diffuser, MMDiT  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--guidance", type=float, default=4.0)
    a = ap.parse_args()
    dev = "cuda"
    torch.manual_seed(0)
    m = MMDiT(simple_dit=True, input_channels=4, output_channels=4, inner_dim=384, embedding_dim=384, num_heads=6, mlp_ratio=4,
              patch_size=2, depth=12, n_classes=1000, classifier_free=True).to(dev).eval()
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=a.steps)
    y = torch.randint(0, 1000, (a.batch,), device=dev)

    def run():
        return d.generate({"y": y}, data_shape=(a.batch, 4, 32, 32), use_tqdm=False, guidance_scale=a.guidance)["x"]

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        out = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    nfwd = a.steps * (2 if a.guidance > 0 else 1)
    print(json.dumps({"workload": f"DiT-S/2 {a.steps}-step Euler sampling, CFG {a.guidance}", "batch": a.batch, "s_per_batch": dt,
                      "images_per_s": a.batch / dt, "ms_per_forward": dt / nfwd * 1e3, "finite": bool(torch.isfinite(out).all())}))


if __name__ == "__main__":
    main()
