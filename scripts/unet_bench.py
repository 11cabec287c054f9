"""Training-step time of the MNIST-DDPM UNet (configs/train_mnist_ddpm.yaml: UNet 32x32x1, mc=128, B=128) on one MI355X.
Not the headline benchmark (bench.py is DiT-S/2); a secondary number for the UNet row of SURVEY.md §8."""

import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from diffulab_amd import Diffuser
from diffulab_amd.networks.denoisers import UNetModel
from diffulab_amd.training.optim import FusedAdamW


def main() -> None:
    if os.environ.get("DL_LAB_CONV_BIG"):  # LAB A/B: 0 = the 128x128 implicit-GEMM kernel everywhere, 1 = conv3x3_big_k where it applies
        from diffulab_amd import ops

        ops.lib().cdll.dl_lab_set_conv_big(int(os.environ["DL_LAB_CONV_BIG"]))
    if os.environ.get("DL_LAB_CONV_HALO"):  # LAB A/B: 1 = conv3x3_halo_k where the shape fits (one activation staging per channel chunk)
        from diffulab_amd import ops

        ops.lib().cdll.dl_lab_set_conv_halo(int(os.environ["DL_LAB_CONV_HALO"]))
    if os.environ.get("DL_LAB_GN_FUSED"):  # LAB A/B: 1 = 128-channel slabs (default), 2 = 512-channel slabs only, 0 = three launches
        from diffulab_amd import ops

        ops.lib().cdll.dl_lab_set_gn_fused(int(os.environ["DL_LAB_GN_FUSED"]))
    if os.environ.get("DL_LAB_NT_DEEP"):  # LAB A/B: 0 = the two-slot ring of the 128 x 128 GEMM kernel everywhere
        from diffulab_amd import ops

        ops.lib().cdll.dl_lab_set_nt_deep(int(os.environ["DL_LAB_NT_DEEP"]))
    if os.environ.get("DL_LAB_WGRAD_HALO"):  # LAB A/B: 0 = the implicit-GEMM weight-gradient kernels of gemm.hip everywhere
        from diffulab_amd import ops

        ops.lib().cdll.dl_lab_set_wgrad_halo(int(os.environ["DL_LAB_WGRAD_HALO"]))
    if os.environ.get("DL_LAB_TN_SPLIT_MODEL"):  # LAB A/B: 0 = the workgroup-count split rules of the atomic weight-gradient GEMMs
        from diffulab_amd import ops

        ops.lib().cdll.dl_lab_set_tn_split_model(int(os.environ["DL_LAB_TN_SPLIT_MODEL"]))
    if os.environ.get("DL_LAB_CONV_SPLIT128"):  # LAB A/B: 0 = split convolution launches always on 256-wide tiles
        from diffulab_amd import ops

        ops.lib().cdll.dl_lab_set_conv_split128(int(os.environ["DL_LAB_CONV_SPLIT128"]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--graph", action="store_true", help="LAB: the whole step replayed as one hipGraph (scripts/lab/graph_step.py)")
    a = ap.parse_args()
    dev = "cuda"
    torch.manual_seed(0)
    m = UNetModel(image_size=[32, 32], in_channels=1, model_channels=128, out_channels=1, num_res_blocks=2,
                  attention_resolutions=[4, 8, 16], num_heads=2, resblock_updown=True, n_classes=10,
                  use_scale_shift_norm=True, classifier_free=False).to(dev)
    gd = Diffuser(m, sampling_method="ddpm", model_type="gaussian_diffusion", n_steps=1000)
    opt = FusedAdamW(m.parameters(), lr=1e-4)
    x0 = torch.randn(a.batch, 1, 32, 32, device=dev)
    y = torch.randint(0, 10, (a.batch,), device=dev)

    def step():
        opt.zero_grad()
        loss = gd.compute_loss({"x": x0, "y": y, "p": 0.0}, timesteps=gd.draw_timesteps(a.batch))["loss"]
        loss.backward()
        opt.step()
        return loss

    if a.graph:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab"))
        from graph_step import GraphedTrainStep

        gstep = GraphedTrainStep(gd, opt, warmup=3)

        def step():  # noqa: F811
            return gstep({"x": x0, "y": y, "p": 0.0}, gd.draw_timesteps(a.batch))["loss"]

        for _ in range(5):
            step()
        torch.cuda.synchronize()
        print("graph captured:", [type(v).__name__ if v is False else "graph" for v in gstep._graphs.values()])
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    t_issue = (time.perf_counter() - t0) / a.steps  # host time to ISSUE a step (no synchronisation inside the loop)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"workload": "unet-mnist-ddpm train step", "batch": a.batch, "ms_per_step": dt * 1e3,
                      "images_per_s": a.batch / dt, "host_issue_ms_per_step": t_issue * 1e3,
                      "params_M": sum(p.numel() for p in m.parameters()) / 1e6, "loss": float(loss)}))
    if os.environ.get("UNET_HOST_SPLIT"):  # LAB: host time to ISSUE each phase of a step (no synchronisation)
        acc = {"zero_grad": 0.0, "forward": 0.0, "backward": 0.0, "optimizer": 0.0}
        torch.cuda.synchronize()
        n = 10
        for _ in range(n):
            t0 = time.perf_counter()
            opt.zero_grad()
            t1 = time.perf_counter()
            loss = gd.compute_loss({"x": x0, "y": y, "p": 0.0}, timesteps=gd.draw_timesteps(a.batch))["loss"]
            t2 = time.perf_counter()
            loss.backward()
            t3 = time.perf_counter()
            opt.step()
            t4 = time.perf_counter()
            for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                acc[k] += v
        torch.cuda.synchronize()
        print("host issue ms per step:", {k: round(v / n * 1e3, 2) for k, v in acc.items()})
        import threading

        import cProfile
        import io
        import pstats

        prof = cProfile.Profile()
        threading.setprofile(lambda *a_: None)  # (the autograd worker thread exists already: profile the backward by running the engine's backward directly)
        m.zero_grad()
        loss = gd.compute_loss({"x": x0, "y": y, "p": 0.0}, timesteps=gd.draw_timesteps(a.batch))["loss"]
        torch.cuda.synchronize()
        eng = m.engine
        dpred = torch.randn(a.batch, 1, 32, 32, device=dev)
        m._prepare_grads()
        prof.enable()
        eng.backward(dpred)
        prof.disable()
        torch.cuda.synchronize()
        st = io.StringIO()
        pstats.Stats(prof, stream=st).sort_stats("tottime").print_stats(28)
        print(st.getvalue()[:6500])
    if os.environ.get("UNET_HOST_PROFILE"):  # LAB: where the host time of a step goes
        import cProfile
        import io
        import pstats

        pr = cProfile.Profile()
        pr.enable()
        for _ in range(3):
            step()
        pr.disable()
        torch.cuda.synchronize()
        st = io.StringIO()
        pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(30)
        print(st.getvalue()[:7000])


if __name__ == "__main__":
    main()
