"""Training-step time of the DiT configurations of BASELINE.json on one MI355X (secondary numbers next to bench.py):
    python scripts/train_step_bench.py cifar --batch 32     # configs/train_cifar10_flow_matching.yaml (dit.yaml dims, RGB 32x32)
    python scripts/train_step_bench.py s2 --batch 256       # the headline DiT-S/2 workload (same step as bench.py)
    python scripts/train_step_bench.py repa --batch 128     # DiT-B/REPA dims (768/12 heads/12 blocks, 32x8x8 latents) + REPA loss
    python scripts/train_step_bench.py sprint --batch 32    # configs/model/sprint.yaml (512/8, 2+8+2 blocks, 75 % of the tokens skip the deep blocks)
    python scripts/train_step_bench.py ddt --batch 256      # configs/model/ddt.yaml (512/8, 8 encoder + 4 per-token-conditioned decoder blocks)
    python scripts/train_step_bench.py dit12 --batch 32     # the same 12 blocks without token dropping (what SPRINT is compared with)
    python scripts/train_step_bench.py joint --batch 16     # joint text-image MMDiT: 768/12 heads, 12 MMDiTBlocks, 128x32x32 latents at patch 1
                                                            # (1024 image tokens) + 128 text tokens of width 1024, ragged key mask
    python scripts/train_step_bench.py sprint_joint --batch 16  # configs/train_imagenet_repa_txt_to_img_sprint.yaml: 768/12, 2 joint encoder +
                                                            # 8 single-stream deep (256 of 1024 image tokens) + 2 joint decoder blocks
    python scripts/train_step_bench.py ddt_joint --batch 16    # configs/train_imagenet_repa_txt_to_img.yaml: 640/10 heads, 8 joint encoder + 4 decoder blocks
    python scripts/train_step_bench.py repa_rs --batch 128  # same + the Perceiver resampler (configs/train_imagenet_flow_matching_repa.yaml)
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from diffulab_amd import DDT, Diffuser, MMDiT, SprintDiT  # noqa: E402
from diffulab_amd.training import FusedAdamW  # noqa: E402
from diffulab_amd.training.losses import RepaLoss  # noqa: E402

CFG = {
    "cifar": (dict(input_channels=3, output_channels=3, inner_dim=512, embedding_dim=512, num_heads=8, mlp_ratio=4, patch_size=2,
                   depth=10, n_classes=10, classifier_free=False), (3, 32, 32)),
    "s2": (dict(input_channels=4, output_channels=4, inner_dim=384, embedding_dim=384, num_heads=6, mlp_ratio=4, patch_size=2,
                depth=12, n_classes=1000, classifier_free=True), (4, 32, 32)),
    "repa": (dict(input_channels=32, output_channels=32, inner_dim=768, embedding_dim=256, num_heads=12, mlp_ratio=4, patch_size=1,
                  depth=12, n_classes=1000, classifier_free=True), (32, 8, 8)),
}
CFG["repa_rs"] = CFG["repa"]
CFG["dit12"] = (dict(CFG["cifar"][0], depth=12), (3, 32, 32))
SPRINT = dict(input_channels=3, output_channels=3, inner_dim=512, embedding_dim=512, num_heads=8, mlp_ratio=4, patch_size=2,
              encoder_depth=2, deep_layers_depth=8, decoder_depth=2, n_classes=10, classifier_free=False, drop_rate=0.75)
CFG["sprint"] = (SPRINT, (3, 32, 32))
CFG["ddt"] = (dict(input_channels=3, output_channels=3, inner_dim=512, num_heads=8, mlp_ratio=4, patch_size=2, encoder_depth=8,
                   decoder_depth=4, n_classes=10, classifier_free=False), (3, 32, 32))
JOINT = dict(input_channels=128, output_channels=128, inner_dim=768, embedding_dim=768, num_heads=12, mlp_ratio=4, patch_size=1,
             depth=12, classifier_free=True, rope_base=2000, rope_axes_dim=[16, 24, 24])
CFG["joint"] = (JOINT, (128, 32, 32))
SPRINT_JOINT = dict(input_channels=128, output_channels=128, inner_dim=768, embedding_dim=768, num_heads=12, mlp_ratio=4, patch_size=1,
                    encoder_depth=2, deep_layers_depth=8, n_single_stream_blocks=8, decoder_depth=2, classifier_free=True,
                    rope_base=2000, rope_axes_dim=[16, 24, 24])
CFG["sprint_joint"] = (SPRINT_JOINT, (128, 32, 32))
CFG["ddt_joint"] = (dict(input_channels=128, output_channels=128, inner_dim=640, num_heads=10, mlp_ratio=4, patch_size=1, encoder_depth=8,
                         decoder_depth=4, classifier_free=True, rope_base=1000, rope_axes_dim=[20, 22, 22]), (128, 32, 32))
RS = dict(depth=3, dim=1024, head_dim=64, num_heads=8, ff_mult=4, num_latents=256)


def main() -> None:
    if os.environ.get("DL_LAB_ATTN_PIPE"):  # LAB A/B: 0 = the chain forms of the attention kernels everywhere
        from diffulab_amd import ops
        ops.lib().cdll.dl_lab_set_attn_pipe(int(os.environ["DL_LAB_ATTN_PIPE"]))
    if os.environ.get("DL_LAB_NT_DEEP"):  # LAB A/B: 0 = the two-slot ring of the 128 x 128 GEMM kernel everywhere
        from diffulab_amd import ops
        ops.lib().cdll.dl_lab_set_nt_deep(int(os.environ["DL_LAB_NT_DEEP"]))
    if os.environ.get("DL_LAB_TN_SPLIT_MODEL"):  # LAB A/B: 0 = the workgroup-count split rules of the atomic weight-gradient GEMMs
        from diffulab_amd import ops
        ops.lib().cdll.dl_lab_set_tn_split_model(int(os.environ["DL_LAB_TN_SPLIT_MODEL"]))
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=list(CFG))
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--graph", action="store_true", help="the whole step as one hipGraph (training/graph_step.py)")
    ap.add_argument("--hi-prio", action="store_true", help="run the step on a high-priority HIP stream (the engines' side stream keeps the default priority)")
    a = ap.parse_args()
    dev = "cuda"
    kw, shape = CFG[a.config]
    torch.manual_seed(0)
    ctx = None
    if a.config in ("joint", "sprint_joint", "ddt_joint"):
        from diffulab_amd.networks.embedders import PrecomputedEmbedder

        Lc, Cd = 128, 1024
        emb = PrecomputedEmbedder(torch.randn(1, Lc, Cd), 7)
        if a.config == "ddt_joint":
            m = DDT(simple_ddt=False, context_embedder=emb, **kw).to(dev)
        else:
            m = (MMDiT if a.config == "joint" else SprintDiT)(simple_dit=False, context_embedder=emb, **kw).to(dev)
        keep = torch.arange(Lc, device=dev)[None, :] < torch.randint(8, Lc + 1, (a.batch, 1), device=dev)
        ctx = {"embeddings": torch.randn(a.batch, Lc, Cd, device=dev, dtype=torch.bfloat16), "attn_mask": keep}
    elif a.config == "ddt":
        m = DDT(simple_ddt=True, **kw).to(dev)
    else:
        m = (SprintDiT if a.config == "sprint" else MMDiT)(simple_dit=True, **kw).to(dev)
    extra, params = [], list(m.parameters())
    if a.config.startswith("repa"):
        rs = a.config == "repa_rs"
        rl = RepaLoss(alignment_layer=8, denoiser_dimension=768, hidden_dim=1024, load_dino=False, embedding_dim=1024, coeff=0.5,
                      use_resampler=rs, resampler_params=RS if rs else None).to(dev)
        rl.set_model(m)
        extra, params = [rl], params + list(rl.parameters())
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=50, extra_args={"logits_normal": True},
                 extra_losses=extra)
    opt = FusedAdamW(params, lr=1e-4, weight_decay=0.01)
    x0 = torch.randn(a.batch, *shape, device=dev)
    y = torch.randint(0, kw["n_classes"], (a.batch,), device=dev) if kw.get("n_classes") else None
    dst = torch.randn(a.batch, 256 if a.config == "repa_rs" else 64, 1024, device=dev) if a.config.startswith("repa") else None
    p = 0.1 if kw["classifier_free"] else 0.0

    def step():
        opt.zero_grad()
        t = d.draw_timesteps(a.batch).to(dev, non_blocking=True)
        inputs = {"x": x0, "initial_context": ctx, "p": p} if ctx is not None else {"x": x0, "y": y, "p": p}
        losses = d.compute_loss(inputs, timesteps=t, extra_args={"dst_features": dst} if dst is not None else {})
        sum(losses.values()).backward()
        opt.step()

    if a.graph:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab"))
        from graph_step import GraphedTrainStep  # scripts/lab (lab code, not part of the package)

        gs = GraphedTrainStep(d, opt, warmup=3)

        def step():  # noqa: F811
            t = d.draw_timesteps(a.batch).to(dev, non_blocking=True)
            inputs = {"x": x0, "initial_context": ctx, "p": p} if ctx is not None else {"x": x0, "y": y, "p": p}
            gs(inputs, t, {"dst_features": dst} if dst is not None else {})

    if a.hi_prio:
        torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_issue = (time.perf_counter() - t0) / a.steps  # host time to issue a step (the GPU runs behind it): == ms_per_step when launch-bound
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"config": a.config, "batch": a.batch, "graph": bool(a.graph and any(v not in (None, False) for v in gs._graphs.values())) if a.graph else False, "ms_per_step": round(dt * 1e3, 3), "images_per_s": round(a.batch / dt, 1),
                      "host_issue_ms_per_step": round(t_issue * 1e3, 3),
                      "params_M": round(sum(q.numel() for q in m.parameters()) / 1e6, 1)}))


if __name__ == "__main__":
    main()
