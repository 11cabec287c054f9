"""LAB: Diffuser.generate through the captured hipGraph forward vs the eager forward on the code paths added in round 5 (UNet attention
above 64 tokens, padded DiT / DDT attention on odd token grids): same samples."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import diffulab_amd as da  # noqa: E402
from diffulab_amd import Diffuser  # noqa: E402
from diffulab_amd.networks.embedders import PrecomputedEmbedder  # noqa: E402

DEV = "cuda"


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def build(kind):
    torch.manual_seed(0)
    if kind == "unet_small":
        m = da.UNetModel(image_size=[32, 32], in_channels=3, model_channels=64, out_channels=3, num_res_blocks=1, attention_resolutions=[4],
                         channel_mult="1, 2, 2", num_heads=4, use_scale_shift_norm=True, resblock_updown=True, n_classes=10, classifier_free=True)
        d = Diffuser(m, sampling_method="ddpm", model_type="gaussian_diffusion", n_steps=1000)
        d.set_steps(4)
        data = lambda: {"x": torch.randn(4, 3, 32, 32, generator=torch.Generator().manual_seed(1)).to(DEV), "y": torch.arange(4, device=DEV)}  # noqa: E731
    elif kind == "unet_attn256":
        m = da.UNetModel(image_size=[32, 32], in_channels=3, model_channels=64, out_channels=3, num_res_blocks=1, attention_resolutions=[2, 4],
                         channel_mult="1, 2, 2", num_heads=4, use_scale_shift_norm=True, resblock_updown=True, n_classes=10, classifier_free=True)
        d = Diffuser(m, sampling_method="ddpm", model_type="gaussian_diffusion", n_steps=1000)
        d.set_steps(4)
        data = lambda: {"x": torch.randn(4, 3, 32, 32, generator=torch.Generator().manual_seed(1)).to(DEV), "y": torch.arange(4, device=DEV)}  # noqa: E731
    elif kind == "dit_12x12":
        m = da.MMDiT(simple_dit=True, input_channels=4, output_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, mlp_ratio=4, patch_size=2,
                     depth=2, n_classes=10, classifier_free=True)
        d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=5)
        data = lambda: {"x": torch.randn(4, 4, 24, 24, generator=torch.Generator().manual_seed(1)).to(DEV), "y": torch.arange(4, device=DEV)}  # noqa: E731
    else:
        emb = PrecomputedEmbedder(torch.randn(1, 77, 32) * 0.5, null_embedding_seq_len=7)
        m = da.DDT(simple_ddt=False, context_embedder=emb, input_channels=16, output_channels=16, inner_dim=128, num_heads=2, mlp_ratio=2,
                   patch_size=1, encoder_depth=1, decoder_depth=2, classifier_free=True, rope_axes_dim=[16, 24, 24], rope_base=1000)
        d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=5)
        g = torch.Generator().manual_seed(2)
        ctx = {"embeddings": (torch.randn(2, 77, 32, generator=g) * 0.5).to(DEV), "attn_mask": (torch.arange(77)[None] < torch.tensor([77, 30])[:, None]).to(DEV)}
        data = lambda: {"x": torch.randn(2, 16, 24, 40, generator=torch.Generator().manual_seed(1)).to(DEV), "initial_context": ctx}  # noqa: E731
    with torch.no_grad():
        for q in m.parameters():
            if float(q.abs().sum()) == 0:
                q.normal_(0, 0.05)
    return m.to(DEV).eval(), d, data


for kind in ("unet_attn256", "unet_small", "dit_12x12", "ddt_joint_24x40"):
    for gs, pair in ((2.0, "1"), (2.0, "0"), (0.0, "1")):
        outs = {}
        for graph in ("1", "0", "0b"):
            os.environ["DL_HIPGRAPH"] = graph[0]
            os.environ["DL_CFG_PAIR"] = pair
            m, d, data = build(kind)
            torch.manual_seed(5)  # (stochastic samplers draw from the device generator)
            outs[graph] = d.generate(data(), use_tqdm=False, guidance_scale=gs)["x"].float()
        e, e0 = rel(outs["1"], outs["0"]), rel(outs["0b"], outs["0"])
        print(f"{kind:18s} guidance {gs} pair {pair}: hipGraph vs eager rel {e:.2e} (eager vs eager {e0:.2e}) finite {bool(torch.isfinite(outs['1']).all())}  |x| {outs['1'].abs().mean().item():.3f}", flush=True)
print("done")
