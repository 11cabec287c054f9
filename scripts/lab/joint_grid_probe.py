"""LAB: which image-token grids the joint text-image engines (and the class-conditional DiT) take in the bf16 regime."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import diffulab_amd as da  # noqa: E402
from diffulab_amd.networks.embedders import PrecomputedEmbedder  # noqa: E402

DEV = "cuda"
kw = dict(input_channels=16, output_channels=16, inner_dim=128, num_heads=2, mlp_ratio=2, patch_size=1, classifier_free=True,
          rope_axes_dim=[16, 24, 24], rope_base=2000)
for gh, gw in ((12, 12), (6, 10), (28, 36), (24, 40), (20, 48), (32, 32), (36, 28), (16, 24), (10, 10)):
    res = []
    for fam in ("mmdit_joint", "sprint_joint", "ddt_joint", "dit"):
        emb = PrecomputedEmbedder(torch.randn(1, 77, 32) * 0.5, null_embedding_seq_len=7)
        if fam == "mmdit_joint":
            m = da.MMDiT(simple_dit=False, context_embedder=emb, embedding_dim=64, depth=2, n_single_stream_blocks=1, **kw)
        elif fam == "sprint_joint":
            m = da.SprintDiT(simple_dit=False, context_embedder=emb, embedding_dim=64, encoder_depth=1, deep_layers_depth=2,
                             n_single_stream_blocks=1, decoder_depth=1, drop_rate=0.75, **kw)
        elif fam == "ddt_joint":
            m = da.DDT(simple_ddt=False, context_embedder=emb, encoder_depth=1, decoder_depth=1, **kw)
        else:
            k2 = {k: v for k, v in kw.items() if k not in ("rope_axes_dim", "rope_base")}
            m = da.MMDiT(simple_dit=True, embedding_dim=64, depth=2, n_classes=10, **k2)
        m = m.to(DEV).train()
        B = 2
        x = torch.randn(B, 16, gh, gw, device=DEV)
        t = torch.rand(B, device=DEV)
        try:
            if fam == "dit":
                out = m(x=x, timesteps=t, y=torch.randint(0, 10, (B,), device=DEV), p=0.0)["x"]
            else:
                ic = {"embeddings": torch.randn(B, 77, 32, device=DEV) * 0.5, "attn_mask": torch.arange(77, device=DEV)[None] < torch.tensor([77, 20], device=DEV)[:, None]}
                out = m(x=x, timesteps=t, initial_context=ic, p=0.0)["x"]
            out.sum().backward()
            torch.cuda.synchronize()
            res.append(f"{fam}: ok")
        except NotImplementedError as e:
            res.append(f"{fam}: REFUSED ({str(e)[:70]})")
        except Exception as e:  # noqa: BLE001
            res.append(f"{fam}: CRASH {type(e).__name__} {str(e)[:90]}")
    print(f"grid {gh}x{gw} = {gh * gw} tokens:", " | ".join(res), flush=True)
