"""20-step guided Euler sampling (40 forwards) at small batch with and without the hipGraph replay of the inference forward:
    DL_HIPGRAPH=0 python scripts/graph_sampler_bench.py ; python scripts/graph_sampler_bench.py"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from diffulab_amd import Diffuser, SprintDiT, DDT, MMDiT
from diffulab_amd.networks.embedders import PrecomputedEmbedder
dev = "cuda"
def bench(name, m, inputs, shape, B):
    m = m.to(dev).eval()
    d = Diffuser(m, sampling_method="euler", model_type="rectified_flow", n_steps=20)
    def run():
        return d.generate({"x": torch.randn(B, *shape, device=dev), **inputs}, use_tqdm=False, guidance_scale=2.0)
    run(); torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize()
    print(name, f"DL_HIPGRAPH={os.environ.get('DL_HIPGRAPH','1')}", f"{(time.perf_counter()-t0)/40*1e3:.3f} ms/forward", flush=True)
B = 4
y = torch.randint(0, 10, (B,), device=dev)
bench("sprint", SprintDiT(simple_dit=True, input_channels=3, inner_dim=512, embedding_dim=512, num_heads=8, patch_size=2, encoder_depth=2, deep_layers_depth=8, decoder_depth=2, n_classes=10, classifier_free=True), {"y": y}, (3, 32, 32), B)
bench("ddt", DDT(simple_ddt=True, input_channels=3, inner_dim=512, num_heads=8, patch_size=2, encoder_depth=8, decoder_depth=4, n_classes=10, classifier_free=True), {"y": y}, (3, 32, 32), B)
ctx = {"embeddings": torch.randn(16, 128, 1024, device=dev), "attn_mask": torch.ones(16, 128, dtype=torch.bool, device=dev)}
emb = PrecomputedEmbedder(torch.randn(1, 128, 1024), 7)
bench("sprint_joint", SprintDiT(simple_dit=False, context_embedder=emb, input_channels=128, inner_dim=768, embedding_dim=768, num_heads=12, patch_size=1, encoder_depth=2, deep_layers_depth=8, n_single_stream_blocks=8, decoder_depth=2, classifier_free=True, rope_base=2000, rope_axes_dim=[16, 24, 24]), {"initial_context": ctx}, (128, 32, 32), 16)
