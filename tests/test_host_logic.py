"""CPU tests of the host-side logic of the product package (no kernels are called): schedule tables, timestep draws,
respacing, module/state_dict contract, flat-arena layout, optimizer/arena plumbing.  Expected values are the committed
reference outputs (tests/golden/schedules.npz) -- bit-exact."""

import copy
import os

import numpy as np
import pytest
import torch

from diffulab_amd.diffuse.modelizations.flow import Flow
from diffulab_amd.diffuse.modelizations.gaussian_diffusion import GaussianDiffusion
from diffulab_amd.diffuse.modelizations.utils import space_timesteps
from diffulab_amd.engine import DiTDims, ParamLayout, rope_grid_tables
from diffulab_amd.networks.denoisers import MMDiT


def t2n(x):
    return x.detach().numpy()


def test_flow_grid_and_draws_match_reference(golden):
    g = golden("schedules")
    for n in (4, 50, 100):
        assert np.array_equal(np.array(Flow(n_steps=n).timesteps), g[f"flow_ts_n{n}"])
        for sh in (4.63, 6.93):
            f = Flow(n_steps=n, shift=sh)
            assert np.array_equal(np.array(f.timesteps), g[f"flow_ts_ctor_n{n}_shift{sh}"])  # ctor shift: draws only
            f.set_steps(n, shift=sh)
            assert np.array_equal(np.array(f.timesteps), g[f"flow_ts_n{n}_shift{sh}"])
    for seed in (0, 1, 2):
        for B in (4, 64):
            for tag, kw in (("uniform", {}), ("logit", {"logits_normal": True}),
                            ("logit_shift", {"logits_normal": True, "shift": 4.63}), ("xpred", {"prediction_type": "x"})):
                torch.manual_seed(seed)
                assert np.array_equal(t2n(Flow(n_steps=10, **kw).draw_timesteps(B)), g[f"draw_flow_{tag}_s{seed}_b{B}"])
            torch.manual_seed(seed)
            got = GaussianDiffusion(n_steps=1000).draw_timesteps(B)
            assert got.dtype == torch.int32 and np.array_equal(t2n(got), g[f"draw_ddpm_s{seed}_b{B}"])
    with pytest.raises(NotImplementedError):
        Flow(n_steps=4).set_steps(4, schedule="cosine")
    with pytest.raises(AssertionError):
        Flow(n_steps=4, prediction_type="eps")


def test_gaussian_tables_and_respacing_match_reference(golden):
    g = golden("schedules")
    for sched in ("linear", "cosine"):
        d = GaussianDiffusion(n_steps=1000, schedule=sched)
        for nm in ("betas", "alphas_bar", "sqrt_alphas_bar"):
            assert np.array_equal(t2n(getattr(d, nm)), g[f"gd_{sched}_{nm}"]), nm
        for nm in ("alphas_bar_prev", "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1",
                   "posterior_mean_coef2"):
            assert np.array_equal(t2n(getattr(d.sampler, nm)), g[f"gd_{sched}_{nm}"]), nm
    for n in (50, 100, 250):
        d = GaussianDiffusion(n_steps=1000)
        d.set_steps(n)
        assert np.array_equal(np.array(d.timestep_map), g[f"respace_{n}_map"])
        assert np.array_equal(t2n(d.betas), g[f"respace_{n}_betas"])
        assert np.array_equal(t2n(d.sampler.posterior_variance), g[f"respace_{n}_postvar"])
    d = GaussianDiffusion(n_steps=1000)
    d.set_steps(30, section_counts="10,10,10")
    assert np.array_equal(np.array(d.timestep_map), g["respace_sections_map"])
    assert np.array_equal(np.array(sorted(space_timesteps(1000, 10))), g["space_1000_10"])
    with pytest.raises(ValueError):
        space_timesteps(1000, 10, ddim=True)
    with pytest.raises(ValueError):
        space_timesteps(10, "6,6")
    with pytest.raises(ValueError):
        GaussianDiffusion(n_steps=10, sampling_method="euler")
    with pytest.raises(NotImplementedError):
        GaussianDiffusion(n_steps=10, schedule="sigmoid")


def test_rope_tables_match_reference(golden):
    g = golden("prims")
    c, s = rope_grid_tables(16, 16, [32, 32], 10_000.0)
    assert np.array_equal(t2n(c), g["rope_cos_16x16"]) and np.array_equal(t2n(s), g["rope_sin_16x16"])
    c, s = rope_grid_tables(3, 5, [8, 24], 2000.0)
    assert np.array_equal(t2n(c), g["rope_cos_3x5"]) and np.array_equal(t2n(s), g["rope_sin_3x5"])


def test_module_contract_and_layout():
    kw = dict(simple_dit=True, input_channels=4, inner_dim=384, embedding_dim=384, num_heads=6, patch_size=2, depth=12,
              n_classes=1000, classifier_free=True)
    m = MMDiT(**kw)
    assert sum(p.numel() for p in m.parameters()) == 39_922_576  # SURVEY.md Appendix B (DiT-S/2 dims)
    lay = ParamLayout(m.dims)
    named = dict(m.named_parameters())
    assert set(named) == set(lay.entries)
    spans = sorted((off, off + int(np.prod(shape))) for off, shape in lay.entries.values())
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), "overlapping parameters in the arena"
    assert all(off % 4 == 0 for off, _ in lay.entries.values())
    # the adaLN matrices are one contiguous [L*6D+2D, E] block
    first = lay.entries["layers.0.modulation.lin.weight"][0]
    last_off, last_shape = lay.entries["last_layer.adaLN_modulation.1.weight"]
    assert last_off + int(np.prod(last_shape)) - first == lay.mod_rows * 384
    # reference init: zero adaLN, xavier elsewhere
    assert float(m.layers[3].modulation.lin.weight.detach().abs().sum()) == 0.0
    assert float(m.layers[3].attention.qkv.weight.detach().abs().sum()) > 0.0
    # deep copies (EMA) keep parameters, drop the engine
    m2 = copy.deepcopy(m)
    assert m2._engine is None and torch.equal(m2.conv_proj.weight, m.conv_proj.weight)
    # unsupported reference modes fail loudly, never silently fall back
    with pytest.raises(AssertionError):  # the reference's own assertion (mmdit.py:642): joint blocks need a context embedder
        MMDiT(simple_dit=False)
    from diffulab_amd.networks.embedders import PrecomputedEmbedder

    emb = PrecomputedEmbedder(torch.zeros(1, 16, 96), null_embedding_seq_len=3)
    assert emb.n_output == 1 and emb.output_size == (96,) and int(emb.null_embedding_mask.sum()) == 3
    # the per-(device, dtype) copy of the null embedding follows IN-PLACE updates of the attribute (the reference converts per call)
    ctx = {"embeddings": torch.ones(2, 16, 96), "attn_mask": torch.ones(2, 16, dtype=torch.bool)}
    assert float(emb.drop_conditions(ctx, 1.0)["embeddings"].abs().sum()) == 0.0
    emb.null_embedding.add_(2.0)
    assert bool((emb.drop_conditions(ctx, 1.0)["embeddings"] == 2.0).all())
    emb.null_embedding_mask[:] = True
    assert bool(emb.drop_conditions(ctx, 1.0)["attn_mask"].all())
    emb.null_embedding.zero_()
    jkw = dict(simple_dit=False, context_embedder=emb, input_channels=4, inner_dim=128, embedding_dim=64, num_heads=2, patch_size=2,
               depth=2, rope_axes_dim=[16, 24, 24])
    mj = MMDiT(**jkw)
    assert mj.context_embed.weight.shape == (128, 96) and hasattr(mj.layers[0], "modulation_context")
    ms = MMDiT(**{**jkw, "n_single_stream_blocks": 1})  # the last block of the stack becomes a single-stream block
    assert hasattr(ms.layers[0], "modulation_context") and hasattr(ms.layers[1], "mlp") and ms.layers[1].modulation[1].out_features == 384
    with pytest.raises(NotImplementedError):  # the reference's default 3-axis split (64 // 3 = 21, odd) is not a valid RoPE width
        MMDiT(**{**jkw, "rope_axes_dim": None})
    with pytest.raises(NotImplementedError):
        DiTDims(inner_dim=384, num_heads=4).validate()
    with pytest.raises(RuntimeError):
        m(x=torch.zeros(1, 4, 32, 32), timesteps=torch.zeros(1), y=torch.zeros(1, dtype=torch.long))  # CPU: no fallback


def test_bench_launches_its_own_ranks_without_touching_a_gpu():
    """`python bench.py --gpus N` (no external launcher): the parent spawns N children with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set on 127.0.0.1, never initialises the HIP runtime itself and returns the children's status."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--launch-check"], capture_output=True,
                         text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = sorted((json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")), key=lambda r: r["rank"])
    assert [r["rank"] for r in rows] == [0, 1, 2] and [r["local_rank"] for r in rows] == [0, 1, 2]
    assert all(r["world"] == 3 and r["master"] == "127.0.0.1" and not r["cuda_initialized"] for r in rows)
    assert len({r["port"] for r in rows}) == 1
    # a failing rank is the launcher's exit status
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check", "--steps", "x"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert bad.returncode != 0


def test_fused_adamw_flat_arena_detection_is_cached_and_revalidated():
    """FusedAdamW._flat (round 5): the arena scan is remembered per group and a hit is re-validated by pointer compares -- a gradient
    that was detached, re-pointed or newly attached is seen at the next call (host logic only: no kernel runs here)"""
    from diffulab_amd.training.optim import FusedAdamW

    flat, gflat = torch.zeros(64), torch.zeros(64)
    ps = []
    for i in range(4):
        p = torch.nn.Parameter(torch.empty(0))
        p.data = flat[16 * i : 16 * (i + 1)]
        p.grad = gflat[16 * i : 16 * (i + 1)]
        ps.append(p)
    extra = torch.nn.Parameter(torch.zeros(3))  # outside the arena (an auxiliary head in the same group), no gradient yet
    opt = FusedAdamW(ps + [extra], lr=1e-3)
    g = opt.param_groups[0]
    scans = []
    orig = opt._flat_scan
    opt._flat_scan = lambda group: (scans.append(1), orig(group))[1]
    pb, gb, rest = opt._flat(g)
    assert pb.data_ptr() == flat.data_ptr() and gb.data_ptr() == gflat.data_ptr() and rest == [extra] and len(scans) == 1
    assert opt._flat(g)[2] == [extra] and len(scans) == 1            # hit: no rescan
    extra.grad = torch.zeros(3)                                       # an outside tensor gains a gradient: rescan (it might be a view)
    assert opt._flat(g)[2] == [extra] and len(scans) == 2
    assert opt._flat(g) is not None and len(scans) == 2
    ps[2].grad = None                                                 # a detached gradient: the arena no longer holds the whole group
    r = opt._flat(g)
    assert len(scans) == 3 and (r is None or ps[2] in r[2])
    ps[2].grad = gflat[32:48]
    assert opt._flat(g)[2] == [extra] and len(scans) == 4
    ps[1].grad = torch.zeros(16)                                      # re-pointed outside the gradient arena
    assert opt._flat(g) is None and len(scans) == 5
