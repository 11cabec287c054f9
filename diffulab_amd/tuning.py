"""A/B switches of the engines: environment variables read when an engine builds a workspace or issues a step.

Every switch is listed here with its shipped default and what it selects; the engines read them only through `on()` / `integer()` /
`text()`, so an unknown name is a programming error, not a silent default.  None of them changes what is computed beyond f32 summation
order; they exist so that a measured decision can be re-measured on other hardware with one variable.  The C ABI
reads no environment (tests/test_abi.py): a switch becomes an argument of the call it affects.
"""

from __future__ import annotations

import os

SWITCHES: dict[str, tuple[str, str]] = {
    # name: (default, meaning)
    "DL_NATIVE_BLOCK": ("1", "DiT blocks issued by the C ABI's block drivers (dl_dit_block_fwd / _bwd) instead of one Python call per kernel"),
    "DL_ROW_GEMM": ("1", "LayerNorm-modulate forward / backward as epilogues of the row-complete 256x384 GEMMs (D = 384, 256 tokens per sample)"),
    "DL_ROW_GEMM_QK": ("0", "QK-norm + RoPE as the epilogue of the qkv GEMM (measured slower than the separate row kernel)"),
    "DL_QKN_ON_LOAD": ("1", "QK-RMSNorm statistics leave with the qkv GEMM (dl_gemm_nt_ssq) and norm + RoPE are applied as the attention "
                       "forward stages q and k (dl_attn_fwd_qkn): no qk_norm_rope_fwd pass (row-complete path, <= 256 tokens)"),
    "DL_MLP_RECOMPUTE": ("1", "SwiGLU backward recomputes the MLP-up pre-activations instead of storing them in the forward"),
    "DL_ATTN_V_IN_PLACE": ("1", "attention reads V / writes dV inside the token-major qkv rows (N <= 256)"),
    "DL_QK_INPLACE": ("1", "attention backward writes dQ / dK token-major and the QK-norm backward runs in place on the dqkv rows"),
    "DL_WGRAD_GROUP": ("1", "the four weight gradients of a block as one atomics-free launch (dl_gemm_tn_group)"),
    "DL_WGRAD_INLINE": ("0", "grouped weight gradients on the main stream instead of the side stream"),
    "DL_WGRAD_SERIAL": ("0", "per-problem weight gradients on the main stream (engines without the grouped form)"),
    "DL_SIDE_WGS": ("", "workgroup cap of the side-stream weight gradients (default: 256 grouped, 128 per-problem)"),
    "DL_SIDE_CU_MASK": ("", "CU mask of the side stream (dl_stream_create_masked): 'i4' = every 4th CU, 'b128' = the first 128"),
    "DL_SIDE_LOW_PRIORITY": ("0", "DiT engines: the side stream of the weight gradients is created with the device's lowest stream priority "
                             "(measured +-0: DiT-S/2 20.83 / 20.80 vs 20.73 / 20.77 ms; SPRINT joint 28.77 / 28.58 vs 28.79 / 28.62; DDT 38.93 / 38.97 vs "
                             "39.01 / 39.00; joint MMDiT 49.14 / 49.13 vs 49.24 / 49.31; REPA 11.71 / 11.68 vs 11.76 / 11.73)"),
    "DL_UNET_SIDE_LOW_PRIORITY": ("1", "UNet: the same (the dispatcher serves the main chain's workgroups first: 26.42 / 26.50 vs 26.66 / 26.64 ms)"),
    "DL_JOIN_LAST": ("1", "single GPU: the side stream is joined after the conditioning backward instead of before it"),
    "DL_DP_RESERVE_CUS": ("0", "data parallel: CUs left to the communication library's workgroups (persistent main-chain grids shrink by it; "
                          "measured a loss at the headline shape, whose tile counts are whole multiples of 256: DESIGN.md section 5)"),
    "DL_MAIN_WGS": ("", "workgroup budget of the persistent main-chain kernels (experiments; default: all CUs, or CUs - reserve with a reducer)"),
    "DL_DP_EARLY_MOD": ("1", "data parallel: each block's adaLN rows are reduced as the block finishes"),
    "DL_HIPGRAPH": ("1", "samplers replay the denoiser forward as a captured hipGraph"),
    "DL_CFG_PAIR": ("1", "guided sampler steps run the conditional and the label-dropped forward as ONE forward over [x ; x] "
                    "(class-conditional MMDiT / DDT / UNetModel: same values per row, the weights stream once)"),
    "DL_LAUNCH_PLAN": ("0", "UNet (bf16 regime): a training forward / backward is recorded on its second run for a shape and re-issued as a "
                       "flat list of C calls afterwards (ops.LaunchPlan: the engine's Python costs the host ~24 us per launch, 716 "
                       "launches per step).  Measured (profiles/r06_e_*): the step at B = 64 is NOT host-bound -- its main queue is busy "
                       "19.2 of 20.3 ms and `host_issue` is the full launch queue pushing back -- so the plan changes nothing there "
                       "(20.03 vs 20.11 ms); it pays where the host is the slower side (under a profiler's per-launch overhead: 20.3 vs "
                       "22.7 ms), hence opt-in"),
    "DL_UNET_SIDE": ("1", "UNet weight gradients on a side stream"),
    "DL_UNET_NT_PAIR": ("1", "UNet AttentionBlock: the q and kv projections (and their data gradients) as one launch of the 128 x 128 GEMM "
                        "kernel where both are small (dl_gemm_nt_pair: 64 + 128 tiles on 256 CUs at the 4 x 4 level); bit-identical"),
    "DL_UNET_SPLITK": ("1", "split-K convolutions at the UNet's low-resolution levels (partial images + fixed-order fold: -7 % per step)"),
    "DL_UNET_WGRAD_WGS": ("128", "workgroup cap of the UNet's side-stream convolution weight gradients (0 = one per CU).  Round 6, with the "
                          "tap-reusing kernel and partial images (every workgroup stores its 295 KB of accumulators once, so the cap also "
                          "halves the stage bytes), B = 128: 0 -> 25.4 ms/step, 160 -> 24.1, 128 -> 23.9, 96 -> 24.4, 64 -> 25.0, 48 -> 26.8; "
                          "B = 64: 0 -> 19.0, 128 -> 18.0 (profiles/r06_q_*).  Round 5 (implicit-GEMM kernels, atomics): 0 was best"),
    "DL_UNET_FOLD_BATCHED": ("1", "UNet convolution weight gradients stay in a persistent transposed staging arena during the backward; one "
                             "launch at its end folds all of them into the [Co, Ci, 3, 3] gradients (instead of a zero-fill and a fold per convolution)"),
    "DL_UNET_WGRAD_PARTS": ("1", "UNet convolution weight gradients as partial images per pixel range (plain stores + fixed-order fold in the "
                            "batched fold launch: bit-reproducible) instead of f32 atomics into one image.  With the tap-reusing kernel of "
                            "round 6 (conv_wgrad.hip) the atomics were 56 % of a launch (1.45 TB/s of read-modify-writes against 6 TB/s of "
                            "stores) and the fold takes one tap of a channel tile per workgroup when eight or more images wait: B = 128 "
                            "24.7 -> 23.9 ms/step, B = 64 19.0 -> 18.0 (same workgroup cap).  With the implicit-GEMM kernels alone it was "
                            "slower (28.5 vs 26.8 ms: session e)"),
    "DL_UNET_DET_COLSUM": ("0", "UNet bias gradients through the bit-reproducible column sum (measured 2 % slower)"),
}


def text(name: str) -> str:
    return os.environ.get(name, SWITCHES[name][0])


def on(name: str) -> bool:
    return text(name) not in ("", "0")


def integer(name: str, default: int) -> int:
    v = text(name)
    return int(v) if v else default
