"""fp32-class launch sequences of the DiT hot path: the reference's DEFAULT precision (`precision_type="no"`,
training/trainers/common.py:76,105; configs/trainer/default.yaml:4).

Same parameter arena, same layout and the same engine interface as `engine.DiTEngine` (bind / forward / backward / feature), but every
activation is f32 and every product runs on the exact-f32 matrix instruction through `dl_f32_gemm` straight on the f32 parameters
(no bf16 weight shadows).  Attention materialises the probabilities [B, H, N, N] (kept per block for the backward) and uses the same
strided GEMM entry point with (sample, head) batch strides inside the token-major qkv rows.  Every per-sample sum has a single
producer, so a step is bit-reproducible.  The regime exists for parity with the reference's fp32 path (north-star: loss curve to
1e-4; SURVEY 8(c): fp32-mode kernels <= 1e-5), not for throughput: at the headline shape it runs at the f32 MFMA rate (1/16 of bf16).

Reference sites restated by the sequences below: mmdit.py:853-928 (simple_dit_forward / forward), mmdit.py:288-309 (DiTBlock),
mmdit.py:75-104 (DiTAttention), mmdit.py:542-549 (ModulatedLastLayer), nn.py:91-164 (embeddings), nn.py:427-486 (QKNorm, SwiGLU).
"""

from __future__ import annotations

import torch
from torch import Tensor

from . import ops
from .engine import DiTDims, ParamLayout, rope_grid_tables


class DiTEngineF32:
    precision = "fp32"

    def __init__(self, dims: DiTDims, device: torch.device | str = "cuda") -> None:
        D = dims.inner_dim
        if D % 4 or D > 1024 or dims.head_dim % 4 or sum(dims.rope_axes_dim) % 4 or len(dims.rope_axes_dim) != 2:
            raise NotImplementedError("fp32 DiT path: inner_dim % 4 == 0 up to 1024, head_dim and rotary width multiples of 4")
        self.d = dims
        self.dev = torch.device(device)
        self.layout = ParamLayout(dims)
        self.prefixes = self.layout.prefixes
        self.params: Tensor | None = None
        self.grads: Tensor | None = None
        self.param_version = 0
        self.reducer = None
        self._ws_key: tuple | None = None
        self._ws_cache: dict[tuple, tuple] = {}
        self._rope: dict[tuple[int, int], tuple[Tensor, Tensor]] = {}
        ent = self.layout.entries
        starts = [ent[n][0] for n in self.layout.block_first] + [self.layout.size]
        self.layer_ranges = [(starts[i], starts[i + 1]) for i in range(len(self.prefixes))]
        self._train = False

    # ------------------------------------------------------------------ parameters (f32 arena: the GEMMs read it directly)
    def bind(self, params: Tensor, grads: Tensor | None) -> None:
        assert params.dtype == torch.float32 and params.numel() == self.layout.size and params.is_cuda
        self.params, self.grads = params, grads
        self._pviews: dict[str, Tensor] = {}
        self._gviews: dict[str, Tensor] = {}

    def P(self, name: str) -> Tensor:
        v = self._pviews.get(name)
        if v is None:
            v = self._pviews[name] = self.layout.view(self.params, name)
        return v

    def G(self, name: str) -> Tensor:
        v = self._gviews.get(name)
        if v is None:
            v = self._gviews[name] = self.layout.view(self.grads, name)
        return v

    def W(self, name: str) -> Tensor:
        """a weight as the 2-D [out, in] matrix the GEMMs take"""
        v = self.P(name)
        return v if v.dim() == 2 else v.view(v.shape[0], -1)

    def GW(self, name: str) -> Tensor:
        v = self.G(name)
        return v if v.dim() == 2 else v.view(v.shape[0], -1)

    def refresh_shadows(self, force: bool = False) -> None:  # (interface of the bf16 engine: there is nothing to refresh here)
        pass

    def params_changed(self) -> None:
        pass

    def _mod_matrix(self, flat: Tensor) -> tuple[Tensor, Tensor]:
        lay, E = self.layout, self.d.embedding_dim
        R = lay.mod_rows
        w0, b0 = lay.entries[lay.mod_w0][0], lay.entries[lay.mod_b0][0]
        return flat[w0 : w0 + R * E].view(R, E), flat[b0 : b0 + R]

    # ------------------------------------------------------------------ workspace
    def _alloc(self, B: int, H: int, W: int, train: bool) -> None:
        key = (B, H, W, train)
        if key == self._ws_key:
            return
        if key in self._ws_cache:
            self.ws, self.geo = self._ws_cache[key]
            self._ws_key = key
            return
        d, dev = self.d, self.dev
        D, E, p, L = d.inner_dim, d.embedding_dim, d.patch_size, d.depth
        gh, gw = H // p, W // p
        N = gh * gw
        M = B * N
        if N % 4 or N > 4096:
            raise NotImplementedError(f"fp32 DiT path: tokens per sample must be a multiple of 4 up to 4096 (got {N})")
        Hh, F = d.num_heads, d.mlp_ratio * D
        Fi, Fo = d.input_channels * p * p, d.output_channels * p * p
        R = self.layout.mod_rows

        def z(*shape):
            with torch.inference_mode(False):
                return torch.zeros(*shape, device=dev, dtype=torch.float32)

        w: dict[str, object] = {}
        w["tokP"] = z(M, Fi)
        w["temb"], w["pre1"], w["h1"] = z(B, d.frequency_embedding), z(B, E), z(B, E)
        w["e"], w["emb"], w["se"] = z(B, E), z(B, E), z(B, E)
        w["mod"] = z(B, R)
        nl = L if train else 1
        w["x"] = [z(M, D) for _ in range((L + 1) if train else 2)]
        w["layers"] = [{
            "mean1": z(M), "rstd1": z(M), "xm1": z(M, D), "qkv": z(M, 3 * D), "qk": z(M, 2 * D), "rrms": z(M, 2),
            "P": z(B, Hh, N, N), "a": z(M, D), "t1": z(M, D), "x1": z(M, D), "mean2": z(M), "rstd2": z(M), "xm2": z(M, D),
            "u": z(M, 2 * F), "h": z(M, F), "t2": z(M, D),
        } for _ in range(nl)]
        w["meanf"], w["rstdf"], w["xf"] = z(M), z(M), z(M, D)
        w["otok"] = z(M, Fo)
        w["pred"] = z(B, d.output_channels, H, W)
        if train:
            w["dO"] = z(M, Fo)
            w["dxa"], w["dxb"], w["dxm"], w["da"] = z(M, D), z(M, D), z(M, D), z(M, D)
            w["dt1"], w["dt2"] = z(M, D), z(M, D)
            w["dh"], w["du"] = z(M, F), z(M, 2 * F)
            w["dP"] = z(B, Hh, N, N)
            w["dqk"], w["dqkv"] = z(M, 2 * D), z(M, 3 * D)
            w["dmod"] = z(B, R)
            w["dwb"] = z(B, 2, D)
            w["dqs"] = z(B, 2, D)
            w["dse"], w["demb"], w["dh1"], w["dpre1"] = z(B, E), z(B, E), z(B, E), z(B, E)
            # split-K scratch of the weight gradients over all tokens (and of the conditioning path's long contraction): up to 64
            # partial images of the largest weight (dl_f32_gemm folds them in a fixed order)
            big = max(2 * F * D, B * E, B * R // 8, 1 << 18)
            w["scr"] = torch.empty(min(64, max(2, M // 256)) * big, device=dev, dtype=torch.float32)
        self.ws, self._ws_key = w, key
        self.geo = (B, H, W, gh, gw, N, M, Fi, Fo)
        if len(self._ws_cache) >= 4:
            self._ws_cache.pop(next(iter(self._ws_cache)))
        self._ws_cache[key] = (w, self.geo)
        if (gh, gw) not in self._rope:
            c, s = rope_grid_tables(gh, gw, d.rope_axes_dim, d.rope_base)
            self._rope[(gh, gw)] = (c.to(dev), s.to(dev))

    # ------------------------------------------------------------------ forward
    def forward(self, x: Tensor, t: Tensor, y_eff: Tensor | None, train: bool = True, refresh: bool = True) -> Tensor:
        """x f32 [B,C,H,W]; t f32 [B]; y_eff int64 [B] labels after the classifier-free drop, or None -> pred f32 [B,Co,H,W]
        (a workspace buffer: consume it before the next forward)"""
        d = self.d
        B, C, H, W = x.shape
        assert C == d.input_channels and x.dtype == torch.float32 and x.is_cuda
        self._alloc(B, H, W, train)
        w = self.ws
        _, _, _, gh, gw, N, M, Fi, Fo = self.geo
        D, E, L, Hh, dh = d.inner_dim, d.embedding_dim, d.depth, d.num_heads, d.head_dim
        F = d.mlp_ratio * D
        cos, sin = self._rope[(gh, gw)]
        rot = sum(d.rope_axes_dim)
        self._train, self._yeff = train, y_eff
        P, Wt = self.P, self.W

        # stem + conditioning (mmdit.py:757-765, 866-868; nn.py:106-114, 530-531)
        ops.f32_patchify(x, w["tokP"], d.patch_size, ops.PATCH_CPP)
        xs = w["x"]
        ops.f32_linear(w["tokP"], Wt("conv_proj.weight"), xs[0])
        ops.f32_timestep_embedding(t, w["temb"])
        ops.f32_linear(w["temb"], Wt("time_embed.0.weight"), w["h1"], bias=P("time_embed.0.bias"), act=ops.ACT_SILU, pre_out=w["pre1"])
        ops.f32_linear(w["h1"], Wt("time_embed.2.weight"), w["e"], bias=P("time_embed.2.bias"))
        table = P("label_embed.embedding.weight") if d.n_classes is not None else None
        ops.f32_cond_combine_fwd(w["e"], table, y_eff if table is not None else None, w["emb"], w["se"])
        mod_w, mod_b = self._mod_matrix(self.params)
        mod = w["mod"]
        ops.f32_linear(w["se"], mod_w, mod, bias=mod_b)

        scale = dh**-0.5
        pend = None  # (x_base, t, gate) of the sub-layer whose gated residual is applied by the next LayerNorm kernel
        for i in range(L):
            a = w["layers"][i if train else 0]
            xin = xs[i] if train else xs[i & 1]
            pre, mo = f"layers.{i}.", i * 6 * D
            if pend is None:
                ops.f32_ln_modulate_fwd(xin, P(pre + "norm_1.weight"), P(pre + "norm_1.bias"), mod[:, mo : mo + D],
                                        mod[:, mo + D : mo + 2 * D], N, 1e-5, a["xm1"], a["mean1"], a["rstd1"])
            else:
                ops.f32_ln_modulate_fwd(pend[0], P(pre + "norm_1.weight"), P(pre + "norm_1.bias"), mod[:, mo : mo + D],
                                        mod[:, mo + D : mo + 2 * D], N, 1e-5, a["xm1"], a["mean1"], a["rstd1"], t=pend[1],
                                        gate=pend[2], x_out=xin)
            ops.f32_linear(a["xm1"], Wt(pre + "attention.qkv.weight"), a["qkv"])
            ops.f32_qk_norm_rope_fwd(a["qkv"], P(pre + "attention.qk_norm.query_norm.scale"),
                                     P(pre + "attention.qk_norm.key_norm.scale"), cos, sin, a["qk"], a["rrms"], B, N, Hh, dh, rot)
            # S = scale q k^T per (sample, head): heads are 64-wide column blocks of the token-major rows
            ops.f32_gemm(a["qk"], a["qk"], a["P"], N, N, dh, lda=2 * D, ldb=2 * D, ldc=N, b_off=D, batch=(B, Hh),
                         sa=(N * 2 * D, dh), sb=(N * 2 * D, dh), sc=(Hh * N * N, N * N), alpha=scale)
            ops.f32_softmax_fwd(a["P"], B * Hh * N, N)
            # O = P V, V read in place from the v third of qkv, O written as 'b h n d -> b n (h d)' (mmdit.py:100)
            ops.f32_gemm(a["P"], a["qkv"], a["a"], N, dh, N, lda=N, ldb=3 * D, ldc=D, tb=True, b_off=2 * D, batch=(B, Hh),
                         sa=(Hh * N * N, N * N), sb=(N * 3 * D, dh), sc=(N * D, dh))
            ops.f32_linear(a["a"], Wt(pre + "attention.proj_out.weight"), a["t1"])
            ops.f32_ln_modulate_fwd(xin, P(pre + "norm_2.weight"), P(pre + "norm_2.bias"), mod[:, mo + 3 * D : mo + 4 * D],
                                    mod[:, mo + 4 * D : mo + 5 * D], N, 1e-5, a["xm2"], a["mean2"], a["rstd2"], t=a["t1"],
                                    gate=mod[:, mo + 2 * D : mo + 3 * D], x_out=a["x1"])
            ops.f32_linear(a["xm2"], Wt(pre + "mlp_input.0.weight"), a["u"])
            ops.f32_swiglu_fwd(a["u"], a["h"])
            ops.f32_linear(a["h"], Wt(pre + "mlp_input.2.weight"), a["t2"])
            pend = (a["x1"], a["t2"], mod[:, mo + 5 * D : mo + 6 * D])

        xl = xs[L] if train else xs[L & 1]
        mo = L * 6 * D
        ops.f32_ln_modulate_fwd(pend[0], None, None, mod[:, mo : mo + D], mod[:, mo + D : mo + 2 * D], N, 1e-6, w["xf"], w["meanf"],
                                w["rstdf"], t=pend[1], gate=pend[2], x_out=xl)
        ops.f32_linear(w["xf"], Wt("last_layer.linear.weight"), w["otok"], bias=P("last_layer.linear.bias"))
        ops.unpatchify(w["otok"], w["pred"], d.patch_size)
        return w["pred"]

    # ------------------------------------------------------------------ backward
    def feature(self, k: int) -> Tensor:
        """output of block k of the last train-mode forward (f32 [B, N, D]): what a forward hook on ``layers[k]`` sees"""
        assert self._train, "block outputs are only kept by the train-mode launch sequence"
        B, _, _, _, _, N, _, _, _ = self.geo
        return self.ws["x"][k + 1].view(B, N, self.d.inner_dim)

    def backward(self, dpred: Tensor, dfeats: dict[int, Tensor] | None = None) -> None:
        """accumulates d(loss)/d(param) into the flat gradient arena (+=) for the last train-mode forward"""
        assert self._train and self.grads is not None
        d, w = self.d, self.ws
        B, H, W, gh, gw, N, M, Fi, Fo = self.geo
        D, E, L, Hh, dh = d.inner_dim, d.embedding_dim, d.depth, d.num_heads, d.head_dim
        F = d.mlp_ratio * D
        dfeats = {k: g.reshape(M, D).float().contiguous() for k, g in (dfeats or {}).items()}
        cos, sin = self._rope[(gh, gw)]
        rot = sum(d.rope_axes_dim)
        P, Wt, G, GW = self.P, self.W, self.G, self.GW
        mod, dmod, xs, scr = w["mod"], w["dmod"], w["x"], w["scr"]
        scale = dh**-0.5

        # head: last linear (mmdit.py:548) + final adaLN (mmdit.py:543-547)
        ops.f32_patchify(dpred, w["dO"], d.patch_size, ops.PATCH_PPC)
        ops.f32_linear_wgrad(w["dO"], w["xf"], GW("last_layer.linear.weight"), scratch=scr)
        ops.colsum(w["dO"], G("last_layer.linear.bias"), M, Fo, scratch=scr)
        ops.f32_linear_dgrad(w["dO"], Wt("last_layer.linear.weight"), w["dxm"])
        mo, ml = L * 6 * D, (L - 1) * 6 * D
        dx, dx_alt = w["dxa"], w["dxb"]
        ops.f32_ln_modulate_bwd(w["dxm"], xs[L], None, None, mod[:, mo : mo + D], N, w["meanf"], w["rstdf"], dfeats.get(L - 1), dx,
                                dmod[:, mo : mo + D], dmod[:, mo + D : mo + 2 * D], None, gate_t=w["layers"][L - 1]["t2"],
                                gate=mod[:, ml + 5 * D : ml + 6 * D], dt=w["dt2"], dgate=dmod[:, ml + 5 * D : ml + 6 * D])

        for i in reversed(range(L)):
            a = w["layers"][i]
            pre, mo = f"layers.{i}.", i * 6 * D
            # MLP branch (mmdit.py:260-264, 305-308)
            ops.f32_linear_wgrad(w["dt2"], a["h"], GW(pre + "mlp_input.2.weight"), scratch=scr)
            ops.f32_linear_dgrad(w["dt2"], Wt(pre + "mlp_input.2.weight"), w["dh"])
            ops.f32_swiglu_bwd(w["dh"], a["u"], w["du"])
            ops.f32_linear_wgrad(w["du"], a["xm2"], GW(pre + "mlp_input.0.weight"), scratch=scr)
            ops.f32_linear_dgrad(w["du"], Wt(pre + "mlp_input.0.weight"), w["dxm"])
            ops.f32_ln_modulate_bwd(w["dxm"], a["x1"], P(pre + "norm_2.weight"), P(pre + "norm_2.bias"), mod[:, mo + 3 * D : mo + 4 * D],
                                    N, a["mean2"], a["rstd2"], dx, dx_alt, dmod[:, mo + 3 * D : mo + 4 * D],
                                    dmod[:, mo + 4 * D : mo + 5 * D], w["dwb"], gate_t=a["t1"], gate=mod[:, mo + 2 * D : mo + 3 * D],
                                    dt=w["dt1"], dgate=dmod[:, mo + 2 * D : mo + 3 * D])
            ops.reduce_rows_batched_f32(w["dwb"], 0, G(pre + "norm_2.weight"), 0, 1, B, 2 * D)  # [w; b] adjacent; fixed-order fold
            dx, dx_alt = dx_alt, dx
            # attention branch (mmdit.py:75-104)
            ops.f32_linear_wgrad(w["dt1"], a["a"], GW(pre + "attention.proj_out.weight"), scratch=scr)
            ops.f32_linear_dgrad(w["dt1"], Wt(pre + "attention.proj_out.weight"), w["da"])
            hb = dict(batch=(B, Hh))
            # dP = dO V^T
            ops.f32_gemm(w["da"], a["qkv"], w["dP"], N, N, dh, lda=D, ldb=3 * D, ldc=N, b_off=2 * D, sa=(N * D, dh),
                         sb=(N * 3 * D, dh), sc=(Hh * N * N, N * N), **hb)
            # dV = P^T dO -> the v third of dqkv
            ops.f32_gemm(a["P"], w["da"], w["dqkv"], N, dh, N, lda=N, ldb=D, ldc=3 * D, ta=True, tb=True, c_off=2 * D,
                         sa=(Hh * N * N, N * N), sb=(N * D, dh), sc=(N * 3 * D, dh), **hb)
            ops.f32_softmax_bwd(a["P"], w["dP"], B * Hh * N, N)  # dS over dP
            # dQ = scale dS K ; dK = scale dS^T Q (gradients of the normalised + rotated q, k, token-major)
            ops.f32_gemm(w["dP"], a["qk"], w["dqk"], N, dh, N, lda=N, ldb=2 * D, ldc=2 * D, tb=True, b_off=D, sa=(Hh * N * N, N * N),
                         sb=(N * 2 * D, dh), sc=(N * 2 * D, dh), alpha=scale, **hb)
            ops.f32_gemm(w["dP"], a["qk"], w["dqk"], N, dh, N, lda=N, ldb=2 * D, ldc=2 * D, ta=True, tb=True, c_off=D,
                         sa=(Hh * N * N, N * N), sb=(N * 2 * D, dh), sc=(N * 2 * D, dh), alpha=scale, **hb)
            ops.f32_qk_norm_rope_bwd(w["dqk"], a["qkv"], P(pre + "attention.qk_norm.query_norm.scale"),
                                     P(pre + "attention.qk_norm.key_norm.scale"), cos, sin, a["rrms"], w["dqkv"], w["dqs"], B, N, Hh, dh,
                                     rot)
            ops.reduce_rows_batched_f32(w["dqs"], 0, G(pre + "attention.qk_norm.query_norm.scale"), 0, 1, B, 2 * D)  # [q; k] adjacent
            ops.f32_linear_wgrad(w["dqkv"], a["xm1"], GW(pre + "attention.qkv.weight"), scratch=scr)
            ops.f32_linear_dgrad(w["dqkv"], Wt(pre + "attention.qkv.weight"), w["dxm"])
            if i - 1 in dfeats:  # auxiliary-loss gradient on the output of block i-1 (= this block's input)
                ops.f32_add(dx, dfeats[i - 1], dx)
            nxt = {}
            if i > 0:
                mp = (i - 1) * 6 * D
                nxt = dict(gate_t=w["layers"][i - 1]["t2"], gate=mod[:, mp + 5 * D : mp + 6 * D], dt=w["dt2"],
                           dgate=dmod[:, mp + 5 * D : mp + 6 * D])
            ops.f32_ln_modulate_bwd(w["dxm"], xs[i], P(pre + "norm_1.weight"), P(pre + "norm_1.bias"), mod[:, mo : mo + D], N,
                                    a["mean1"], a["rstd1"], dx, dx_alt, dmod[:, mo : mo + D], dmod[:, mo + D : mo + 2 * D], w["dwb"],
                                    **nxt)
            ops.reduce_rows_batched_f32(w["dwb"], 0, G(pre + "norm_1.weight"), 0, 1, B, 2 * D)
            dx, dx_alt = dx_alt, dx

        # stem (no gradient flows to the input latents) and conditioning path
        ops.f32_linear_wgrad(dx, w["tokP"], GW("conv_proj.weight"), scratch=scr)
        g_modw, g_modb = self._mod_matrix(self.grads)
        mod_w, _ = self._mod_matrix(self.params)
        ops.f32_linear_wgrad(dmod, w["se"], g_modw)
        ops.colsum(dmod, g_modb, B, self.layout.mod_rows, scratch=scr)
        ops.f32_linear_dgrad(dmod, mod_w, w["dse"], scratch=scr)
        table = d.n_classes is not None
        ops.f32_cond_combine_bwd(w["dse"], w["emb"], self._yeff if table else None, w["demb"],
                                 G("label_embed.embedding.weight") if table else None)
        ops.colsum(w["demb"], G("time_embed.2.bias"), B, E, scratch=scr)
        ops.f32_linear_wgrad(w["demb"], w["h1"], GW("time_embed.2.weight"))
        ops.f32_linear_dgrad(w["demb"], Wt("time_embed.2.weight"), w["dh1"])
        ops.f32_silu_bwd(w["dh1"], w["pre1"], w["dpre1"])
        ops.f32_linear_wgrad(w["dpre1"], w["temb"], GW("time_embed.0.weight"))
        ops.colsum(w["dpre1"], G("time_embed.0.bias"), B, E, scratch=scr)
        if self.reducer is not None:  # data parallel: one exchange over the whole arena once the backward has ended
            self.reducer.ready(0, self.layout.size)
            self.reducer.finish()
