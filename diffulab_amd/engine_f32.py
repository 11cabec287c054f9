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
    _conv_name = "conv_proj.weight"  # the patch embedding behind the block stack (DDT: its encoder's)

    def __init__(self, dims: DiTDims, device: torch.device | str = "cuda") -> None:
        D = dims.inner_dim
        if D % 4 or D > 1024 or dims.head_dim % 4 or sum(dims.rope_axes_dim) % 4 or len(dims.rope_axes_dim) != 2:
            raise NotImplementedError("fp32 DiT path: inner_dim % 4 == 0 up to 1024, head_dim and rotary width multiples of 4")
        self.d = dims
        self.dev = torch.device(device)
        self.layout = self._make_layout(dims)
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

    def _make_layout(self, d: DiTDims) -> ParamLayout:
        return ParamLayout(d)

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
    def _z(self, *shape) -> Tensor:
        with torch.inference_mode(False):
            return torch.zeros(*shape, device=self.dev, dtype=torch.float32)

    def _block_buffers(self, B: int, nt: int) -> dict:
        """what one DiT block keeps of its forward over B * nt tokens (everything its backward reads)"""
        d, z = self.d, self._z
        D, Hh, F, mt = d.inner_dim, d.num_heads, d.mlp_ratio * d.inner_dim, B * nt
        if nt % 4 or nt > 4096:
            raise NotImplementedError(f"fp32 DiT path: tokens per sample must be a multiple of 4 up to 4096 (got {nt})")
        return {"mean1": z(mt), "rstd1": z(mt), "xm1": z(mt, D), "qkv": z(mt, 3 * D), "qk": z(mt, 2 * D), "rrms": z(mt, 2),
                "P": z(B, Hh, nt, nt), "a": z(mt, D), "t1": z(mt, D), "x1": z(mt, D), "mean2": z(mt), "rstd2": z(mt), "xm2": z(mt, D),
                "u": z(mt, 2 * F), "h": z(mt, F), "t2": z(mt, D)}

    def _chain_buffers(self, B: int, nt: int) -> dict:
        """scratch of the backward chain over B * nt tokens (shared by every block with that token count)"""
        d, z = self.d, self._z
        D, Hh, F, mt = d.inner_dim, d.num_heads, d.mlp_ratio * d.inner_dim, B * nt
        return {"dxa": z(mt, D), "dxb": z(mt, D), "dxm": z(mt, D), "da": z(mt, D), "dt1": z(mt, D), "dt2": z(mt, D), "dh": z(mt, F),
                "du": z(mt, 2 * F), "dP": z(B, Hh, nt, nt), "dqk": z(mt, 2 * D), "dqkv": z(mt, 3 * D)}

    def _common_buffers(self, w: dict, B: int, M: int, train: bool) -> None:
        """stem, conditioning path, head"""
        d, z = self.d, self._z
        D, E, p = d.inner_dim, d.embedding_dim, d.patch_size
        Fi, Fo = d.input_channels * p * p, d.output_channels * p * p
        R = self.layout.mod_rows
        w["tokP"] = z(M, Fi)
        w["temb"], w["pre1"], w["h1"] = z(B, d.frequency_embedding), z(B, E), z(B, E)
        w["e"], w["emb"], w["se"] = z(B, E), z(B, E), z(B, E)
        w["mod"] = z(B, R)
        w["meanf"], w["rstdf"], w["xf"] = z(M), z(M), z(M, D)
        w["otok"] = z(M, Fo)
        if train:
            w["dO"] = z(M, Fo)
            w["dmod"] = z(B, R)
            w["dwb"] = z(B, 2, D)
            w["dqs"] = z(B, 2, D)
            w["dse"], w["demb"], w["dh1"], w["dpre1"] = z(B, E), z(B, E), z(B, E), z(B, E)
            # split-K scratch of the weight gradients over all tokens (and of the conditioning path's long contraction): up to 64
            # partial images of the largest weight (dl_f32_gemm folds them in a fixed order)
            F = d.mlp_ratio * D
            big = max(2 * F * D, B * E, B * R // 8, 1 << 18)
            w["scr"] = torch.empty(min(64, max(2, M // 256)) * big, device=self.dev, dtype=torch.float32)

    def _alloc(self, B: int, H: int, W: int, train: bool) -> None:
        key = (B, H, W, train)
        if key == self._ws_key:
            return
        if key in self._ws_cache:
            self.ws, self.geo = self._ws_cache[key]
            self._ws_key = key
            return
        d = self.d
        D, p, L = d.inner_dim, d.patch_size, d.depth
        gh, gw = H // p, W // p
        N = gh * gw
        M = B * N
        z = self._z
        w: dict[str, object] = {}
        self._common_buffers(w, B, M, train)
        w["x"] = [z(M, D) for _ in range((L + 1) if train else 2)]
        w["layers"] = [self._block_buffers(B, N) for _ in range(L if train else 1)]
        w["pred"] = z(B, d.output_channels, H, W)
        if train:
            w["s"] = self._chain_buffers(B, N)
        self._publish(w, key, (B, H, W, gh, gw, N, M, d.input_channels * p * p, d.output_channels * p * p))

    def _publish(self, w: dict, key: tuple, geo: tuple) -> None:
        self.ws, self._ws_key, self.geo = w, key, geo
        if len(self._ws_cache) >= 4:
            self._ws_cache.pop(next(iter(self._ws_cache)))
        self._ws_cache[key] = (w, geo)
        gh, gw = geo[3], geo[4]
        if (gh, gw) not in self._rope:
            c, s = rope_grid_tables(gh, gw, self.d.rope_axes_dim, self.d.rope_base)
            self._rope[(gh, gw)] = (c.to(self.dev), s.to(self.dev))

    # ------------------------------------------------------------------ forward pieces
    def _stem_cond_fwd(self, x: Tensor, t: Tensor, y_eff: Tensor | None, x0: Tensor) -> Tensor:
        """patch embedding -> x0; timestep / label conditioning -> the modulation rows of every block (mmdit.py:757-765, 866-868;
        nn.py:106-114, 530-531)"""
        d, w, P, Wt = self.d, self.ws, self.P, self.W
        ops.f32_patchify(x, w["tokP"], d.patch_size, ops.PATCH_CPP)
        ops.f32_linear(w["tokP"], Wt(self._conv_name), x0)
        ops.f32_timestep_embedding(t, w["temb"])
        ops.f32_linear(w["temb"], Wt("time_embed.0.weight"), w["h1"], bias=P("time_embed.0.bias"), act=ops.ACT_SILU, pre_out=w["pre1"])
        ops.f32_linear(w["h1"], Wt("time_embed.2.weight"), w["e"], bias=P("time_embed.2.bias"))
        table = P("label_embed.embedding.weight") if d.n_classes is not None else None
        ops.f32_cond_combine_fwd(w["e"], table, y_eff if table is not None else None, w["emb"], w["se"])
        mod_w, mod_b = self._mod_matrix(self.params)
        ops.f32_linear(w["se"], mod_w, w["mod"], bias=mod_b)
        return w["mod"]

    def _blk_fwd(self, a: dict, pre: str, mo: int, xin: Tensor, pend, B: int, nt: int, pos: Tensor | None = None,
                 mod: Tensor | None = None, rpm: int | None = None):
        """one DiTBlock (mmdit.py:288-309) over B * nt tokens.  pend = (x_base, t, gate) of the previous sub-layer whose gated
        residual this block's first LayerNorm kernel applies (and writes to xin), or None when xin already holds the block input;
        returns this block's own pending residual.  mod / rpm: the modulation matrix and the token rows that share one of its rows
        (default: the per-sample matrix ws["mod"], nt rows; DDT's decoder: one row per token)."""
        d, P, Wt = self.d, self.P, self.W
        D, Hh, dh = d.inner_dim, d.num_heads, d.head_dim
        gh, gw = self.geo[3], self.geo[4]
        cos, sin = self._rope[(gh, gw)]
        rot = sum(d.rope_axes_dim)
        mod = self.ws["mod"] if mod is None else mod
        rpm = nt if rpm is None else rpm
        if pend is None:
            ops.f32_ln_modulate_fwd(xin, P(pre + "norm_1.weight"), P(pre + "norm_1.bias"), mod[:, mo : mo + D],
                                    mod[:, mo + D : mo + 2 * D], rpm, 1e-5, a["xm1"], a["mean1"], a["rstd1"])
        else:
            ops.f32_ln_modulate_fwd(pend[0], P(pre + "norm_1.weight"), P(pre + "norm_1.bias"), mod[:, mo : mo + D],
                                    mod[:, mo + D : mo + 2 * D], rpm, 1e-5, a["xm1"], a["mean1"], a["rstd1"], t=pend[1],
                                    gate=pend[2], x_out=xin)
        ops.f32_linear(a["xm1"], Wt(pre + "attention.qkv.weight"), a["qkv"])
        ops.f32_qk_norm_rope_fwd(a["qkv"], P(pre + "attention.qk_norm.query_norm.scale"),
                                 P(pre + "attention.qk_norm.key_norm.scale"), cos, sin, a["qk"], a["rrms"], B, nt, Hh, dh, rot, pos=pos)
        # S = scale q k^T per (sample, head): heads are 64-wide column blocks of the token-major rows
        ops.f32_gemm(a["qk"], a["qk"], a["P"], nt, nt, dh, lda=2 * D, ldb=2 * D, ldc=nt, b_off=D, batch=(B, Hh),
                     sa=(nt * 2 * D, dh), sb=(nt * 2 * D, dh), sc=(Hh * nt * nt, nt * nt), alpha=dh**-0.5)
        ops.f32_softmax_fwd(a["P"], B * Hh * nt, nt)
        # O = P V, V read in place from the v third of qkv, O written as 'b h n d -> b n (h d)' (mmdit.py:100)
        ops.f32_gemm(a["P"], a["qkv"], a["a"], nt, dh, nt, lda=nt, ldb=3 * D, ldc=D, tb=True, b_off=2 * D, batch=(B, Hh),
                     sa=(Hh * nt * nt, nt * nt), sb=(nt * 3 * D, dh), sc=(nt * D, dh))
        ops.f32_linear(a["a"], Wt(pre + "attention.proj_out.weight"), a["t1"])
        ops.f32_ln_modulate_fwd(xin, P(pre + "norm_2.weight"), P(pre + "norm_2.bias"), mod[:, mo + 3 * D : mo + 4 * D],
                                mod[:, mo + 4 * D : mo + 5 * D], rpm, 1e-5, a["xm2"], a["mean2"], a["rstd2"], t=a["t1"],
                                gate=mod[:, mo + 2 * D : mo + 3 * D], x_out=a["x1"])
        ops.f32_linear(a["xm2"], Wt(pre + "mlp_input.0.weight"), a["u"])
        ops.f32_swiglu_fwd(a["u"], a["h"])
        ops.f32_linear(a["h"], Wt(pre + "mlp_input.2.weight"), a["t2"])
        return (a["x1"], a["t2"], mod[:, mo + 5 * D : mo + 6 * D])

    def _head_fwd(self, x: Tensor, pend, N: int, mod: Tensor | None = None, rpm: int | None = None, mo: int | None = None) -> Tensor:
        """ModulatedLastLayer (mmdit.py:542-549) + unpatchify; pend as in _blk_fwd (its residual is written to x); mod / rpm as in
        _blk_fwd, mo = first column of the layer's [scale | shift] rows"""
        d, w = self.d, self.ws
        D = d.inner_dim
        mo = d.depth * 6 * D if mo is None else mo
        mod = w["mod"] if mod is None else mod
        kw = {} if pend is None else dict(t=pend[1], gate=pend[2], x_out=x)
        ops.f32_ln_modulate_fwd(x if pend is None else pend[0], None, None, mod[:, mo : mo + D], mod[:, mo + D : mo + 2 * D],
                                N if rpm is None else rpm, 1e-6, w["xf"], w["meanf"], w["rstdf"], **kw)
        ops.f32_linear(w["xf"], self.W("last_layer.linear.weight"), w["otok"], bias=self.P("last_layer.linear.bias"))
        ops.unpatchify(w["otok"], w["pred"], d.patch_size)
        return w["pred"]

    def forward(self, x: Tensor, t: Tensor, y_eff: Tensor | None, train: bool = True, refresh: bool = True) -> Tensor:
        """x f32 [B,C,H,W]; t f32 [B]; y_eff int64 [B] labels after the classifier-free drop, or None -> pred f32 [B,Co,H,W]
        (a workspace buffer: consume it before the next forward)"""
        d = self.d
        B, C, H, W = x.shape
        assert C == d.input_channels and x.dtype == torch.float32 and x.is_cuda
        self._alloc(B, H, W, train)
        w = self.ws
        N, D, L = self.geo[5], d.inner_dim, d.depth
        self._train, self._yeff = train, y_eff
        xs = w["x"]
        self._stem_cond_fwd(x, t, y_eff, xs[0])
        pend = None  # (x_base, t, gate) of the sub-layer whose gated residual is applied by the next LayerNorm kernel
        for i in range(L):
            pend = self._blk_fwd(w["layers"][i if train else 0], f"layers.{i}.", i * 6 * D, xs[i] if train else xs[i & 1], pend, B, N)
        return self._head_fwd(xs[L] if train else xs[L & 1], pend, N)

    # ------------------------------------------------------------------ backward pieces
    def feature(self, k: int) -> Tensor:
        """output of block k of the last train-mode forward (f32 [B, N, D]): what a forward hook on ``layers[k]`` sees"""
        assert self._train, "block outputs are only kept by the train-mode launch sequence"
        B, _, _, _, _, N, _, _, _ = self.geo
        return self.ws["x"][k + 1].view(B, N, self.d.inner_dim)

    @staticmethod
    def _other(s: dict, cur: Tensor) -> Tensor:
        """ping-pong target of the next LayerNorm backward (never the buffer it reads)"""
        return s["dxb"] if cur.data_ptr() == s["dxa"].data_ptr() else s["dxa"]

    def _head_bwd(self, dpred: Tensor, x_last: Tensor, s: dict, N: int, dres: Tensor | None, gate_fused: dict,
                  mods: tuple | None = None, mo: int | None = None) -> Tensor:
        """last linear (mmdit.py:548) + final adaLN (mmdit.py:543-547); returns the gradient at the head's input (s["dxa"]).
        mods = (mod, dmod, rows per modulation row) when they are not the per-sample matrices (N rows)"""
        d, w = self.d, self.ws
        D, M = d.inner_dim, x_last.shape[0]
        mo = d.depth * 6 * D if mo is None else mo
        mod, dmod, N = mods if mods is not None else (w["mod"], w["dmod"], N)
        scr = w["scr"]
        ops.f32_patchify(dpred, w["dO"], d.patch_size, ops.PATCH_PPC)
        ops.f32_linear_wgrad(w["dO"], w["xf"], self.GW("last_layer.linear.weight"), scratch=scr)
        ops.colsum(w["dO"], self.G("last_layer.linear.bias"), M, w["dO"].shape[1], scratch=scr)
        ops.f32_linear_dgrad(w["dO"], self.W("last_layer.linear.weight"), s["dxm"])
        ops.f32_ln_modulate_bwd(s["dxm"], x_last, None, None, mod[:, mo : mo + D], N, w["meanf"], w["rstdf"], dres, s["dxa"],
                                dmod[:, mo : mo + D], dmod[:, mo + D : mo + 2 * D], None, **gate_fused)
        return s["dxa"]

    def _blk_bwd(self, a: dict, pre: str, mo: int, xin: Tensor, s: dict, dx: Tensor, B: int, nt: int, pos: Tensor | None,
                 gate_fused: dict, dxin_aux: Tensor | None = None, mods: tuple | None = None) -> Tensor:
        """backward of _blk_fwd.  On entry s["dt2"] = gradient of this block's MLP output t2 and dx = gradient of the block output
        through the residual path; gate_fused = the gated-residual backward of the PREVIOUS sub-layer fused into the last LayerNorm
        backward (gate_t / gate / dt / dgate), {} at a stage start; dxin_aux = gradient of an auxiliary loss on the block INPUT (it has to
        pass through that fused gate backward too); mods as in _head_bwd.  Returns the gradient at the block input."""
        d, w = self.d, self.ws
        D, Hh, dh, F = d.inner_dim, d.num_heads, d.head_dim, d.mlp_ratio * d.inner_dim
        P, Wt, G, GW = self.P, self.W, self.G, self.GW
        mod, dmod, rpm = mods if mods is not None else (w["mod"], w["dmod"], nt)
        groups = B * nt // rpm
        dwb = w["dwb"] if groups == B else w["dwbt"]  # partial [groups, 2, D] sums of the affine LayerNorm gradients
        scr = w["scr"]
        gh, gw = self.geo[3], self.geo[4]
        cos, sin = self._rope[(gh, gw)]
        rot = sum(d.rope_axes_dim)
        scale = dh**-0.5
        # MLP branch (mmdit.py:260-264, 305-308)
        ops.f32_linear_wgrad(s["dt2"], a["h"], GW(pre + "mlp_input.2.weight"), scratch=scr)
        ops.f32_linear_dgrad(s["dt2"], Wt(pre + "mlp_input.2.weight"), s["dh"])
        ops.f32_swiglu_bwd(s["dh"], a["u"], s["du"])
        ops.f32_linear_wgrad(s["du"], a["xm2"], GW(pre + "mlp_input.0.weight"), scratch=scr)
        ops.f32_linear_dgrad(s["du"], Wt(pre + "mlp_input.0.weight"), s["dxm"])
        dx_alt = self._other(s, dx)
        ops.f32_ln_modulate_bwd(s["dxm"], a["x1"], P(pre + "norm_2.weight"), P(pre + "norm_2.bias"), mod[:, mo + 3 * D : mo + 4 * D],
                                rpm, a["mean2"], a["rstd2"], dx, dx_alt, dmod[:, mo + 3 * D : mo + 4 * D],
                                dmod[:, mo + 4 * D : mo + 5 * D], dwb, gate_t=a["t1"], gate=mod[:, mo + 2 * D : mo + 3 * D],
                                dt=s["dt1"], dgate=dmod[:, mo + 2 * D : mo + 3 * D])
        self._fold_groups(dwb, G(pre + "norm_2.weight"), groups, 2 * D)  # [w; b] adjacent; fixed-order fold
        dx = dx_alt
        # attention branch (mmdit.py:75-104)
        ops.f32_linear_wgrad(s["dt1"], a["a"], GW(pre + "attention.proj_out.weight"), scratch=scr)
        ops.f32_linear_dgrad(s["dt1"], Wt(pre + "attention.proj_out.weight"), s["da"])
        hb = dict(batch=(B, Hh))
        # dP = dO V^T
        ops.f32_gemm(s["da"], a["qkv"], s["dP"], nt, nt, dh, lda=D, ldb=3 * D, ldc=nt, b_off=2 * D, sa=(nt * D, dh),
                     sb=(nt * 3 * D, dh), sc=(Hh * nt * nt, nt * nt), **hb)
        # dV = P^T dO -> the v third of dqkv
        ops.f32_gemm(a["P"], s["da"], s["dqkv"], nt, dh, nt, lda=nt, ldb=D, ldc=3 * D, ta=True, tb=True, c_off=2 * D,
                     sa=(Hh * nt * nt, nt * nt), sb=(nt * D, dh), sc=(nt * 3 * D, dh), **hb)
        ops.f32_softmax_bwd(a["P"], s["dP"], B * Hh * nt, nt)  # dS over dP
        # dQ = scale dS K ; dK = scale dS^T Q (gradients of the normalised + rotated q, k, token-major)
        ops.f32_gemm(s["dP"], a["qk"], s["dqk"], nt, dh, nt, lda=nt, ldb=2 * D, ldc=2 * D, tb=True, b_off=D, sa=(Hh * nt * nt, nt * nt),
                     sb=(nt * 2 * D, dh), sc=(nt * 2 * D, dh), alpha=scale, **hb)
        ops.f32_gemm(s["dP"], a["qk"], s["dqk"], nt, dh, nt, lda=nt, ldb=2 * D, ldc=2 * D, ta=True, tb=True, c_off=D,
                     sa=(Hh * nt * nt, nt * nt), sb=(nt * 2 * D, dh), sc=(nt * 2 * D, dh), alpha=scale, **hb)
        ops.f32_qk_norm_rope_bwd(s["dqk"], a["qkv"], P(pre + "attention.qk_norm.query_norm.scale"),
                                 P(pre + "attention.qk_norm.key_norm.scale"), cos, sin, a["rrms"], s["dqkv"], w["dqs"], B, nt, Hh, dh,
                                 rot, pos=pos)
        ops.reduce_rows_batched_f32(w["dqs"], 0, G(pre + "attention.qk_norm.query_norm.scale"), 0, 1, B, 2 * D)  # [q; k] adjacent
        ops.f32_linear_wgrad(s["dqkv"], a["xm1"], GW(pre + "attention.qkv.weight"), scratch=scr)
        ops.f32_linear_dgrad(s["dqkv"], Wt(pre + "attention.qkv.weight"), s["dxm"])
        if dxin_aux is not None:
            ops.f32_add(dx, dxin_aux, dx)
        dx_alt = self._other(s, dx)
        ops.f32_ln_modulate_bwd(s["dxm"], xin, P(pre + "norm_1.weight"), P(pre + "norm_1.bias"), mod[:, mo : mo + D], rpm,
                                a["mean1"], a["rstd1"], dx, dx_alt, dmod[:, mo : mo + D], dmod[:, mo + D : mo + 2 * D], dwb,
                                **gate_fused)
        self._fold_groups(dwb, G(pre + "norm_1.weight"), groups, 2 * D)
        return dx_alt

    def _prev_gate(self, a_prev: dict, mp: int, s: dict, mods: tuple | None = None) -> dict:
        """arguments that fuse the gated-residual backward of the block in front (its t2 / MLP gate) into a LayerNorm backward"""
        D = self.d.inner_dim
        mod, dmod = (self.ws["mod"], self.ws["dmod"]) if mods is None else mods[:2]
        return dict(gate_t=a_prev["t2"], gate=mod[:, mp + 5 * D : mp + 6 * D], dt=s["dt2"], dgate=dmod[:, mp + 5 * D : mp + 6 * D])

    def _fold_groups(self, partial: Tensor, out: Tensor, groups: int, n: int) -> None:
        """out[:n] += the sum of `groups` partial rows, in a fixed order (per-sample partials: one launch; per-token: slab partials)"""
        if groups <= 1024:
            ops.reduce_rows_batched_f32(partial, 0, out, 0, 1, groups, n)
        else:
            ops.colsum(partial.view(groups, n), out, groups, n, scratch=self.ws["scr"])

    def _stem_cond_bwd(self, dx0: Tensor, extra_de: Tensor | None = None) -> None:
        """patch-embedding weight gradient (no gradient flows to the input latents) and the conditioning path; extra_de: a further
        gradient of the time embedding e (DDT's per-token decoder conditioning)"""
        d, w = self.d, self.ws
        B, E = self.geo[0], d.embedding_dim
        G, GW, Wt = self.G, self.GW, self.W
        dmod, scr = w["dmod"], w["scr"]
        ops.f32_linear_wgrad(dx0, w["tokP"], GW(self._conv_name), scratch=scr)
        g_modw, g_modb = self._mod_matrix(self.grads)
        mod_w, _ = self._mod_matrix(self.params)
        ops.f32_linear_wgrad(dmod, w["se"], g_modw)
        ops.colsum(dmod, g_modb, B, self.layout.mod_rows, scratch=scr)
        ops.f32_linear_dgrad(dmod, mod_w, w["dse"], scratch=scr)
        table = d.n_classes is not None
        ops.f32_cond_combine_bwd(w["dse"], w["emb"], self._yeff if table else None, w["demb"],
                                 G("label_embed.embedding.weight") if table else None)
        if extra_de is not None:
            ops.f32_add(w["demb"], extra_de, w["demb"])
        ops.colsum(w["demb"], G("time_embed.2.bias"), B, E, scratch=scr)
        ops.f32_linear_wgrad(w["demb"], w["h1"], GW("time_embed.2.weight"))
        ops.f32_linear_dgrad(w["demb"], Wt("time_embed.2.weight"), w["dh1"])
        ops.f32_silu_bwd(w["dh1"], w["pre1"], w["dpre1"])
        ops.f32_linear_wgrad(w["dpre1"], w["temb"], GW("time_embed.0.weight"))
        ops.colsum(w["dpre1"], G("time_embed.0.bias"), B, E, scratch=scr)
        if self.reducer is not None:  # data parallel: one exchange over the whole arena once the backward has ended
            self.reducer.ready(0, self.layout.size)
            self.reducer.finish()

    def backward(self, dpred: Tensor, dfeats: dict[int, Tensor] | None = None) -> None:
        """accumulates d(loss)/d(param) into the flat gradient arena (+=) for the last train-mode forward"""
        assert self._train and self.grads is not None
        d, w = self.d, self.ws
        B, N, M = self.geo[0], self.geo[5], self.geo[6]
        D, L = d.inner_dim, d.depth
        dfeats = {k: g.reshape(M, D).float().contiguous() for k, g in (dfeats or {}).items()}
        xs, s = w["x"], w["s"]
        dx = self._head_bwd(dpred, xs[L], s, N, dfeats.get(L - 1), self._prev_gate(w["layers"][L - 1], (L - 1) * 6 * D, s))
        for i in reversed(range(L)):
            fused = self._prev_gate(w["layers"][i - 1], (i - 1) * 6 * D, s) if i > 0 else {}
            # (dfeats[i - 1]: auxiliary-loss gradient on the output of block i-1 = this block's input)
            dx = self._blk_bwd(w["layers"][i], f"layers.{i}.", i * 6 * D, xs[i], s, dx, B, N, None, fused, dfeats.get(i - 1))
        self._stem_cond_bwd(dx)
