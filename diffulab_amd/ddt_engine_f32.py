"""fp32-class launch sequences of DDT(simple_ddt=True): the precision the reference's class-conditional configurations give this denoiser
(``model=ddt`` composed with e.g. train_cifar10_flow_matching.yaml inherits trainer/default.yaml's `precision_type: "no"`; this
repository's ``configs/train_cifar10_ddt.yaml`` is that composition; reference networks/denoisers/ddt.py:346-464).

Encoder = a stage of DiT blocks with per-sample adaLN rows (`sprint_engine_f32.SprintEngineF32._stage_fwd`).  Decoder = DiT blocks on
a second patch embedding of the input whose conditioning is PER TOKEN, z = silu(encoder output + time embedding): the adaLN linears of
all decoder blocks and of the last layer are ONE f32 GEMM over the tokens ([B*N, D] x [D, 6D*depth + 2D]) and the LayerNorm kernels
read / write one modulation row per token (rows_per_mod = 1), as in `ddt_engine.PerTokenDecoder`.  Same arena layout as the bf16 engine
(`ddt_engine.DDTLayout`); every sum has one producer.
"""

from __future__ import annotations

import torch
from torch import Tensor

from . import ops
from .ddt_engine import DDTDims, DDTLayout
from .sprint_engine_f32 import SprintEngineF32


class DDTEngineF32(SprintEngineF32):
    _conv_name = "conv_proj_encoder.weight"

    def _make_layout(self, d: DDTDims) -> DDTLayout:  # type: ignore[override]
        return DDTLayout(d)

    def _tmod(self, flat: Tensor) -> tuple[Tensor, Tensor]:
        """the stacked per-token adaLN matrix [R, D] and bias [R] of the decoder blocks + last layer"""
        lay, D = self.layout, self.d.inner_dim
        R = lay.tmod_rows
        w0, b0 = lay.entries[lay.tmod_w0][0], lay.entries[lay.tmod_b0][0]
        return flat[w0 : w0 + R * D].view(R, D), flat[b0 : b0 + R]

    # ------------------------------------------------------------------ workspace
    def _alloc(self, B: int, H: int, W: int, train: bool, k: int | None = None) -> None:  # type: ignore[override]
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
        M, R = B * N, self.layout.tmod_rows
        z = self._z
        w: dict[str, object] = {}
        self._common_buffers(w, B, M, train)
        blk = []
        for _ in range(L):
            a = self._block_buffers(B, N)
            a["x0"] = z(M, D)
            blk.append(a)
        w["blk"] = blk
        w["x_stem"], w["xdec_in"], w["enc_out"], w["sz"], w["xl"] = z(M, D), z(M, D), z(M, D), z(M, D), z(M, D)
        w["tmod"] = z(M, R)
        w["pred"] = z(B, d.output_channels, H, W)
        if train:
            w[f"s{N}"] = self._chain_buffers(B, N)
            w["dtmod"] = z(M, R)
            w["dwbt"] = z(M, 2, D)  # per-token partials of the decoder's affine LayerNorm gradients
            w["dsz"], w["denc"] = z(M, D), z(M, D)
            w["dtemb"] = z(B, d.embedding_dim)
            if w["scr"].numel() < 64 * R * D:  # split-K scratch of the stacked adaLN weight gradient
                w["scr"] = torch.empty(64 * R * D, device=self.dev, dtype=torch.float32)
        self._publish(w, key, (B, H, W, gh, gw, N, M, d.input_channels * p * p, d.output_channels * p * p))

    # ------------------------------------------------------------------ forward
    def forward(self, x: Tensor, t: Tensor, y_eff: Tensor | None, train: bool = True, refresh: bool = True) -> Tensor:  # type: ignore[override]
        d = self.d
        B, C, H, W = x.shape
        assert C == d.input_channels and x.dtype == torch.float32 and x.is_cuda
        self._alloc(B, H, W, train)
        w = self.ws
        N = self.geo[5]
        D, L, ne = d.inner_dim, d.depth, d.encoder_depth
        self._train, self._yeff = train, y_eff
        self._stem_cond_fwd(x, t, y_eff, w["x_stem"])
        ops.f32_linear(w["tokP"], self.W("conv_proj_decoder.weight"), w["xdec_in"])
        self._stage_fwd(range(0, ne), w["x_stem"], N, None, w["enc_out"])
        # per-token conditioning of the decoder and the stacked adaLN GEMM of its blocks + last layer (ddt.py:423-431)
        ops.f32_ddt_cond_fwd(w["enc_out"], w["e"], B, N, w["sz"])
        tw, tb = self._tmod(self.params)
        tm = w["tmod"]
        ops.f32_linear(w["sz"], tw, tm, bias=tb)
        pend = None
        for j in range(L - ne):
            a = w["blk"][ne + j]
            a["xin"] = w["xdec_in"] if pend is None else a["x0"]
            pend = self._blk_fwd(a, self.prefixes[ne + j], j * 6 * D, a["xin"], pend, B, N, None, mod=tm, rpm=1)
        return self._head_fwd(w["xl"], pend, N, mod=tm, rpm=1, mo=(L - ne) * 6 * D)

    def feature(self, kblk: int) -> Tensor:
        """output of encoder block k (``layers[k]``) of the last train-mode forward"""
        assert self._train and 0 <= kblk < self.d.encoder_depth
        B, N, D = self.geo[0], self.geo[5], self.d.inner_dim
        if kblk + 1 < self.d.encoder_depth:
            return self.ws["blk"][kblk + 1]["x0"].view(B, N, D)
        return self.ws["enc_out"].view(B, N, D)

    # ------------------------------------------------------------------ backward
    def backward(self, dpred: Tensor, dfeats: dict[int, Tensor] | None = None) -> None:
        assert self._train and self.grads is not None
        d, w = self.d, self.ws
        B, N, M = self.geo[0], self.geo[5], self.geo[6]
        D, L, ne = d.inner_dim, d.depth, d.encoder_depth
        nd = L - ne
        dfe = {kb: g.reshape(M, D).float().contiguous() for kb, g in (dfeats or {}).items()}
        s, scr = w[f"s{N}"], w["scr"]
        mods = (w["tmod"], w["dtmod"], 1)
        dx = self._head_bwd(dpred, w["xl"], s, N, None, self._prev_gate(w["blk"][L - 1], (nd - 1) * 6 * D, s, mods), mods, nd * 6 * D)
        for j in reversed(range(nd)):
            a = w["blk"][ne + j]
            fused = self._prev_gate(w["blk"][ne + j - 1], (j - 1) * 6 * D, s, mods) if j > 0 else {}
            dx = self._blk_bwd(a, self.prefixes[ne + j], j * 6 * D, a["xin"], s, dx, B, N, None, fused, None, mods)
        ops.f32_linear_wgrad(dx, w["tokP"], self.GW("conv_proj_decoder.weight"), scratch=scr)
        # stacked per-token adaLN GEMM: weight / bias gradients and the gradient of its input silu(z)
        tw, _ = self._tmod(self.params)
        gw_, gb_ = self._tmod(self.grads)
        dtm = w["dtmod"]
        ops.f32_linear_wgrad(dtm, w["sz"], gw_, scratch=scr)
        ops.colsum(dtm, gb_, M, self.layout.tmod_rows, scratch=scr)
        ops.f32_linear_dgrad(dtm, tw, w["dsz"])
        ops.f32_ddt_cond_bwd(w["dsz"], w["enc_out"], w["e"], B, N, w["denc"], w["dtemb"])
        dx0 = self._stage_bwd(range(0, ne), w["denc"], N, None, dfe)
        self._stem_cond_bwd(dx0, extra_de=w["dtemb"])
