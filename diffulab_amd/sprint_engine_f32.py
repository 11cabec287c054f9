"""fp32-class launch sequences of SprintDiT (simple_dit): the precision the reference's class-conditional configurations give this
denoiser (``model=sprint`` composed with e.g. train_cifar10_flow_matching.yaml inherits trainer/default.yaml's `precision_type: "no"`;
this repository's ``configs/train_cifar10_sprint.yaml`` is that composition).  `engine_f32.DiTEngineF32`'s stem / conditioning / block / head pieces
composed like `sprint_engine.SprintEngine` (reference networks/denoisers/sprint.py:505-573):

    encoder blocks (all N tokens) -> gather the kept tokens -> deep blocks (k tokens, RoPE rows picked by position index)
    -> restore into a mask-token canvas -> fuse Linear(2D -> D) on [restored ; encoder output] -> decoder blocks -> last layer

with the f32 token-routing kernels of csrc/f32.hip.  Every sum has one producer (partials + fixed-order folds): a step is
bit-reproducible.
"""

from __future__ import annotations

import torch
from torch import Tensor

from . import ops
from .engine import ParamLayout
from .engine_f32 import DiTEngineF32
from .sprint_engine import Route, SprintDims


class SprintEngineF32(DiTEngineF32):
    route: Route | None = None

    def _make_layout(self, d: SprintDims) -> ParamLayout:  # (the bf16 engine's arena: sprint_engine.SprintEngine._make_layout)
        D = d.inner_dim
        pre = ([f"layers.{i}." for i in range(d.encoder_depth)] + [f"deep_layers.{i}." for i in range(d.deep_layers_depth)]
               + [f"decoder_layers.{i}." for i in range(d.decoder_depth)])
        return ParamLayout(d, pre, extra=(("mask_token", (1, 1, D)), ("fuse.weight", (D, 2 * D))))

    # ------------------------------------------------------------------ workspace
    def _alloc(self, B: int, H: int, W: int, train: bool, k: int | None = None) -> None:  # type: ignore[override]
        d = self.d
        p = d.patch_size
        gh, gw = H // p, W // p
        N = gh * gw
        k = N if k is None else k
        key = (B, H, W, train, k)
        if key == self._ws_key:
            return
        if key in self._ws_cache:
            self.ws, self.geo = self._ws_cache[key]
            self._ws_key = key
            return
        D, M = d.inner_dim, B * N
        z = self._z
        w: dict[str, object] = {}
        self._common_buffers(w, B, M, train)
        ne, nd = d.encoder_depth, d.deep_layers_depth
        tokens = [N] * ne + [k] * nd + [N] * d.decoder_depth
        blk = []
        for nt in tokens:  # (every block keeps its own activations also in inference: the stages are short)
            a = self._block_buffers(B, nt)
            a["x0"] = z(B * nt, D)  # the block input when it is a materialised pending residual
            blk.append(a)
        w["blk"] = blk
        w["x_stem"] = z(M, D)
        w["cat"] = z(M, 2 * D)  # [restored canvas | encoder output]
        w["xd0"], w["xd_out"] = z(B * k, D), z(B * k, D)
        w["xfuse"], w["xdec"] = z(M, D), z(M, D)
        w["pred"] = z(B, d.output_channels, H, W)
        if train:
            for nt in {N, k}:
                w[f"s{nt}"] = self._chain_buffers(B, nt)
            w["dleft"], w["dright"], w["dxd"] = z(M, D), z(M, D), z(B * k, D)
        self._publish(w, key, (B, H, W, gh, gw, N, M, d.input_channels * p * p, d.output_channels * p * p))

    # ------------------------------------------------------------------ stages
    def _stage_fwd(self, blocks: range, xin: Tensor, nt: int, pos: Tensor | None, out: Tensor) -> None:
        """DiT blocks `blocks` over xin [B*nt, D]; the stage output (last gated residual materialised) goes to `out` (rows may be
        strided)"""
        w, D, B = self.ws, self.d.inner_dim, self.geo[0]
        pend = None
        for bi in blocks:
            a = w["blk"][bi]
            a["xin"] = xin if pend is None else a["x0"]
            pend = self._blk_fwd(a, self.prefixes[bi], bi * 6 * D, a["xin"], pend, B, nt, pos)
        ops.f32_gated_residual_fwd(pend[0], pend[1], pend[2], nt, out)

    def _stage_bwd(self, blocks: range, dx: Tensor, nt: int, pos: Tensor | None, dfe: dict[int, Tensor]) -> Tensor:
        """backward of _stage_fwd: dx = gradient at the stage output (one of the chain buffers s{nt}["dxa"/"dxb"] or any other
        contiguous f32 [B*nt, D]); returns the gradient at the stage input"""
        w, D, B = self.ws, self.d.inner_dim, self.geo[0]
        mod, dmod = w["mod"], w["dmod"]
        s = w[f"s{nt}"]
        blocks = list(blocks)
        last = blocks[-1]
        if last in dfe:
            ops.f32_add(dx, dfe[last], dx)
        ml = last * 6 * D
        ops.f32_gate_bwd(dx, w["blk"][last]["t2"], mod[:, ml + 5 * D : ml + 6 * D], nt, s["dt2"], dmod[:, ml + 5 * D : ml + 6 * D])
        for j in reversed(range(len(blocks))):
            bi = blocks[j]
            a = w["blk"][bi]
            fused, aux = {}, None
            if j > 0:
                bp = blocks[j - 1]
                fused, aux = self._prev_gate(w["blk"][bp], bp * 6 * D, s), dfe.get(bp)
            dx = self._blk_bwd(a, self.prefixes[bi], bi * 6 * D, a["xin"], s, dx, B, nt, pos, fused, aux)
        return dx

    # ------------------------------------------------------------------ forward
    def forward(self, x: Tensor, t: Tensor, y_eff: Tensor | None, train: bool = True, refresh: bool = True,
                route: Route | None = None) -> Tensor:
        d = self.d
        B, C, H, W = x.shape
        assert C == d.input_channels and x.dtype == torch.float32 and x.is_cuda
        route = route if route is not None else self.route
        assert route is not None, "SprintEngineF32.forward needs the token routing of this step"
        self._alloc(B, H, W, train, route.k)
        w = self.ws
        N, M = self.geo[5], self.geo[6]
        D, L, k = d.inner_dim, d.depth, route.k
        ne, nd = d.encoder_depth, d.deep_layers_depth
        self._train, self._yeff, self._route = train, y_eff, route
        self._stem_cond_fwd(x, t, y_eff, w["x_stem"])
        cat = w["cat"]
        self._stage_fwd(range(0, ne), w["x_stem"], N, None, cat[:, D:])
        mask = self.P("mask_token").view(D)
        if route.skip_deep:
            ops.f32_restore_tokens(w["xd_out"], route.inv, mask, cat[:, :D], B, N, k, D)  # inv == -1 everywhere: mask-token canvas
        else:
            ops.f32_gather_tokens(cat[:, D:], route.idx, w["xd0"], B, N, k, D)
            self._stage_fwd(range(ne, ne + nd), w["xd0"], k, route.idx.view(-1), w["xd_out"])
            ops.f32_restore_tokens(w["xd_out"], route.inv, mask, cat[:, :D], B, N, k, D)
        ops.f32_linear(cat, self.W("fuse.weight"), w["xfuse"])
        self._stage_fwd(range(ne + nd, L), w["xfuse"], N, None, w["xdec"])
        return self._head_fwd(w["xdec"], None, N)

    def feature(self, kblk: int) -> Tensor:
        """output of encoder block k (``layers[k]``, the blocks a REPA hook can attach to) of the last train-mode forward"""
        assert self._train and 0 <= kblk < self.d.encoder_depth
        B, N, D = self.geo[0], self.geo[5], self.d.inner_dim
        if kblk + 1 < self.d.encoder_depth:
            return self.ws["blk"][kblk + 1]["x0"].view(B, N, D)
        return self.ws["cat"].view(B, N, 2 * D)[:, :, D:]

    # ------------------------------------------------------------------ backward
    def backward(self, dpred: Tensor, dfeats: dict[int, Tensor] | None = None) -> None:
        assert self._train and self.grads is not None
        d, w = self.d, self.ws
        B, N, M = self.geo[0], self.geo[5], self.geo[6]
        D, L = d.inner_dim, d.depth
        ne, nd = d.encoder_depth, d.deep_layers_depth
        route = self._route
        k = route.k
        dfe = {kb: g.reshape(M, D).float().contiguous() for kb, g in (dfeats or {}).items()}
        sN, scr = w[f"s{N}"], w["scr"]
        w["dmod"].zero_()  # (with the deep path skipped its blocks' modulation gradients have no producer)
        dx = self._head_bwd(dpred, w["xdec"], sN, N, None, {})
        dxf = self._stage_bwd(range(ne + nd, L), dx, N, None, {})
        # fuse: xfuse = [restored | enc] Wf^T
        ops.f32_linear_wgrad(dxf, w["cat"], self.GW("fuse.weight"), scratch=scr)
        wf = self.W("fuse.weight")
        ops.f32_linear_dgrad(dxf, wf[:, :D], w["dleft"])
        ops.f32_linear_dgrad(dxf, wf[:, D:], w["dright"])
        ops.f32_masked_colsum(w["dleft"], route.inv.view(-1), self.G("mask_token").view(D), M, D, scr)
        if not route.skip_deep:
            ops.f32_gather_tokens(w["dleft"], route.idx, w["dxd"], B, N, k, D, keep=route.keep)
            dxd0 = self._stage_bwd(range(ne, ne + nd), w["dxd"], k, route.idx.view(-1), {})
            ops.f32_scatter_tokens_add(dxd0, route.idx, w["dright"], B, N, k, D)
        dx0 = self._stage_bwd(range(0, ne), w["dright"], N, None, dfe)
        self._stem_cond_bwd(dx0)
