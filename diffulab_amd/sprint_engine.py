"""Host-side runtime of SprintDiT (simple_dit): the DiT engine's arena / shadows / conditioning path with the block stack cut
into three stages and the SPRINT token routing between them (reference networks/denoisers/sprint.py:505-573):

    encoder blocks (all N tokens) -> gather the kept tokens -> deep blocks (k tokens, RoPE rows picked by position index)
    -> restore into a mask-token canvas -> fuse Linear(2D -> D) on [restored ; encoder output] -> decoder blocks -> last layer

Inside a stage the launch sequence is the DiT engine's (gated residuals absorbed by the next LayerNorm-modulate kernel, weight
gradients on the side stream); at a stage boundary the pending residual is materialised by dl_gated_residual_fwd and its backward
is dl_gate_bwd.  The concatenation [restored ; encoder output] is one [B*N, 2D] buffer written in place by the restore kernel
(left half) and the encoder's last residual (right half), so `fuse` is a single GEMM with K = 2D.
"""

from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import Tensor

from . import ops, tuning
from .engine import DiTDims, DiTEngine, ParamLayout, _must, _rup, rope_grid_tables


@dataclass
class SprintDims(DiTDims):
    encoder_depth: int = 2
    deep_layers_depth: int = 8
    decoder_depth: int = 2
    drop_rate: float = 0.75

    def __post_init__(self) -> None:
        super().__post_init__()
        self.depth = self.encoder_depth + self.deep_layers_depth + self.decoder_depth

    def n_kept(self, S: int) -> int:
        """sprint.py:342"""
        return max(1, int(S * (1.0 - float(self.drop_rate))))


@dataclass
class Route:
    """token routing of one forward: idx int32 [B, k] kept positions (ascending per sample), inv int32 [B, N] their inverse
    (-1 = the canvas keeps the mask token there; every entry of a sample whose deep path is dropped), keep int32 [B] (0 = deep
    path dropped) or None, skip_deep: the p >= 1 branch (no deep layers at all)"""
    idx: Tensor | None
    inv: Tensor
    keep: Tensor | None
    k: int
    skip_deep: bool = False
    pos_lat: Tensor | None = None  # joint form: RoPE table row of every [text ; kept image] token, int32 [B * (Lc + k)]


class SprintEngine(DiTEngine):
    def _make_layout(self, d: SprintDims) -> ParamLayout:
        D = d.inner_dim
        pre = ([f"layers.{i}." for i in range(d.encoder_depth)] + [f"deep_layers.{i}." for i in range(d.deep_layers_depth)]
               + [f"decoder_layers.{i}." for i in range(d.decoder_depth)])
        return ParamLayout(d, pre, extra=(("mask_token", (1, 1, D)), ("fuse.weight", (D, 2 * D))))

    def _extra_shadows(self, reg) -> None:
        reg("fuse.weight", self.d.inner_dim, 2 * self.d.inner_dim)

    # ------------------------------------------------------------------ workspace
    def _alloc(self, B: int, H: int, W: int, train: bool, k: int | None = None) -> None:  # type: ignore[override]
        d, dev = self.d, self.dev
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
        D, E, L = d.inner_dim, d.embedding_dim, d.depth
        for n in (N, k):
            if n % 64 or (n > 256 and (n % 256 or n > 2048)):
                raise NotImplementedError(f"SprintDiT HIP path: token counts must be multiples of 64 up to 256, or of 256 up to "
                                          f"2048 (grid {gh}x{gw} = {N} tokens, {k} kept)")
        M, Bp, Fo = B * N, _rup(B, 64), p * p * d.output_channels
        bf, f32 = torch.bfloat16, torch.float32
        F = d.mlp_ratio * D

        def z(*shape, dtype=bf):
            with torch.inference_mode(False):
                return torch.zeros(*shape, device=dev, dtype=dtype)

        w: dict[str, object] = {"tokP": z(M, self._ki), "temb": z(Bp, d.frequency_embedding), "pre1": z(Bp, E), "h1": z(Bp, E),
                                "e": z(Bp, E, dtype=f32), "emb": z(Bp, E, dtype=f32), "se": z(Bp, E),
                                "mod": z(Bp, self.layout.mod_rows)}
        ne, nd = d.encoder_depth, d.deep_layers_depth
        tokens = [N] * ne + [k] * nd + [N] * d.decoder_depth
        blk = []
        for nt in tokens:
            mt = B * nt
            a = {"x0": z(mt, D), "mean1": z(mt, dtype=f32), "rstd1": z(mt, dtype=f32), "xm1": z(mt, D), "qkv": z(mt, 3 * D),
                 "q": z(B, d.num_heads, nt, 64), "k": z(B, d.num_heads, nt, 64), "v": z(B, d.num_heads, nt, 64),
                 "rrms": z(mt, 2, dtype=f32), "a": z(mt, D), "lse": z(B, d.num_heads, nt, dtype=f32), "t1": z(mt, D),
                 "x1": z(mt, D), "mean2": z(mt, dtype=f32), "rstd2": z(mt, dtype=f32), "xm2": z(mt, D), "u": ops.mlp_u_buffer(z, mt, D, F, train),
                 "h": z(mt, F), "t2": z(mt, D)}
            if train:
                a["wg"] = {"dt2": z(mt, D), "du": z(mt, 2 * F), "dt1": z(mt, D), "dqkv": z(mt, 3 * D)}
                a["dwb"] = z(2, B, 2, D, dtype=f32)
            blk.append(a)
        w["blk"] = blk
        w["x_stem"] = z(M, D)
        w["x"] = [w["x_stem"]]               # (_stem_fwd writes the patch embedding into ws["x"][0])
        w["cat"] = z(M, 2 * D)               # [restored canvas | encoder output]
        w["xd0"], w["xd_out"] = z(B * k, D), z(B * k, D)
        w["xfuse"], w["xdec"] = z(M, D), z(M, D)
        w["meanf"], w["rstdf"], w["xf"] = z(M, dtype=f32), z(M, dtype=f32), z(M, D)
        w["otok"] = z(M, _rup(Fo, 8), dtype=f32)
        w["pred"] = z(B, d.output_channels, H, W, dtype=f32)
        if train:
            w["dO"] = z(M, self._ko)
            for nt in {N, k}:  # chain scratch per token count
                mt = B * nt
                w[f"s{nt}"] = {"dxa": z(mt, D), "dxb": z(mt, D), "dxm": z(mt, D), "da": z(mt, D), "dh": z(mt, F),
                               "dq": z(B, d.num_heads, nt, 64), "dk": z(B, d.num_heads, nt, 64), "dv": z(B, d.num_heads, nt, 64)}
            w["dleft"], w["dright"], w["dxd"] = z(M, D), z(M, D), z(B * k, D)
            w["dmod"] = z(Bp, self.layout.mod_rows)
            w["dmod32"] = z(Bp, self.layout.mod_rows, dtype=f32)
            w["dse"], w["demb"], w["demb16"] = z(Bp, E, dtype=f32), z(Bp, E, dtype=f32), z(Bp, E)
            w["dh1"], w["dpre1"] = z(Bp, E, dtype=f32), z(Bp, E)
            w["scr_last"], w["scr_conv"] = z(_rup(Fo, 8), D, dtype=f32), z(D, self._ki, dtype=f32)
            # scratch of the bit-reproducible (and faster) small GEMMs / column sums of the conditioning backward (engine._cond_bwd)
            w["det_scr"] = torch.empty(max(2 * self.layout.mod_rows * d.embedding_dim, 1 << 22), device=self.dev, dtype=f32)
            if ops.WgradGroups.widths_ok(D, F) and tuning.on("DL_WGRAD_GROUP"):  # grouped weight gradients (ops.WgradGroups)
                w["tn_slab"] = torch.empty(ops.WgradGroups.slab_floats(D, F), device=self.dev, dtype=f32)
            if D <= 512 and tuning.on("DL_QK_INPLACE"):  # scale-gradient partials of the in-place QK-norm backward (ops.qk_inplace_ok)
                w["qk_part"] = torch.empty(1024 * 2 * D, device=self.dev, dtype=f32)
        self.ws, self._ws_key = w, key
        self.geo = (B, H, W, gh, gw, N, M, Bp, Fo)
        if len(self._ws_cache) >= 8:
            self._ws_cache.pop(next(iter(self._ws_cache)))
        self._ws_cache[key] = (w, self.geo)
        if (gh, gw) not in self._rope:
            c, s = rope_grid_tables(gh, gw, d.rope_axes_dim, d.rope_base)
            self._rope[(gh, gw)] = (c.to(dev), s.to(dev))

    # ------------------------------------------------------------------ stages
    def _stage_fwd(self, blocks: range, xin: Tensor, nt: int, pos: Tensor | None, out: Tensor) -> None:
        """DiT blocks `blocks` (indices into the block list) over xin [B*nt, D]; the stage output (last gated residual
        materialised) is written to `out` (rows may be strided)"""
        d, w, sh = self.d, self.ws, self.sh
        B = self.geo[0]
        D, Hh = d.inner_dim, d.num_heads
        gh, gw = self.geo[3], self.geo[4]
        cos, sin = self._rope[(gh, gw)]
        rot = sum(d.rope_axes_dim)
        mod = w["mod"]
        pend = None
        for bi in blocks:
            a, pre, mo = w["blk"][bi], self.prefixes[bi], bi * 6 * D
            n1w, n1b = self.P(pre + "norm_1.weight"), self.P(pre + "norm_1.bias")
            if pend is None:
                xcur = xin
                ops.ln_modulate_fwd(xcur, n1w, n1b, mod[:, mo : mo + D], mod[:, mo + D : mo + 2 * D], nt, 1e-5, a["xm1"], a["mean1"],
                                    a["rstd1"])
            else:
                xcur = a["x0"]
                ops.ln_modulate_fwd(pend[0], n1w, n1b, mod[:, mo : mo + D], mod[:, mo + D : mo + 2 * D], nt, 1e-5, a["xm1"],
                                    a["mean1"], a["rstd1"], t=pend[1], gate=pend[2], x_out=xcur)
            a["xin"] = xcur
            ops.gemm_nt(a["xm1"], sh[pre + "attention.qkv.weight|f"], a["qkv"])
            ops.qk_norm_rope_fwd(a["qkv"], self.P(pre + "attention.qk_norm.query_norm.scale"),
                                 self.P(pre + "attention.qk_norm.key_norm.scale"), cos, sin, a["q"], a["k"],
                                 None if ops.v_in_place(nt) else a["v"], a["rrms"], B, nt,
                                 Hh, 64, rot, pos=pos)
            if ops.v_in_place(nt):  # V read in place from the qkv rows (engine.py)
                ops.attn_fwd_qkv(a["q"], a["k"], a["qkv"], a["a"], a["lse"], B, Hh, nt, 64, 64**-0.5)
            else:
                ops.attn_fwd(a["q"], a["k"], a["v"], a["a"], a["lse"], B, Hh, nt, 64, 64**-0.5)
            ops.gemm_nt(a["a"], sh[pre + "attention.proj_out.weight|f"], a["t1"])
            ops.ln_modulate_fwd(xcur, self.P(pre + "norm_2.weight"), self.P(pre + "norm_2.bias"), mod[:, mo + 3 * D : mo + 4 * D],
                                mod[:, mo + 4 * D : mo + 5 * D], nt, 1e-5, a["xm2"], a["mean2"], a["rstd2"], t=a["t1"],
                                gate=mod[:, mo + 2 * D : mo + 3 * D], x_out=a["x1"])
            if not ops.gemm_nt_swiglu(a["xm2"], sh[pre + "mlp_input.0.weight|g"], a["u"] if self._train else None, a["h"]):
                ops.gemm_nt(a["xm2"], sh[pre + "mlp_input.0.weight|f"], a["u"])
                ops.swiglu_fwd(a["u"], a["h"])
            ops.gemm_nt(a["h"], sh[pre + "mlp_input.2.weight|f"], a["t2"])
            pend = (a["x1"], a["t2"], mod[:, mo + 5 * D : mo + 6 * D])
        ops.gated_residual_fwd(pend[0], pend[1], pend[2], nt, out)

    def _stage_bwd(self, blocks: range, dx: Tensor, nt: int, pos: Tensor | None, dfe: dict[int, Tensor], wgrad, fold_norm,
                   side) -> Tensor:
        """backward of _stage_fwd: dx = gradient at the stage output, contiguous bf16 [B*nt, D] (overwritten); returns the
        gradient at the stage input.  dfe: {block index: auxiliary-loss gradient at that block's output}"""
        d, w, sh = self.d, self.ws, self.sh
        B = self.geo[0]
        D, Hh = d.inner_dim, d.num_heads
        gh, gw = self.geo[3], self.geo[4]
        cos, sin = self._rope[(gh, gw)]
        rot = sum(d.rope_axes_dim)
        mod, dmod = w["mod"], w["dmod32"]
        s = w[f"s{nt}"]
        blocks = list(blocks)
        last = blocks[-1]
        inplace_qk = ops.qk_inplace_ok(w, B, nt)
        if last in dfe:
            ops.add_bf16(dx, dfe[last], dx)
        ml = last * 6 * D
        ops.gate_bwd(dx, w["blk"][last]["t2"], mod[:, ml + 5 * D : ml + 6 * D], nt, w["blk"][last]["wg"]["dt2"],
                     dmod[:, ml + 5 * D : ml + 6 * D])

        def other(cur: Tensor) -> Tensor:  # ping-pong target of the next LayerNorm backward (never the buffer it reads)
            return s["dxb"] if cur.data_ptr() == s["dxa"].data_ptr() else s["dxa"]

        for j in reversed(range(len(blocks))):
            bi = blocks[j]
            a, pre, mo = w["blk"][bi], self.prefixes[bi], bi * 6 * D
            g = a["wg"]
            wgrad(g["dt2"], a["h"], pre + "mlp_input.2.weight")
            ops.mlp_swiglu_bwd(g["dt2"], sh[pre + "mlp_input.2.weight|t"], a["xm2"], sh[pre + "mlp_input.0.weight|g"], a["u"], s["dh"], g["du"])
            wgrad(g["du"], a["xm2"], pre + "mlp_input.0.weight")
            ops.gemm_nt(g["du"], sh[pre + "mlp_input.0.weight|t"], s["dxm"])
            dx_alt = other(dx)
            ops.ln_modulate_bwd(s["dxm"], a["x1"], self.P(pre + "norm_2.weight"), self.P(pre + "norm_2.bias"),
                                mod[:, mo + 3 * D : mo + 4 * D], nt, a["mean2"], a["rstd2"], dx, dx_alt, dmod[:, mo + 3 * D : mo + 4 * D],
                                dmod[:, mo + 4 * D : mo + 5 * D], a["dwb"][1], gate_t=a["t1"], gate=mod[:, mo + 2 * D : mo + 3 * D],
                                dt=g["dt1"], dgate=dmod[:, mo + 2 * D : mo + 3 * D])
            fold_norm(a["dwb"][1], pre + "norm_2.weight")
            dx = dx_alt
            wgrad(g["dt1"], a["a"], pre + "attention.proj_out.weight")
            ops.gemm_nt(g["dt1"], sh[pre + "attention.proj_out.weight|t"], s["da"])
            if inplace_qk:  # dQ, dK, dV token-major into dqkv; the QK-norm backward transforms the q / k thirds in place
                ops.attn_bwd_tok(a["q"], a["k"], a["qkv"], a["a"], s["da"], a["lse"], g["dqkv"], B, Hh, nt, 64, 64**-0.5)
                _must(ops.qk_norm_rope_bwd_inplace(a["qkv"], self.P(pre + "attention.qk_norm.query_norm.scale"),
                                                    self.P(pre + "attention.qk_norm.key_norm.scale"), cos, sin, a["rrms"], g["dqkv"],
                                                    self.G(pre + "attention.qk_norm.query_norm.scale"), w["qk_part"], B, nt, Hh, 64, rot,
                                                    pos=pos))
            else:
                if ops.v_in_place(nt):
                    ops.attn_bwd_qkv(a["q"], a["k"], a["qkv"], a["a"], s["da"], a["lse"], s["dq"], s["dk"], g["dqkv"], B, Hh, nt, 64,
                                     64**-0.5)
                else:
                    ops.attn_bwd(a["q"], a["k"], a["v"], a["a"], s["da"], a["lse"], s["dq"], s["dk"], s["dv"], B, Hh, nt, 64, 64**-0.5)
                ops.qk_norm_rope_bwd(s["dq"], s["dk"], None if ops.v_in_place(nt) else s["dv"], a["qkv"],
                                     self.P(pre + "attention.qk_norm.query_norm.scale"),
                                     self.P(pre + "attention.qk_norm.key_norm.scale"), cos, sin, a["rrms"], g["dqkv"],
                                     self.G(pre + "attention.qk_norm.query_norm.scale"), B, nt, Hh, 64, rot, pos=pos)
            wgrad(g["dqkv"], a["xm1"], pre + "attention.qkv.weight")
            ops.gemm_nt(g["dqkv"], sh[pre + "attention.qkv.weight|t"], s["dxm"])
            nxt = {}
            if j > 0:
                bp = blocks[j - 1]
                if bp in dfe:  # auxiliary-loss gradient on the previous block's output (= this block's input)
                    ops.add_bf16(dx, dfe[bp], dx)
                mp = bp * 6 * D
                nxt = dict(gate_t=w["blk"][bp]["t2"], gate=mod[:, mp + 5 * D : mp + 6 * D], dt=w["blk"][bp]["wg"]["dt2"],
                           dgate=dmod[:, mp + 5 * D : mp + 6 * D])
            dx_alt = other(dx)
            ops.ln_modulate_bwd(s["dxm"], a["xin"], self.P(pre + "norm_1.weight"), self.P(pre + "norm_1.bias"), mod[:, mo : mo + D],
                                nt, a["mean1"], a["rstd1"], dx, dx_alt, dmod[:, mo : mo + D], dmod[:, mo + D : mo + 2 * D],
                                a["dwb"][0], **nxt)
            fold_norm(a["dwb"][0], pre + "norm_1.weight")
            dx = dx_alt
            getattr(wgrad, "flush", lambda: None)()
            if self.reducer is not None:
                self.reducer.ready(*self.layer_ranges[bi], extra_events=(side.record_event(),))
        return dx

    # ------------------------------------------------------------------ forward
    def forward(self, x: Tensor, t: Tensor, y_eff: Tensor | None, train: bool = True, refresh: bool = True,
                route: Route | None = None) -> Tensor:
        d = self.d
        B, C, H, W = x.shape
        assert C == d.input_channels and x.dtype == torch.float32 and x.is_cuda
        route = route if route is not None else self.route
        assert route is not None, "SprintEngine.forward needs the token routing of this step"
        self._alloc(B, H, W, train, route.k)
        if refresh:
            self.refresh_shadows(force=train)
        w, sh = self.ws, self.sh
        _, _, _, gh, gw, N, M, Bp, Fo = self.geo
        D, L, k = d.inner_dim, d.depth, route.k
        ne, nd = d.encoder_depth, d.deep_layers_depth
        self._train, self._yeff, self._route = train, y_eff, route
        mod = self._stem_fwd(x, t, y_eff)
        cat = w["cat"]
        self._stage_fwd(range(0, ne), w["x_stem"], N, None, cat[:, D:])
        mask = self.P("mask_token").view(D)
        if route.skip_deep:
            ops.restore_tokens(w["xd_out"], route.inv, mask, cat[:, :D], B, N, k, D)  # inv == -1 everywhere: mask-token canvas
        else:
            ops.gather_tokens(cat[:, D:], route.idx, w["xd0"], B, N, k, D)
            self._stage_fwd(range(ne, ne + nd), w["xd0"], k, route.idx.view(-1), w["xd_out"])
            ops.restore_tokens(w["xd_out"], route.inv, mask, cat[:, :D], B, N, k, D)
        ops.gemm_nt(cat, sh["fuse.weight|f"], w["xfuse"], M=M, N=D, K=2 * D)
        self._stage_fwd(range(ne + nd, L), w["xfuse"], N, None, w["xdec"])
        mo = L * 6 * D
        ops.ln_modulate_fwd(w["xdec"], None, None, mod[:, mo : mo + D], mod[:, mo + D : mo + 2 * D], N, 1e-6, w["xf"], w["meanf"],
                            w["rstdf"])
        ops.gemm_nt(w["xf"], sh["last_layer.linear.weight|f"], w["otok"], bias=self.P("last_layer.linear.bias"), M=M, N=Fo, K=D)
        ops.unpatchify(w["otok"], w["pred"], d.patch_size)
        return w["pred"]

    route: Route | None = None

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
        d, w, sh = self.d, self.ws, self.sh
        B, H, W, gh, gw, N, M, Bp, Fo = self.geo
        D, L = d.inner_dim, d.depth
        ne, nd = d.encoder_depth, d.deep_layers_depth
        route = self._route
        k = route.k
        dfe = {kb: g.reshape(-1, D).to(torch.bfloat16).contiguous() for kb, g in (dfeats or {}).items()}
        mod, dmod = w["mod"], w["dmod32"]
        dmod[:B].zero_()
        Fo8 = _rup(Fo, 8)
        sN = w[f"s{N}"]

        # head
        ops.patchify(dpred, w["dO"], d.patch_size, ops.PATCH_PPC)
        gl = self.G("last_layer.linear.weight")
        if Fo == Fo8:
            ops.gemm_tn(w["dO"], w["xf"], gl, M=Fo, N=D)
        else:
            w["scr_last"].zero_()
            ops.gemm_tn(w["dO"], w["xf"], w["scr_last"], M=Fo8, N=D)
            ops.reduce_rows_f32(w["scr_last"], gl, 1, Fo * D)
        ops.colsum(w["dO"], self.G("last_layer.linear.bias"), M, Fo)
        ops.gemm_nt(w["dO"], sh["last_layer.linear.weight|t"], sN["dxm"], M=M, N=D, K=self._ko)
        mo = L * 6 * D
        ops.ln_modulate_bwd(sN["dxm"], w["xdec"], None, None, mod[:, mo : mo + D], N, w["meanf"], w["rstdf"], None, sN["dxa"],
                            dmod[:, mo : mo + D], dmod[:, mo + D : mo + 2 * D], None)

        main = torch.cuda.current_stream()
        side = self._side_stream()
        side.wait_stream(main)
        side_wgs = tuning.integer("DL_SIDE_WGS", 128)

        def on_side(fn) -> None:
            ev = main.record_event()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                fn()

        wgrad = ops.grouped_wgrad_fn(self.G, w.get("tn_slab"), on_side, side_wgs)  # (one atomics-free launch per block where the tile divides)

        def fold_norm(partial: Tensor, gname: str) -> None:
            ev = main.record_event()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                ops.reduce_rows_f32(partial, self.G(gname), B, 2 * D, clear=True)

        dxf = self._stage_bwd(range(ne + nd, L), sN["dxa"], N, None, {}, wgrad, fold_norm, side)
        # fuse: xfuse = [restored | enc] Wf^T
        ops.gemm_tn(dxf, w["cat"], self.G("fuse.weight"))  # (main stream: dxf is chain scratch that the next stage reuses)
        wt = sh["fuse.weight|t"]
        ops.gemm_nt(dxf, wt[:D], w["dleft"], M=M, N=D, K=D)
        ops.gemm_nt(dxf, wt[D:], w["dright"], M=M, N=D, K=D)
        ops.masked_colsum(w["dleft"], route.inv.view(-1), self.G("mask_token").view(D), M, D)
        if not route.skip_deep:
            ops.gather_tokens(w["dleft"], route.idx, w["dxd"], B, N, k, D, keep=route.keep)
            dxd0 = self._stage_bwd(range(ne, ne + nd), w["dxd"], k, route.idx.view(-1), {}, wgrad, fold_norm, side)
            ops.scatter_tokens_add(dxd0, route.idx, w["dright"], B, N, k, D)
        elif self.reducer is not None:
            for bi in reversed(range(ne, ne + nd)):
                self.reducer.ready(*self.layer_ranges[bi])
        dx0 = self._stage_bwd(range(0, ne), w["dright"], N, None, dfe, wgrad, fold_norm, side)
        main.wait_stream(side)
        self._cond_bwd(dx0)
