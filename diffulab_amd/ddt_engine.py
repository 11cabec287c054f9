"""Host-side runtime of DDT with simple_ddt=True (reference networks/denoisers/ddt.py:346-464; configs/model/ddt.yaml).

Encoder = a stage of DiT blocks with per-sample adaLN conditioning (the DiT engine's sequence).  Decoder = DiT blocks on a second
patch embedding of the input whose conditioning is PER TOKEN, z = silu(encoder_output + time_embedding): the adaLN linears of all
decoder blocks and of the last layer are stacked into ONE GEMM over the tokens ([B*N, D] x [D, 6D*depth + 2D]); the
LayerNorm-modulate kernels read one modulation row per token (rows_per_mod = 1) and the backward writes the per-token modulation
gradients as bf16 rows straight into the gradient matrix of that GEMM (dl_ln_modulate_bwd_tok).
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import torch
from torch import Tensor

from . import ops, tuning
from .engine import DiTDims, _must, _rup, rope_grid_tables
from .sprint_engine import SprintEngine

N_PART = 32  # partial slabs of the affine LayerNorm gradients of a per-token LayerNorm backward


@dataclass
class DDTDims(DiTDims):
    encoder_depth: int = 8
    decoder_depth: int = 4

    def __post_init__(self) -> None:
        self.embedding_dim = self.inner_dim  # ddt.py:139-176
        super().__post_init__()
        self.depth = self.encoder_depth + self.decoder_depth


class DDTLayout:
    def __init__(self, d: DDTDims) -> None:
        D, p = d.inner_dim, d.patch_size
        E = D
        self.entries: dict[str, tuple[int, tuple[int, ...]]] = {}
        self.size = 0
        enc = [f"layers.{i}." for i in range(d.encoder_depth)]
        dec = [f"decoder_layers.{i}." for i in range(d.decoder_depth)]
        self.prefixes = enc + dec

        def add(name: str, shape: tuple[int, ...], align: int = 64) -> None:
            self.size = _rup(self.size, align)
            self.entries[name] = (self.size, shape)
            self.size += math.prod(shape)

        for suffix, shp in (("weight", lambda r: (r, E)), ("bias", lambda r: (r,))):
            for i, pre in enumerate(enc):  # per-sample adaLN of the encoder blocks: one [ne*6D, E] matrix
                add(pre + "modulation.lin." + suffix, shp(6 * D), align=1 if i else 64)
        for suffix, shp in (("weight", lambda r: (r, E)), ("bias", lambda r: (r,))):
            for i, pre in enumerate(dec):  # per-token adaLN of the decoder blocks + last layer: one [nd*6D + 2D, D] matrix
                add(pre + "modulation.lin." + suffix, shp(6 * D), align=1 if i else 64)
            add("last_layer.adaLN_modulation.1." + suffix, shp(2 * D), align=1)
        self.mod_rows, self.tmod_rows = d.encoder_depth * 6 * D, d.decoder_depth * 6 * D + 2 * D
        self.mod_w0, self.mod_b0 = enc[0] + "modulation.lin.weight", enc[0] + "modulation.lin.bias"
        self.tmod_w0, self.tmod_b0 = dec[0] + "modulation.lin.weight", dec[0] + "modulation.lin.bias"
        if d.n_classes is not None:
            add("label_embed.embedding.weight", (d.n_classes + (1 if d.classifier_free else 0), E))
        add("time_embed.0.weight", (E, d.frequency_embedding))
        add("time_embed.0.bias", (E,))
        add("time_embed.2.weight", (E, E))
        add("time_embed.2.bias", (E,))
        add("conv_proj_encoder.weight", (D, d.input_channels, p, p))
        add("conv_proj_decoder.weight", (D, d.input_channels, p, p))
        add("last_layer.linear.weight", (p * p * d.output_channels, D))
        add("last_layer.linear.bias", (p * p * d.output_channels,))
        self.block_first = [pre + "norm_1.weight" for pre in self.prefixes]
        for pre in self.prefixes:
            add(pre + "norm_1.weight", (D,))
            add(pre + "norm_1.bias", (D,), align=1)
            add(pre + "norm_2.weight", (D,))
            add(pre + "norm_2.bias", (D,), align=1)
            add(pre + "attention.qk_norm.query_norm.scale", (D,))
            add(pre + "attention.qk_norm.key_norm.scale", (D,), align=1)
            add(pre + "attention.qkv.weight", (3 * D, D))
            add(pre + "attention.proj_out.weight", (D, D))
            add(pre + "mlp_input.0.weight", (2 * d.mlp_ratio * D, D))
            add(pre + "mlp_input.2.weight", (D, d.mlp_ratio * D))
        self.size = _rup(self.size, 64)

    def view(self, flat: Tensor, name: str) -> Tensor:
        off, shape = self.entries[name]
        return flat[off : off + math.prod(shape)].view(shape)


class PerTokenDecoder:
    """DDT decoder shared by the class-conditional and the joint-encoder engines: DiT blocks `first .. first + nd` of the block
    list on ws["xdec_in"], conditioned per token on silu(ws["enc_out"] + time embedding) through ONE stacked adaLN GEMM
    (ws["tmod"]), then the modulated last layer.  Expects ws keys xdec_in, enc_out, sz, tmod, xl, xf, meanf, rstdf, otok, pred, e and
    DiT-style per-block dicts ws["blk"][bi] (with "wg" and "dwbp" in training), layout.tmod_rows / tmod_w0 / tmod_b0."""

    def _decoder_fwd(self, first: int, nd: int, cos: Tensor, sin: Tensor) -> Tensor:
        d, w, sh = self.d, self.ws, self.sh
        B, _, _, _, _, N, M, _, Fo = self.geo
        D, Hh = d.inner_dim, d.num_heads
        rot = sum(d.rope_axes_dim)
        train = self._train
        ne, L = first, first + nd
        # per-token conditioning of the decoder and the stacked adaLN GEMM of its blocks + last layer
        ops.ddt_cond_fwd(w["enc_out"], w["e"], B, N, w["sz"])
        R = self.layout.tmod_rows
        tb = self.params[self.layout.entries[self.layout.tmod_b0][0] :][:R]
        tm = w["tmod"]
        ops.gemm_nt(w["sz"], sh["@tmod|f"], tm, bias=tb, M=M, N=R, K=D)
        pend = None
        for j in range(L - ne):
            bi = ne + j
            a, pre, mo = w["blk"][bi], self.prefixes[bi], j * 6 * D
            n1w, n1b = self.P(pre + "norm_1.weight"), self.P(pre + "norm_1.bias")
            if pend is None:
                xcur = w["xdec_in"]
                ops.ln_modulate_fwd(xcur, n1w, n1b, tm[:, mo : mo + D], tm[:, mo + D : mo + 2 * D], 1, 1e-5, a["xm1"], a["mean1"],
                                    a["rstd1"])
            else:
                xcur = a["x0"]
                ops.ln_modulate_fwd(pend[0], n1w, n1b, tm[:, mo : mo + D], tm[:, mo + D : mo + 2 * D], 1, 1e-5, a["xm1"], a["mean1"],
                                    a["rstd1"], t=pend[1], gate=pend[2], x_out=xcur)
            a["xin"] = xcur
            ops.gemm_nt(a["xm1"], sh[pre + "attention.qkv.weight|f"], a["qkv"])
            ops.qk_norm_rope_fwd(a["qkv"], self.P(pre + "attention.qk_norm.query_norm.scale"),
                                 self.P(pre + "attention.qk_norm.key_norm.scale"), cos, sin, a["q"], a["k"],
                                 None if ops.v_in_place(N) else a["v"], a["rrms"], B, N, Hh,
                                 64, rot)
            if ops.attn_needs_padding(N):  # (multi-aspect-ratio buckets: 28 x 36, 24 x 40 ... tokens) q / k / v live in rows padded to a
                # multiple of 256 per (sample, head); the pad keys are masked, the pad queries' outputs never copied out
                Np = a["q"].shape[2]
                ops.attn_fwd_ex(a["q"], a["k"], a["v"], a["ao"], a["lse"], B, Hh, Np, Np, 64, 64**-0.5, w["kb_d"])
                ops.copy_rows3d(a["ao"], Np * D, D, a["a"], N * D, D, B, N, D)
            elif ops.v_in_place(N):  # V read in place from the qkv rows (engine.py)
                ops.attn_fwd_qkv(a["q"], a["k"], a["qkv"], a["a"], a["lse"], B, Hh, N, 64, 64**-0.5)
            else:
                ops.attn_fwd(a["q"], a["k"], a["v"], a["a"], a["lse"], B, Hh, N, 64, 64**-0.5)
            ops.gemm_nt(a["a"], sh[pre + "attention.proj_out.weight|f"], a["t1"])
            ops.ln_modulate_fwd(xcur, self.P(pre + "norm_2.weight"), self.P(pre + "norm_2.bias"), tm[:, mo + 3 * D : mo + 4 * D],
                                tm[:, mo + 4 * D : mo + 5 * D], 1, 1e-5, a["xm2"], a["mean2"], a["rstd2"], t=a["t1"],
                                gate=tm[:, mo + 2 * D : mo + 3 * D], x_out=a["x1"])
            if not ops.gemm_nt_swiglu(a["xm2"], sh[pre + "mlp_input.0.weight|g"], a["u"] if train else None, a["h"]):
                ops.gemm_nt(a["xm2"], sh[pre + "mlp_input.0.weight|f"], a["u"])
                ops.swiglu_fwd(a["u"], a["h"])
            ops.gemm_nt(a["h"], sh[pre + "mlp_input.2.weight|f"], a["t2"])
            pend = (a["x1"], a["t2"], tm[:, mo + 5 * D : mo + 6 * D])
        mo = (L - ne) * 6 * D
        ops.ln_modulate_fwd(pend[0], None, None, tm[:, mo : mo + D], tm[:, mo + D : mo + 2 * D], 1, 1e-6, w["xf"], w["meanf"],
                            w["rstdf"], t=pend[1], gate=pend[2], x_out=w["xl"])
        ops.gemm_nt(w["xf"], sh["last_layer.linear.weight|f"], w["otok"], bias=self.P("last_layer.linear.bias"), M=M, N=Fo, K=D)
        ops.unpatchify(w["otok"], w["pred"], d.patch_size)
        return w["pred"]

    def _decoder_bwd(self, dpred: Tensor, first: int, nd: int, cos: Tensor, sin: Tensor, s: dict, wgrad, fold, side) -> Tensor:
        """backward of _decoder_fwd.  s: chain scratch for N tokens (dxa, dxb, dxm, da, dh, dq, dk, dv); returns the gradient of
        ws["enc_out"] (ws["denc"]); ws["dtemb"] receives the decoder's gradient of the time embedding"""
        d, w, sh = self.d, self.ws, self.sh
        B, _, _, _, _, N, M, _, Fo = self.geo
        D, Hh = d.inner_dim, d.num_heads
        rot = sum(d.rope_axes_dim)
        ne, L = first, first + nd
        tm, dtm = w["tmod"], w["dtmod"]
        inplace_qk = ops.qk_inplace_ok(w, B, N)
        Fo8 = _rup(Fo, 8)
        ops.patchify(dpred, w["dO"], d.patch_size, ops.PATCH_PPC)
        gl = self.G("last_layer.linear.weight")
        if Fo == Fo8:
            ops.gemm_tn(w["dO"], w["xf"], gl, M=Fo, N=D)
        else:
            w["scr_last"].zero_()
            ops.gemm_tn(w["dO"], w["xf"], w["scr_last"], M=Fo8, N=D)
            ops.reduce_rows_f32(w["scr_last"], gl, 1, Fo * D)
        ops.colsum(w["dO"], self.G("last_layer.linear.bias"), M, Fo)
        ops.gemm_nt(w["dO"], sh["last_layer.linear.weight|t"], s["dxm"], M=M, N=D, K=self._ko)
        mo = nd * 6 * D
        ml = (nd - 1) * 6 * D
        al = w["blk"][L - 1]
        dx, dx_alt = s["dxa"], s["dxb"]
        ops.ln_modulate_bwd_tok(s["dxm"], w["xl"], None, None, tm[:, mo : mo + D], w["meanf"], w["rstdf"], None, dx,
                                dtm[:, mo : mo + D], dtm[:, mo + D : mo + 2 * D], None, gate_t=al["t2"],
                                gate=tm[:, ml + 5 * D : ml + 6 * D], dt=al["wg"]["dt2"], dgate=dtm[:, ml + 5 * D : ml + 6 * D])

        for j in reversed(range(nd)):
            bi = ne + j
            a, pre, mo = w["blk"][bi], self.prefixes[bi], j * 6 * D
            g = a["wg"]
            wgrad(g["dt2"], a["h"], pre + "mlp_input.2.weight")
            ops.mlp_swiglu_bwd(g["dt2"], sh[pre + "mlp_input.2.weight|t"], a["xm2"], sh[pre + "mlp_input.0.weight|g"], a["u"], s["dh"], g["du"])
            wgrad(g["du"], a["xm2"], pre + "mlp_input.0.weight")
            ops.gemm_nt(g["du"], sh[pre + "mlp_input.0.weight|t"], s["dxm"])
            ops.ln_modulate_bwd_tok(s["dxm"], a["x1"], self.P(pre + "norm_2.weight"), self.P(pre + "norm_2.bias"),
                                    tm[:, mo + 3 * D : mo + 4 * D], a["mean2"], a["rstd2"], dx, dx_alt, dtm[:, mo + 3 * D : mo + 4 * D],
                                    dtm[:, mo + 4 * D : mo + 5 * D], a["dwbp"][1], gate_t=a["t1"], gate=tm[:, mo + 2 * D : mo + 3 * D],
                                    dt=g["dt1"], dgate=dtm[:, mo + 2 * D : mo + 3 * D])
            fold(a["dwbp"][1], pre + "norm_2.weight", N_PART)
            dx, dx_alt = dx_alt, dx
            wgrad(g["dt1"], a["a"], pre + "attention.proj_out.weight")
            ops.gemm_nt(g["dt1"], sh[pre + "attention.proj_out.weight|t"], s["da"])
            if inplace_qk:  # dQ, dK, dV token-major into dqkv; the QK-norm backward transforms the q / k thirds in place
                ops.attn_bwd_tok(a["q"], a["k"], a["qkv"], a["a"], s["da"], a["lse"], g["dqkv"], B, Hh, N, 64, 64**-0.5)
                _must(ops.qk_norm_rope_bwd_inplace(a["qkv"], self.P(pre + "attention.qk_norm.query_norm.scale"),
                                                    self.P(pre + "attention.qk_norm.key_norm.scale"), cos, sin, a["rrms"], g["dqkv"],
                                                    self.G(pre + "attention.qk_norm.query_norm.scale"), w["qk_part"], B, N, Hh, 64, rot))
            else:
                if ops.attn_needs_padding(N):
                    Np = a["q"].shape[2]
                    ops.copy_rows3d(s["da"], N * D, D, w["dao_d"], Np * D, D, B, N, D)  # (the pad rows of dao_d stay zero)
                    ops.attn_bwd_ex(a["q"], a["k"], a["v"], a["ao"], w["dao_d"], a["lse"], s["dq"], s["dk"], s["dv"], B, Hh, Np, Np, 64,
                                    64**-0.5, w["kb_d"])
                elif ops.v_in_place(N):
                    ops.attn_bwd_qkv(a["q"], a["k"], a["qkv"], a["a"], s["da"], a["lse"], s["dq"], s["dk"], g["dqkv"], B, Hh, N, 64,
                                     64**-0.5)
                else:
                    ops.attn_bwd(a["q"], a["k"], a["v"], a["a"], s["da"], a["lse"], s["dq"], s["dk"], s["dv"], B, Hh, N, 64, 64**-0.5)
                ops.qk_norm_rope_bwd(s["dq"], s["dk"], None if ops.v_in_place(N) else s["dv"], a["qkv"],
                                     self.P(pre + "attention.qk_norm.query_norm.scale"),
                                     self.P(pre + "attention.qk_norm.key_norm.scale"), cos, sin, a["rrms"], g["dqkv"],
                                     self.G(pre + "attention.qk_norm.query_norm.scale"), B, N, Hh, 64, rot)
            wgrad(g["dqkv"], a["xm1"], pre + "attention.qkv.weight")
            ops.gemm_nt(g["dqkv"], sh[pre + "attention.qkv.weight|t"], s["dxm"])
            nxt = {}
            if j > 0:
                mp = (j - 1) * 6 * D
                ap = w["blk"][bi - 1]
                nxt = dict(gate_t=ap["t2"], gate=tm[:, mp + 5 * D : mp + 6 * D], dt=ap["wg"]["dt2"], dgate=dtm[:, mp + 5 * D : mp + 6 * D])
            ops.ln_modulate_bwd_tok(s["dxm"], a["xin"], self.P(pre + "norm_1.weight"), self.P(pre + "norm_1.bias"), tm[:, mo : mo + D],
                                    a["mean1"], a["rstd1"], dx, dx_alt, dtm[:, mo : mo + D], dtm[:, mo + D : mo + 2 * D], a["dwbp"][0],
                                    **nxt)
            fold(a["dwbp"][0], pre + "norm_1.weight", N_PART)
            dx, dx_alt = dx_alt, dx
            getattr(wgrad, "flush", lambda: None)()
            if self.reducer is not None:
                self.reducer.ready(*self.layer_ranges[bi], extra_events=(side.record_event(),))
        # decoder patch embedding
        Fi = d.input_channels * d.patch_size**2
        gc = self.G("conv_proj_decoder.weight").view(D, Fi)
        if Fi % 8 == 0:
            ops.gemm_tn(dx, w["tokP"], gc, M=D, N=Fi)
        else:
            w["scr_conv"].zero_()
            ops.gemm_tn(dx, w["tokP"], w["scr_conv"], M=D, N=_rup(Fi, 8))
            gc.add_(w["scr_conv"][:, :Fi])
        # stacked per-token adaLN GEMM: weight / bias gradients and the gradient of its input silu(z)
        R = self.layout.tmod_rows
        gw_ = self.grads[self.layout.entries[self.layout.tmod_w0][0] :][: R * D].view(R, D)
        gb_ = self.grads[self.layout.entries[self.layout.tmod_b0][0] :][:R]
        # (825 GFLOP at B = 256: through the atomics-free tiled form where its tile divides -- 2.05 -> 1.0 ms -- else the atomic kernel)
        if w.get("tmod_slab") is None or not ops.gemm_tn_group([(dtm, w["sz"], gw_)], w["tmod_slab"]):
            ops.gemm_tn(dtm, w["sz"], gw_)
        ops.colsum(dtm, gb_, M, R)
        ops.gemm_nt(dtm, sh["@tmod|t"], w["dsz"], M=M, N=D, K=R)
        w["dtemb"].zero_()
        ops.ddt_cond_bwd(w["dsz"], w["enc_out"], w["e"], B, N, w["denc"], w["dtemb"])

        return w["denc"]


class DDTEngine(SprintEngine, PerTokenDecoder):
    _conv_name = "conv_proj_encoder.weight"

    def _make_layout(self, d: DDTDims) -> DDTLayout:  # type: ignore[override]
        return DDTLayout(d)

    def _extra_shadows(self, reg) -> None:
        d = self.d
        reg("conv_proj_decoder.weight", d.inner_dim, d.input_channels * d.patch_size**2, dgrad=False)
        reg("@tmod", self.layout.tmod_rows, d.inner_dim)

    def _src(self, name: str, shape: tuple[int, int]) -> Tensor:
        if name == "@tmod":
            off = self.layout.entries[self.layout.tmod_w0][0]
            return self.params[off : off + shape[0] * shape[1]].view(shape)
        return super()._src(name, shape)

    # ------------------------------------------------------------------ workspace
    def _alloc(self, B: int, H: int, W: int, train: bool, k: int | None = None) -> None:  # type: ignore[override]
        d, dev = self.d, self.dev
        key = (B, H, W, train)
        if key == self._ws_key:
            return
        if key in self._ws_cache:
            self.ws, self.geo = self._ws_cache[key]
            self._ws_key = key
            return
        D, E, p, L = d.inner_dim, d.embedding_dim, d.patch_size, d.depth
        gh, gw = H // p, W // p
        N = gh * gw
        if N % 64 or (N > 256 and (N % 256 or N > 2048)):
            raise NotImplementedError(f"DDT HIP path: token grid {gh}x{gw} must give a multiple of 64 tokens up to 256, or of 256 up "
                                      "to 2048")
        M, Bp, Fo, F = B * N, _rup(B, 64), p * p * d.output_channels, d.mlp_ratio * D
        bf, f32 = torch.bfloat16, torch.float32
        R = self.layout.tmod_rows

        def z(*shape, dtype=bf):
            with torch.inference_mode(False):
                return torch.zeros(*shape, device=dev, dtype=dtype)

        def zr(rows, *rest, dtype=bf):  # row buffer: zero rows pad it to a multiple of 64 for the weight-gradient GEMMs
            return z(_rup(rows, 64), *rest, dtype=dtype)[:rows]

        w: dict[str, object] = {"tokP": z(M, self._ki), "temb": z(Bp, d.frequency_embedding), "pre1": z(Bp, E), "h1": z(Bp, E),
                                "e": z(Bp, E, dtype=f32), "emb": z(Bp, E, dtype=f32), "se": z(Bp, E),
                                "mod": z(Bp, self.layout.mod_rows)}
        blk = []
        for _ in range(L):
            a = {"x0": z(M, D), "mean1": z(M, dtype=f32), "rstd1": z(M, dtype=f32), "xm1": z(M, D), "qkv": z(M, 3 * D),
                 "q": z(B, d.num_heads, N, 64), "k": z(B, d.num_heads, N, 64), "v": z(B, d.num_heads, N, 64),
                 "rrms": z(M, 2, dtype=f32), "a": z(M, D), "lse": z(B, d.num_heads, N, dtype=f32), "t1": z(M, D), "x1": z(M, D),
                 "mean2": z(M, dtype=f32), "rstd2": z(M, dtype=f32), "xm2": z(M, D), "u": ops.mlp_u_buffer(z, M, D, F, train), "h": z(M, F), "t2": z(M, D)}
            if train:
                a["wg"] = {"dt2": z(M, D), "du": z(M, 2 * F), "dt1": z(M, D), "dqkv": z(M, 3 * D)}
                a["dwb"] = z(2, B, 2, D, dtype=f32)        # encoder blocks: per-sample partials
                a["dwbp"] = z(2, N_PART, 2, D, dtype=f32)  # decoder blocks: slab partials of the per-token LayerNorm backward
            blk.append(a)
        w["blk"] = blk
        w["x"] = [z(M, D)]
        w["xdec_in"], w["enc_out"], w["sz"] = z(M, D), z(M, D), z(M, D)
        w["tmod"] = z(M, R)
        w["xl"] = z(M, D)
        w["meanf"], w["rstdf"], w["xf"] = z(M, dtype=f32), z(M, dtype=f32), z(M, D)
        w["otok"] = z(M, _rup(Fo, 8), dtype=f32)
        w["pred"] = z(B, d.output_channels, H, W, dtype=f32)
        if train:
            w["dO"] = z(M, self._ko)
            w[f"s{N}"] = {"dxa": z(M, D), "dxb": z(M, D), "dxm": z(M, D), "da": z(M, D), "dh": z(M, F),
                          "dq": z(B, d.num_heads, N, 64), "dk": z(B, d.num_heads, N, 64), "dv": z(B, d.num_heads, N, 64)}
            w["dtmod"] = z(M, R)
            if ops.wgrad_tile_ok(R, D) and M % 32 == 0 and M >= 2048:  # two token ranges of the stacked per-token adaLN weight gradient
                w["tmod_slab"] = torch.empty(2 * R * D, device=self.dev, dtype=torch.float32)
            w["dsz"], w["denc"] = z(M, D), z(M, D)
            w["dtemb"] = z(Bp, E, dtype=f32)
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

    # ------------------------------------------------------------------ forward
    def forward(self, x: Tensor, t: Tensor, y_eff: Tensor | None, train: bool = True, refresh: bool = True) -> Tensor:  # type: ignore[override]
        d = self.d
        B, C, H, W = x.shape
        assert C == d.input_channels and x.dtype == torch.float32 and x.is_cuda
        self._alloc(B, H, W, train)
        if refresh:
            self.refresh_shadows(force=train)
        w, sh = self.ws, self.sh
        _, _, _, gh, gw, N, M, Bp, Fo = self.geo
        D, L, Hh = d.inner_dim, d.depth, d.num_heads
        ne = d.encoder_depth
        cos, sin = self._rope[(gh, gw)]
        rot = sum(d.rope_axes_dim)
        self._train, self._yeff = train, y_eff
        self._stem_fwd(x, t, y_eff)
        ops.gemm_nt(w["tokP"], sh["conv_proj_decoder.weight|f"], w["xdec_in"], M=M, N=D, K=self._ki)
        self._stage_fwd(range(0, ne), w["x"][0], N, None, w["enc_out"])
        return self._decoder_fwd(ne, L - ne, cos, sin)

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
        B, H, W, gh, gw, N, M, Bp, Fo = self.geo
        D, L = d.inner_dim, d.depth
        ne = d.encoder_depth
        dfe = {kb: g.reshape(-1, D).to(torch.bfloat16).contiguous() for kb, g in (dfeats or {}).items()}
        cos, sin = self._rope[(gh, gw)]
        w["dmod32"][:B].zero_()
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

        def fold(partial: Tensor, gname: str, groups: int) -> None:
            ev = main.record_event()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                ops.reduce_rows_f32(partial, self.G(gname), groups, 2 * D, clear=True)

        denc = self._decoder_bwd(dpred, ne, L - ne, cos, sin, w[f"s{N}"], wgrad, fold, side)

        def fold_norm(partial: Tensor, gname: str) -> None:
            fold(partial, gname, B)

        dx0 = self._stage_bwd(range(0, ne), denc, N, None, dfe, wgrad, fold_norm, side)
        main.wait_stream(side)
        self._cond_bwd(dx0, extra_demb=w["dtemb"])


# ================================================================================================ joint-encoder form
from .mmdit_engine import STREAMS, JointDims, joint_rope_tables  # noqa: E402
from .sprint_joint_engine import SprintJointEngine  # noqa: E402


@dataclass
class DDTJointDims(JointDims):
    encoder_depth: int = 8
    decoder_depth: int = 4

    def __post_init__(self) -> None:
        self.embedding_dim = self.inner_dim
        super().__post_init__()
        self.depth = self.encoder_depth + self.decoder_depth

    def kinds(self) -> list[tuple[str, str]]:
        return ([(f"layers.{i}.", "J") for i in range(self.encoder_depth)]
                + [(f"decoder_layers.{i}.", "D") for i in range(self.decoder_depth)])


class DDTJointLayout:
    def __init__(self, d: DDTJointDims) -> None:
        D, p = d.inner_dim, d.patch_size
        E, F = D, d.mlp_ratio * D
        self.entries: dict[str, tuple[int, tuple[int, ...]]] = {}
        self.size = 0
        kinds = d.kinds()
        self.prefixes = [pre for pre, _ in kinds]
        enc = [pre for pre, k in kinds if k == "J"]
        dec = [pre for pre, k in kinds if k == "D"]

        def add(name: str, shape: tuple[int, ...], align: int = 64) -> None:
            self.size = _rup(self.size, align)
            self.entries[name] = (self.size, shape)
            self.size += math.prod(shape)

        for suffix, shp in (("weight", lambda r: (r, E)), ("bias", lambda r: (r,))):
            first = True
            for pre in enc:  # per-sample adaLN of the joint encoder blocks: [input 6D | context 6D] each
                for st in STREAMS:
                    add(pre + f"modulation_{st}.lin." + suffix, shp(6 * D), align=64 if first else 1)
                    first = False
        for suffix, shp in (("weight", lambda r: (r, E)), ("bias", lambda r: (r,))):
            for i, pre in enumerate(dec):  # per-token adaLN of the decoder blocks + last layer
                add(pre + "modulation.lin." + suffix, shp(6 * D), align=1 if i else 64)
            add("last_layer.adaLN_modulation.1." + suffix, shp(2 * D), align=1)
        self.mod_off = [i * 12 * D for i in range(len(enc))] + [0] * len(dec)
        self.mod_rows, self.tmod_rows = len(enc) * 12 * D, len(dec) * 6 * D + 2 * D
        self.mod_w0, self.mod_b0 = enc[0] + "modulation_input.lin.weight", enc[0] + "modulation_input.lin.bias"
        self.tmod_w0, self.tmod_b0 = dec[0] + "modulation.lin.weight", dec[0] + "modulation.lin.bias"
        add("time_embed.0.weight", (E, d.frequency_embedding))
        add("time_embed.0.bias", (E,))
        add("time_embed.2.weight", (E, E))
        add("time_embed.2.bias", (E,))
        add("conv_proj_encoder.weight", (D, d.input_channels, p, p))
        add("conv_proj_decoder.weight", (D, d.input_channels, p, p))
        add("context_embed.weight", (D, d.context_dim))
        add("last_layer.linear.weight", (p * p * d.output_channels, D))
        add("last_layer.linear.bias", (p * p * d.output_channels,))
        self.block_first = []
        for pre, kind in kinds:
            if kind == "J":
                self.block_first.append(pre + "input_norm_1.weight")
                for st in STREAMS:
                    for n in (1, 2):
                        add(pre + f"{st}_norm_{n}.weight", (D,))
                        add(pre + f"{st}_norm_{n}.bias", (D,), align=1)
                    add(pre + f"attention.qk_norm_{st}.query_norm.scale", (D,))
                    add(pre + f"attention.qk_norm_{st}.key_norm.scale", (D,), align=1)
                    add(pre + f"attention.qkv_{st}.weight", (3 * D, D))
                    add(pre + f"attention.{st}_proj_out.weight", (D, D))
                    add(pre + f"mlp_{st}.0.weight", (2 * F, D))
                    add(pre + f"mlp_{st}.2.weight", (D, F))
            else:
                self.block_first.append(pre + "norm_1.weight")
                add(pre + "norm_1.weight", (D,))
                add(pre + "norm_1.bias", (D,), align=1)
                add(pre + "norm_2.weight", (D,))
                add(pre + "norm_2.bias", (D,), align=1)
                add(pre + "attention.qk_norm.query_norm.scale", (D,))
                add(pre + "attention.qk_norm.key_norm.scale", (D,), align=1)
                add(pre + "attention.qkv.weight", (3 * D, D))
                add(pre + "attention.proj_out.weight", (D, D))
                add(pre + "mlp_input.0.weight", (2 * F, D))
                add(pre + "mlp_input.2.weight", (D, F))
        self.size = _rup(self.size, 64)

    def view(self, flat: Tensor, name: str) -> Tensor:
        off, shape = self.entries[name]
        return flat[off : off + math.prod(shape)].view(shape)


class DDTJointEngine(SprintJointEngine, PerTokenDecoder):
    """DDT(simple_ddt=False) (ddt.py:274-344 + 404-464): joint text-image encoder stage (the MMDiTBlock sequence of
    sprint_joint_engine.py; the context half of its last block feeds nothing), then the per-token-conditioned DDT decoder on the
    image rows of the 3-axis RoPE table"""

    _conv_name = "conv_proj_encoder.weight"

    def _make_layout(self, d: DDTJointDims) -> DDTJointLayout:  # type: ignore[override]
        self.kinds = d.kinds()
        return DDTJointLayout(d)

    def _extra_shadows(self, reg) -> None:
        d = self.d
        reg("context_embed.weight", d.inner_dim, d.context_dim, dgrad=False)
        reg("conv_proj_decoder.weight", d.inner_dim, d.input_channels * d.patch_size**2, dgrad=False)
        reg("@tmod", self.layout.tmod_rows, d.inner_dim)

    def _block_shadows(self, reg, pre: str) -> None:
        if dict(self.kinds)[pre] == "J":
            SprintJointEngine._block_shadows(self, reg, pre)
        else:
            super(SprintJointEngine, self)._block_shadows(reg, pre)  # DiT block names

    def _src(self, name: str, shape: tuple[int, int]) -> Tensor:
        if name == "@tmod":
            off = self.layout.entries[self.layout.tmod_w0][0]
            return self.params[off : off + shape[0] * shape[1]].view(shape)
        return super()._src(name, shape)

    def _alloc(self, B: int, H: int, W: int, train: bool, Lc: int = 0, k: int = 0) -> None:  # type: ignore[override]
        d, dev = self.d, self.dev
        key = (B, H, W, train, Lc)
        if key == self._ws_key:
            return
        if key in self._ws_cache:
            self.ws, self.geo = self._ws_cache[key]
            self._ws_key = key
            return
        D, E, p = d.inner_dim, d.embedding_dim, d.patch_size
        gh, gw = H // p, W // p
        N = gh * gw
        Tf = Lc + N
        Tpf = _rup(Tf, 256)
        if Tpf > 2048 or (B * N) % 64 or Lc < 1:
            raise NotImplementedError(f"joint DDT HIP path: context + image tokens <= 2048 (got {Lc} + {N}), batch * image tokens a "
                                      f"multiple of 64 (got {B} * {N})")
        pad = ops.attn_needs_padding(N)  # decoder attention on rows padded to a multiple of 256 with masked pad keys (_decoder_fwd)
        Nd = _rup(N, 256) if pad else N
        M, Bp, Fo, F = B * N, _rup(B, 64), p * p * d.output_channels, d.mlp_ratio * D
        bf, f32 = torch.bfloat16, torch.float32
        Hh, R = d.num_heads, self.layout.tmod_rows

        def z(*shape, dtype=bf):
            with torch.inference_mode(False):
                return torch.zeros(*shape, device=dev, dtype=dtype)

        def zr(rows, *rest, dtype=bf):  # row buffer: zero rows pad it to a multiple of 64 for the weight-gradient GEMMs
            return z(_rup(rows, 64), *rest, dtype=dtype)[:rows]

        w: dict[str, object] = {"tokP": z(M, self._ki), "temb": z(Bp, d.frequency_embedding), "pre1": z(Bp, E), "h1": z(Bp, E),
                                "e": z(Bp, E, dtype=f32), "emb": z(Bp, E, dtype=f32), "se": z(Bp, E),
                                "mod": z(Bp, self.layout.mod_rows)}
        w["ctxP"] = zr(B * Lc, _rup(d.context_dim, 64))
        w["x"] = [z(M, D)]
        w["c0"] = zr(B * Lc, D)
        w["kb_f"] = z(B, Tpf, dtype=f32)
        w["kb_f"][:, Tf:] = float("-inf")
        blk = []
        for _, kind in self.kinds:
            if kind == "J":
                per: dict[str, object] = {"ao": z(B * Tpf, D), "lse": z(B, Hh, Tpf, dtype=f32), "q": z(B, Hh, Tpf, 64),
                                          "k": z(B, Hh, Tpf, 64), "v": z(B, Hh, Tpf, 64)}
                for st, nt in (("input", N), ("context", Lc)):
                    mt = B * nt
                    a = {"x0": zr(mt, D), "mean1": zr(mt, dtype=f32), "rstd1": zr(mt, dtype=f32), "xm1": zr(mt, D), "qkv": zr(mt, 3 * D),
                         "rrms": zr(mt, 2, dtype=f32), "a": zr(mt, D), "t1": zr(mt, D), "x1": zr(mt, D), "mean2": zr(mt, dtype=f32),
                         "rstd2": zr(mt, dtype=f32), "xm2": zr(mt, D), "u": ops.mlp_u_buffer(zr, mt, D, F, train), "h": zr(mt, F), "t2": zr(mt, D)}
                    if train:
                        a["wg"] = {"dt2": zr(mt, D), "du": zr(mt, 2 * F), "dt1": zr(mt, D), "dqkv": zr(mt, 3 * D)}
                        a["dwb"] = z(2, B, 2, D, dtype=f32)
                    per[st] = a
            else:
                per = {"x0": z(M, D), "mean1": z(M, dtype=f32), "rstd1": z(M, dtype=f32), "xm1": z(M, D), "qkv": z(M, 3 * D),
                       "q": z(B, Hh, Nd, 64), "k": z(B, Hh, Nd, 64), "v": z(B, Hh, Nd, 64), "rrms": z(M, 2, dtype=f32), "a": z(M, D),
                       "lse": z(B, Hh, Nd, dtype=f32), "t1": z(M, D), "x1": z(M, D), "mean2": z(M, dtype=f32), "rstd2": z(M, dtype=f32),
                       "xm2": z(M, D), "u": ops.mlp_u_buffer(z, M, D, F, train), "h": z(M, F), "t2": z(M, D)}
                if pad:
                    per["ao"] = z(B * Nd, D)
                if train:
                    per["wg"] = {"dt2": z(M, D), "du": z(M, 2 * F), "dt1": z(M, D), "dqkv": z(M, 3 * D)}
                    per["dwbp"] = z(2, N_PART, 2, D, dtype=f32)
            blk.append(per)
        if pad:
            w["kb_d"] = z(B, Nd, dtype=f32)
            w["kb_d"][:, N:] = float("-inf")
        w["blk"] = blk
        w["xdec_in"], w["enc_out"], w["sz"] = z(M, D), z(M, D), z(M, D)
        w["tmod"] = z(M, R)
        w["xl"] = z(M, D)
        w["meanf"], w["rstdf"], w["xf"] = z(M, dtype=f32), z(M, dtype=f32), z(M, D)
        w["otok"] = z(M, _rup(Fo, 8), dtype=f32)
        w["pred"] = z(B, d.output_channels, H, W, dtype=f32)
        if train:
            w["dO"] = z(M, self._ko)
            w[f"s_x{N}"] = {"dxa": z(M, D), "dxb": z(M, D), "dxm": z(M, D), "dxm2": z(M, D), "da": z(M, D), "dh": z(M, F),
                            "dq": z(B, Hh, Nd, 64), "dk": z(B, Hh, Nd, 64), "dv": z(B, Hh, Nd, 64)}
            if pad:
                w["dao_d"] = z(B * Nd, D)
            w["s_c"] = {"dxa": zr(B * Lc, D), "dxb": zr(B * Lc, D), "dxm": zr(B * Lc, D), "da": zr(B * Lc, D), "dh": zr(B * Lc, F)}
            w["dao_f"] = z(B * Tpf, D)
            w["dq_f"], w["dk_f"], w["dv_f"] = (z(B, Hh, Tpf, 64) for _ in range(3))
            w["dtmod"] = z(M, R)
            if ops.wgrad_tile_ok(R, D) and M % 32 == 0 and M >= 2048:  # two token ranges of the stacked per-token adaLN weight gradient
                w["tmod_slab"] = torch.empty(2 * R * D, device=self.dev, dtype=torch.float32)
            w["dsz"], w["denc"] = z(M, D), z(M, D)
            w["dtemb"] = z(Bp, E, dtype=f32)
            w["dmod"] = z(Bp, self.layout.mod_rows)
            w["dmod32"] = z(Bp, self.layout.mod_rows, dtype=f32)
            w["dse"], w["demb"], w["demb16"] = z(Bp, E, dtype=f32), z(Bp, E, dtype=f32), z(Bp, E)
            w["dh1"], w["dpre1"] = z(Bp, E, dtype=f32), z(Bp, E)
            w["scr_last"], w["scr_conv"] = z(_rup(Fo, 8), D, dtype=f32), z(D, self._ki, dtype=f32)
            # scratch of the bit-reproducible (and faster) small GEMMs / column sums of the conditioning backward (engine._cond_bwd)
            w["det_scr"] = torch.empty(max(2 * self.layout.mod_rows * d.embedding_dim, 1 << 22), device=self.dev, dtype=f32)
            if ops.WgradGroups.widths_ok(D, F) and tuning.on("DL_WGRAD_GROUP"):  # grouped weight gradients (ops.WgradGroups)
                w["tn_slab"] = torch.empty(ops.WgradGroups.slab_floats(D, F), device=self.dev, dtype=f32)
            if D <= 512 and tuning.on("DL_QK_INPLACE") and not pad:  # scale-gradient partials of the in-place QK-norm backward (ops.qk_inplace_ok)
                w["qk_part"] = torch.empty(1024 * 2 * D, device=self.dev, dtype=f32)
        self.ws, self._ws_key = w, key
        self.geo = (B, H, W, gh, gw, N, M, Bp, Fo)
        if len(self._ws_cache) >= 8:
            self._ws_cache.pop(next(iter(self._ws_cache)))
        self._ws_cache[key] = (w, self.geo)
        if (Lc, gh, gw) not in self._rope:
            c, s = joint_rope_tables(Lc, gh, gw, d.rope_axes_dim, d.rope_base)
            self._rope[(Lc, gh, gw)] = (c.to(dev), s.to(dev))

    def forward(self, x: Tensor, t: Tensor, y_eff: Tensor | None = None, train: bool = True, refresh: bool = True) -> Tensor:
        d = self.d
        B, C, H, W = x.shape
        assert C == d.input_channels and x.dtype == torch.float32 and x.is_cuda and self.context is not None
        ctx, keep = self.context
        Lc = ctx.shape[1]
        self._alloc(B, H, W, train, Lc)
        if refresh:
            self.refresh_shadows(force=train)
        w, sh = self.ws, self.sh
        _, _, _, gh, gw, N, M, Bp, Fo = self.geo
        D, Cd, ne = d.inner_dim, d.context_dim, d.encoder_depth
        self._train, self._yeff, self._Lc = train, None, Lc
        self._tabs = self._rope[(Lc, gh, gw)]
        self._stem_fwd(x, t, None)
        ops.gemm_nt(w["tokP"], sh["conv_proj_decoder.weight|f"], w["xdec_in"], M=M, N=D, K=self._ki)
        w["ctxP"][:, :Cd].copy_(ctx.reshape(B * Lc, Cd))
        ops.gemm_nt(w["ctxP"], sh["context_embed.weight|f"], w["c0"], M=B * Lc, N=D, K=w["ctxP"].shape[1])
        kb = w["kb_f"]
        if keep is None:
            kb[:, :Lc].zero_()
        else:
            kb[:, :Lc].zero_().masked_fill_(~keep.to(device=kb.device, dtype=torch.bool), float("-inf"))
        self._jstage_fwd(range(0, ne), w["x"][0], w["c0"], N, None, kb, w["enc_out"], None)
        cos, sin = self._tabs
        return self._decoder_fwd(ne, d.decoder_depth, cos[Lc:], sin[Lc:])

    def feature(self, kblk: int) -> Tensor:
        assert self._train and 0 <= kblk < self.d.encoder_depth
        B, N, D = self.geo[0], self.geo[5], self.d.inner_dim
        if kblk + 1 < self.d.encoder_depth:
            return self.ws["blk"][kblk + 1]["input"]["x0"].view(B, N, D)
        return self.ws["enc_out"].view(B, N, D)

    def backward(self, dpred: Tensor, dfeats: dict[int, Tensor] | None = None) -> None:
        assert self._train and self.grads is not None
        d, w = self.d, self.ws
        B, H, W, gh, gw, N, M, Bp, Fo = self.geo
        D, Cd, ne, Lc = d.inner_dim, d.context_dim, d.encoder_depth, self._Lc
        dfe = {kb_: g.reshape(-1, D).to(torch.bfloat16).contiguous() for kb_, g in (dfeats or {}).items()}
        cos, sin = self._tabs
        w["dmod32"][:B].zero_()
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

        def fold(partial: Tensor, gname: str, groups: int) -> None:
            ev = main.record_event()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                ops.reduce_rows_f32(partial, self.G(gname), groups, 2 * D, clear=True)

        denc = self._decoder_bwd(dpred, ne, d.decoder_depth, cos[Lc:], sin[Lc:], w[f"s_x{N}"], wgrad, fold, side)
        dx0, dc0 = self._jstage_bwd(range(0, ne), denc, None, N, None, w["kb_f"], "f", dfe,
                                    (wgrad, lambda partial, gname: fold(partial, gname, B), side))
        main.wait_stream(side)
        ops.gemm_tn(dc0, w["ctxP"], self.G("context_embed.weight"), M=D, N=Cd)
        self._cond_bwd(dx0, extra_demb=w["dtemb"])
