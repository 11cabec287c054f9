"""Host-side runtime of SprintDiT in its joint text-image form (simple_dit=False; reference networks/denoisers/sprint.py:389-503,
the model of configs/train_imagenet_repa_txt_to_img_sprint.yaml):

    joint encoder blocks (N image + Lc text tokens) -> keep k of the N image tokens -> deep stage on [text ; kept image tokens]:
    joint blocks, then MMDiTSingleStreamBlocks (mmdit.py:442-532: attention and MLP in parallel on the modulated concatenation,
    one 3D-row modulation) -> restore the image canvas, fuse = Linear(2D -> D) on [restored ; encoder image output],
    fuse_context = Linear(2D -> D) on [deep context ; encoder context] -> joint decoder blocks -> last layer

Every stage reuses the launch sequences of the DiT / joint engines over ONE flat arena: per-stream gated residuals are absorbed by
the next LayerNorm-modulate kernel inside a stage and materialised at stage boundaries; the single-stream stage keeps the
concatenated latents [B, Lc + k, D] resident across its blocks (the reference's split / re-concatenate between consecutive blocks
is the identity); RoPE rows of the kept tokens come from a position index, the key-padding mask from an additive key bias.
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import torch
from torch import Tensor

from . import ops, tuning
from .engine import DiTEngine, _rup
from .mmdit_engine import STREAMS, JointDims, joint_rope_tables
from .sprint_engine import Route


@dataclass
class SprintJointDims(JointDims):
    encoder_depth: int = 2
    deep_layers_depth: int = 8
    n_single_stream_blocks: int = 0
    decoder_depth: int = 2
    drop_rate: float = 0.75

    def __post_init__(self) -> None:
        super().__post_init__()
        self.depth = self.encoder_depth + self.deep_layers_depth + self.decoder_depth

    def n_kept(self, S: int) -> int:
        return max(1, int(S * (1.0 - float(self.drop_rate))))

    def kinds(self) -> list[tuple[str, str]]:
        """(state_dict prefix, 'J' joint block | 'S' single-stream block) of every block in execution order"""
        nj = self.deep_layers_depth - self.n_single_stream_blocks
        return ([(f"layers.{i}.", "J") for i in range(self.encoder_depth)]
                + [(f"deep_layers.{i}.", "J" if i < nj else "S") for i in range(self.deep_layers_depth)]
                + [(f"decoder_layers.{i}.", "J") for i in range(self.decoder_depth)])


class SprintJointLayout:
    def __init__(self, d: "SprintJointDims | JointStackDims", sprint: bool = True) -> None:
        D, E, p = d.inner_dim, d.embedding_dim, d.patch_size
        F = d.mlp_ratio * D
        self.entries: dict[str, tuple[int, tuple[int, ...]]] = {}
        self.size = 0
        kinds = d.kinds()
        self.prefixes = [pre for pre, _ in kinds]

        def add(name: str, shape: tuple[int, ...], align: int = 64) -> None:
            self.size = _rup(self.size, align)
            self.entries[name] = (self.size, shape)
            self.size += math.prod(shape)

        def mods(suffix: str, width) -> list[str]:
            names = []
            for pre, kind in kinds:
                if kind == "J":
                    names += [(pre + f"modulation_{st}.lin.{suffix}", width(6)) for st in STREAMS]
                else:
                    names.append((pre + f"modulation.1.{suffix}", width(3)))
            return names

        # stacked adaLN matrix: [input 6D | context 6D] per joint block, 3D per single-stream block, then the last layer's 2D
        self.mod_off, off = [], 0
        for _, kind in kinds:
            self.mod_off.append(off)
            off += (12 if kind == "J" else 3) * D
        self.mod_rows = off + 2 * D
        for i, (n, shp) in enumerate(mods("weight", lambda r: (r * D, E))):
            add(n, shp, align=1 if i else 64)
            if i == 0:
                self.mod_w0 = n
        add("last_layer.adaLN_modulation.1.weight", (2 * D, E), align=1)
        for i, (n, shp) in enumerate(mods("bias", lambda r: (r * D,))):
            add(n, shp, align=1 if i else 64)
            if i == 0:
                self.mod_b0 = n
        add("last_layer.adaLN_modulation.1.bias", (2 * D,), align=1)
        add("time_embed.0.weight", (E, d.frequency_embedding))
        add("time_embed.0.bias", (E,))
        add("time_embed.2.weight", (E, E))
        add("time_embed.2.bias", (E,))
        add("conv_proj.weight", (D, d.input_channels, p, p))
        add("context_embed.weight", (D, d.context_dim))
        add("last_layer.linear.weight", (p * p * d.output_channels, D))
        add("last_layer.linear.bias", (p * p * d.output_channels,))
        if sprint:
            add("mask_token", (1, 1, D))
            add("fuse.weight", (D, 2 * D))
            add("fuse_context.weight", (D, 2 * D))
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
                self.block_first.append(pre + "norm.weight")
                add(pre + "norm.weight", (D,))
                add(pre + "norm.bias", (D,), align=1)
                add(pre + "attention.qk_norm.query_norm.scale", (D,))
                add(pre + "attention.qk_norm.key_norm.scale", (D,), align=1)
                add(pre + "attention.qkv.weight", (3 * D, D))
                add(pre + "attention.proj_out.weight", (D, D))
                add(pre + "mlp.0.weight", (2 * F, D))
                add(pre + "mlp.2.weight", (D, F))
        self.size = _rup(self.size, 64)

    def view(self, flat: Tensor, name: str) -> Tensor:
        off, shape = self.entries[name]
        return flat[off : off + math.prod(shape)].view(shape)


class SprintJointEngine(DiTEngine):
    context: tuple[Tensor, Tensor | None] | None = None
    route: Route | None = None

    def _make_layout(self, d: SprintJointDims) -> SprintJointLayout:  # type: ignore[override]
        self.kinds = d.kinds()
        return SprintJointLayout(d)

    def _extra_shadows(self, reg) -> None:
        D = self.d.inner_dim
        reg("context_embed.weight", D, self.d.context_dim, dgrad=False)
        reg("fuse.weight", D, 2 * D)
        reg("fuse_context.weight", D, 2 * D)

    def _block_shadows(self, reg, pre: str) -> None:
        d, dev = self.d, self.dev
        D, F = d.inner_dim, d.mlp_ratio * d.inner_dim
        kind = dict(self.kinds)[pre]
        names = ([(f"attention.qkv_{st}.weight", f"attention.{st}_proj_out.weight", f"mlp_{st}.0.weight", f"mlp_{st}.2.weight")
                  for st in STREAMS] if kind == "J" else [("attention.qkv.weight", "attention.proj_out.weight", "mlp.0.weight", "mlp.2.weight")])
        for qkv, proj, m0, m2 in names:
            reg(pre + qkv, 3 * D, D)
            reg(pre + proj, D, D)
            reg(pre + m0, 2 * F, D)
            self.sh[pre + m0 + "|g"] = torch.zeros(2 * F, D, device=dev, dtype=torch.bfloat16)
            reg(pre + m2, D, F)

    # ------------------------------------------------------------------ workspace
    def _alloc(self, B: int, H: int, W: int, train: bool, Lc: int = 0, k: int = 0) -> None:  # type: ignore[override]
        d, dev = self.d, self.dev
        key = (B, H, W, train, Lc, k)
        if key == self._ws_key:
            return
        if key in self._ws_cache:
            self.ws, self.geo = self._ws_cache[key]
            self._ws_key = key
            return
        D, E, p = d.inner_dim, d.embedding_dim, d.patch_size
        gh, gw = H // p, W // p
        N = gh * gw
        Tf, Td = Lc + N, Lc + k
        Tpf, Tpd = _rup(Tf, 256), _rup(Td, 256)
        if Tpf > 2048 or (B * N) % 64 or Lc < 1:
            raise NotImplementedError(f"joint SprintDiT HIP path: context + image tokens <= 2048 (got {Lc} + {N}), batch * tokens of "
                                      f"each stream a multiple of 64 (kept {k})")
        M, Bp, Fo, F = B * N, _rup(B, 64), p * p * d.output_channels, d.mlp_ratio * D
        bf, f32 = torch.bfloat16, torch.float32
        Hh = d.num_heads

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
        w["kb_f"], w["kb_d"] = z(B, Tpf, dtype=f32), z(B, Tpd, dtype=f32)
        w["kb_f"][:, Tf:] = float("-inf")
        w["kb_d"][:, Td:] = float("-inf")
        ne, nd = d.encoder_depth, d.deep_layers_depth
        blk = []
        for bi, (_, kind) in enumerate(self.kinds):
            deep = ne <= bi < ne + nd
            nx, Tp = (k, Tpd) if deep else (N, Tpf)
            per: dict[str, object] = {"ao": z(B * Tp, D), "lse": z(B, Hh, Tp, dtype=f32), "q": z(B, Hh, Tp, 64), "k": z(B, Hh, Tp, 64),
                                      "v": z(B, Hh, Tp, 64)}
            if kind == "J":
                for st, nt in (("input", nx), ("context", Lc)):
                    mt = B * nt
                    a = {"x0": zr(mt, D), "mean1": zr(mt, dtype=f32), "rstd1": zr(mt, dtype=f32), "xm1": zr(mt, D), "qkv": zr(mt, 3 * D),
                         "rrms": zr(mt, 2, dtype=f32), "a": zr(mt, D), "t1": zr(mt, D), "x1": zr(mt, D), "mean2": zr(mt, dtype=f32),
                         "rstd2": zr(mt, dtype=f32), "xm2": zr(mt, D), "u": ops.mlp_u_buffer(zr, mt, D, F, train), "h": zr(mt, F), "t2": zr(mt, D)}
                    if train:
                        a["wg"] = {"dt2": zr(mt, D), "du": zr(mt, 2 * F), "dt1": zr(mt, D), "dqkv": zr(mt, 3 * D)}
                        a["dwb"] = z(2, B, 2, D, dtype=f32)
                    per[st] = a
            else:
                mt = B * Td
                per.update({"x0": zr(mt, D), "mean": zr(mt, dtype=f32), "rstd": zr(mt, dtype=f32), "m": zr(mt, D), "qkv": zr(mt, 3 * D),
                            "rrms": zr(mt, 2, dtype=f32), "a": zr(mt, D), "ta": zr(mt, D), "u": ops.mlp_u_buffer(zr, mt, D, F, train), "h": zr(mt, F), "t": zr(mt, D)})
                if train:
                    per["wg"] = {"dt": zr(mt, D), "du": zr(mt, 2 * F), "dqkv": zr(mt, 3 * D)}
                    per["dwb"] = z(1, B, 2, D, dtype=f32)
            blk.append(per)
        w["blk"] = blk
        w["cat"], w["ccat"] = z(M, 2 * D), zr(B * Lc, 2 * D)   # [restored | encoder image], [deep context | encoder context]
        w["c_enc"] = zr(B * Lc, D)
        w["xd0"], w["xd_out"] = z(B * k, D), z(B * k, D)
        w["xj"], w["cj"] = z(B * k, D), zr(B * Lc, D)           # outputs of the deep joint sub-stage
        w["lat0"], w["lat_out"] = zr(B * Td, D), zr(B * Td, D)
        w["xfuse"], w["cfuse"], w["xdec"] = z(M, D), zr(B * Lc, D), z(M, D)
        w["meanf"], w["rstdf"], w["xf"] = z(M, dtype=f32), z(M, dtype=f32), z(M, D)
        w["otok"] = z(M, _rup(Fo, 8), dtype=f32)
        w["pred"] = z(B, d.output_channels, H, W, dtype=f32)
        if train:
            w["dO"] = z(M, self._ko)
            for name, mt in ((f"x{N}", M), (f"x{k}", B * k), ("c", B * Lc), ("l", B * Td)):
                w["s_" + name] = {"dxa": zr(mt, D), "dxb": zr(mt, D), "dxm": zr(mt, D), "dxm2": zr(mt, D), "da": zr(mt, D), "dh": zr(mt, F)}
            for tag, Tp in (("f", Tpf), ("d", Tpd)):
                w["dao_" + tag] = z(B * Tp, D)
                w["dq_" + tag], w["dk_" + tag], w["dv_" + tag] = (z(B, Hh, Tp, 64) for _ in range(3))
            w["dleft"], w["dright"], w["dcl"], w["dcr"] = z(M, D), z(M, D), zr(B * Lc, D), zr(B * Lc, D)
            w["dxd"], w["dlat"], w["dxj"], w["dcj"] = z(B * k, D), zr(B * Td, D), z(B * k, D), zr(B * Lc, D)
            w["dxdec"] = z(M, D)
            w["dmod"] = z(Bp, self.layout.mod_rows)
            w["dmod32"] = z(Bp, self.layout.mod_rows, dtype=f32)
            w["dse"], w["demb"], w["demb16"] = z(Bp, E, dtype=f32), z(Bp, E, dtype=f32), z(Bp, E)
            w["dh1"], w["dpre1"] = z(Bp, E, dtype=f32), z(Bp, E)
            w["scr_last"], w["scr_conv"] = z(_rup(Fo, 8), D, dtype=f32), z(D, self._ki, dtype=f32)
            # scratch of the bit-reproducible (and faster) small GEMMs / column sums of the conditioning backward (engine._cond_bwd)
            w["det_scr"] = torch.empty(max(2 * self.layout.mod_rows * d.embedding_dim, 1 << 22), device=self.dev, dtype=f32)
            if ops.WgradGroups.widths_ok(D, F) and tuning.on("DL_WGRAD_GROUP"):  # grouped weight gradients (ops.WgradGroups)
                w["tn_slab"] = torch.empty(ops.WgradGroups.slab_floats(D, F), device=self.dev, dtype=f32)
        self.ws, self._ws_key = w, key
        self.geo = (B, H, W, gh, gw, N, M, Bp, Fo)
        if len(self._ws_cache) >= 8:
            self._ws_cache.pop(next(iter(self._ws_cache)))
        self._ws_cache[key] = (w, self.geo)
        if (Lc, gh, gw) not in self._rope:
            c, s = joint_rope_tables(Lc, gh, gw, d.rope_axes_dim, d.rope_base)
            self._rope[(Lc, gh, gw)] = (c.to(dev), s.to(dev))

    # ------------------------------------------------------------------ joint stage
    def _jstage_fwd(self, blocks, x_in: Tensor, c_in: Tensor, nx: int, pos: Tensor | None, kb: Tensor, x_out: Tensor,
                    c_out: Tensor | None) -> None:
        """MMDiTBlocks `blocks` on (image stream x_in [B*nx, D], context stream c_in [B*Lc, D]); outputs materialised into x_out /
        c_out (rows may be strided); c_out None: the context half of the last block feeds nothing and is skipped"""
        d, w, sh = self.d, self.ws, self.sh
        B = self.geo[0]
        Lc = self._Lc
        D, Hh = d.inner_dim, d.num_heads
        cos, sin = self._tabs
        rot = sum(d.rope_axes_dim)
        mod = w["mod"]
        Tp = kb.shape[1]
        streams = (("input", nx, Lc, 0, pos), ("context", Lc, 0, 6 * D, None))
        cur = {"input": x_in, "context": c_in}
        pend: dict[str, tuple | None] = {"input": None, "context": None}
        blocks = list(blocks)
        for bi in blocks:
            per, pre, base = w["blk"][bi], self.prefixes[bi], self.layout.mod_off[bi]
            for st, nt, off, mc, ps in streams:
                a, mo = per[st], base + mc
                n1w, n1b = self.P(pre + f"{st}_norm_1.weight"), self.P(pre + f"{st}_norm_1.bias")
                if pend[st] is None:
                    xcur = cur[st]
                    ops.ln_modulate_fwd(xcur, n1w, n1b, mod[:, mo : mo + D], mod[:, mo + D : mo + 2 * D], nt, 1e-5, a["xm1"],
                                        a["mean1"], a["rstd1"])
                else:
                    xcur, pd = a["x0"], pend[st]
                    ops.ln_modulate_fwd(pd[0], n1w, n1b, mod[:, mo : mo + D], mod[:, mo + D : mo + 2 * D], nt, 1e-5, a["xm1"],
                                        a["mean1"], a["rstd1"], t=pd[1], gate=pd[2], x_out=xcur)
                a["xin"] = xcur
                ops.gemm_nt(a["xm1"], sh[pre + f"attention.qkv_{st}.weight|f"], a["qkv"])
                ops.qk_norm_rope_fwd(a["qkv"], self.P(pre + f"attention.qk_norm_{st}.query_norm.scale"),
                                     self.P(pre + f"attention.qk_norm_{st}.key_norm.scale"), cos[off:], sin[off:], per["q"], per["k"],
                                     per["v"], a["rrms"], B, nt, Hh, 64, rot, pos=ps, n_off=off)
            ops.attn_fwd_ex(per["q"], per["k"], per["v"], per["ao"], per["lse"], B, Hh, Tp, Tp, 64, 64**-0.5, kb)
            for st, nt, off, mc, ps in streams:
                if st == "context" and c_out is None and bi == blocks[-1]:
                    continue
                a, mo = per[st], base + mc
                ops.copy_rows3d(per["ao"][off:], Tp * D, D, a["a"], nt * D, D, B, nt, D)
                ops.gemm_nt(a["a"], sh[pre + f"attention.{st}_proj_out.weight|f"], a["t1"])
                ops.ln_modulate_fwd(a["xin"], self.P(pre + f"{st}_norm_2.weight"), self.P(pre + f"{st}_norm_2.bias"),
                                    mod[:, mo + 3 * D : mo + 4 * D], mod[:, mo + 4 * D : mo + 5 * D], nt, 1e-5, a["xm2"], a["mean2"],
                                    a["rstd2"], t=a["t1"], gate=mod[:, mo + 2 * D : mo + 3 * D], x_out=a["x1"])
                if not ops.gemm_nt_swiglu(a["xm2"], sh[pre + f"mlp_{st}.0.weight|g"], a["u"] if self._train else None, a["h"]):
                    ops.gemm_nt(a["xm2"], sh[pre + f"mlp_{st}.0.weight|f"], a["u"])
                    ops.swiglu_fwd(a["u"], a["h"])
                ops.gemm_nt(a["h"], sh[pre + f"mlp_{st}.2.weight|f"], a["t2"])
                pend[st] = (a["x1"], a["t2"], mod[:, mo + 5 * D : mo + 6 * D])
        ops.gated_residual_fwd(pend["input"][0], pend["input"][1], pend["input"][2], nx, x_out)
        if c_out is not None:
            ops.gated_residual_fwd(pend["context"][0], pend["context"][1], pend["context"][2], Lc, c_out)

    def _jstage_bwd(self, blocks, dx_x: Tensor, dx_c: Tensor | None, nx: int, pos: Tensor | None, kb: Tensor, tag: str, dfe, hooks):
        """backward of _jstage_fwd.  dx_x / dx_c: gradients at the stage outputs (contiguous, overwritten; dx_c None: the context output
        was not produced).  Returns the gradients at the stage inputs (image, context)."""
        d, w, sh = self.d, self.ws, self.sh
        wgrad, fold_norm, side = hooks
        B = self.geo[0]
        Lc = self._Lc
        D, Hh = d.inner_dim, d.num_heads
        cos, sin = self._tabs
        rot = sum(d.rope_axes_dim)
        mod, dmod = w["mod"], w["dmod32"]
        Tp = kb.shape[1]
        streams = (("input", nx, Lc, 0, pos), ("context", Lc, 0, 6 * D, None))
        scr = {"input": w[f"s_x{nx}"], "context": w["s_c"]}
        dao, dq, dk, dv = w["dao_" + tag], w["dq_" + tag], w["dk_" + tag], w["dv_" + tag]
        blocks = list(blocks)
        last = blocks[-1]
        dx: dict[str, Tensor | None] = {"input": dx_x, "context": dx_c}

        def other(s, cur):
            return s["dxb"] if (cur is not None and cur.data_ptr() == s["dxa"].data_ptr()) else s["dxa"]

        if last in dfe:
            ops.add_bf16(dx_x, dfe[last], dx_x)
        for st, nt, off, mc, ps in streams:  # backward of the materialised last residual of each stream
            if dx[st] is None:
                continue
            a, mo = w["blk"][last][st], self.layout.mod_off[last] + mc
            ops.gate_bwd(dx[st], a["t2"], mod[:, mo + 5 * D : mo + 6 * D], nt, a["wg"]["dt2"], dmod[:, mo + 5 * D : mo + 6 * D])
        skip_ctx_last = dx_c is None
        for j in reversed(range(len(blocks))):
            bi = blocks[j]
            per, pre, base = w["blk"][bi], self.prefixes[bi], self.layout.mod_off[bi]
            dao.zero_()
            for st, nt, off, mc, ps in streams:
                if st == "context" and skip_ctx_last and bi == last:
                    continue
                a, mo, s, g = per[st], base + mc, scr[st], per[st]["wg"]
                wgrad(g["dt2"], a["h"], pre + f"mlp_{st}.2.weight")
                ops.mlp_swiglu_bwd(g["dt2"], sh[pre + f"mlp_{st}.2.weight|t"], a["xm2"], sh[pre + f"mlp_{st}.0.weight|g"], a["u"], s["dh"], g["du"])
                wgrad(g["du"], a["xm2"], pre + f"mlp_{st}.0.weight")
                ops.gemm_nt(g["du"], sh[pre + f"mlp_{st}.0.weight|t"], s["dxm"])
                nx_ = other(s, dx[st])
                ops.ln_modulate_bwd(s["dxm"], a["x1"], self.P(pre + f"{st}_norm_2.weight"), self.P(pre + f"{st}_norm_2.bias"),
                                    mod[:, mo + 3 * D : mo + 4 * D], nt, a["mean2"], a["rstd2"], dx[st], nx_,
                                    dmod[:, mo + 3 * D : mo + 4 * D], dmod[:, mo + 4 * D : mo + 5 * D], a["dwb"][1], gate_t=a["t1"],
                                    gate=mod[:, mo + 2 * D : mo + 3 * D], dt=g["dt1"], dgate=dmod[:, mo + 2 * D : mo + 3 * D])
                fold_norm(a["dwb"][1], pre + f"{st}_norm_2.weight")
                dx[st] = nx_
                wgrad(g["dt1"], a["a"], pre + f"attention.{st}_proj_out.weight")
                ops.gemm_nt(g["dt1"], sh[pre + f"attention.{st}_proj_out.weight|t"], s["da"])
                ops.copy_rows3d(s["da"], nt * D, D, dao[off:], Tp * D, D, B, nt, D)
            ops.attn_bwd_ex(per["q"], per["k"], per["v"], per["ao"], dao, per["lse"], dq, dk, dv, B, Hh, Tp, Tp, 64, 64**-0.5, kb)
            for st, nt, off, mc, ps in streams:
                a, mo, s, g = per[st], base + mc, scr[st], per[st]["wg"]
                ops.qk_norm_rope_bwd(dq, dk, dv, a["qkv"], self.P(pre + f"attention.qk_norm_{st}.query_norm.scale"),
                                     self.P(pre + f"attention.qk_norm_{st}.key_norm.scale"), cos[off:], sin[off:], a["rrms"], g["dqkv"],
                                     self.G(pre + f"attention.qk_norm_{st}.query_norm.scale"), B, nt, Hh, 64, rot, pos=ps, n_off=off)
                wgrad(g["dqkv"], a["xm1"], pre + f"attention.qkv_{st}.weight")
                ops.gemm_nt(g["dqkv"], sh[pre + f"attention.qkv_{st}.weight|t"], s["dxm"])
                nxt = {}
                if j > 0:
                    bp = blocks[j - 1]
                    if st == "input" and bp in dfe and dx[st] is not None:
                        ops.add_bf16(dx[st], dfe[bp], dx[st])
                    mp = self.layout.mod_off[bp] + mc
                    ap = w["blk"][bp][st]
                    nxt = dict(gate_t=ap["t2"], gate=mod[:, mp + 5 * D : mp + 6 * D], dt=ap["wg"]["dt2"],
                               dgate=dmod[:, mp + 5 * D : mp + 6 * D])
                nx_ = other(s, dx[st])
                ops.ln_modulate_bwd(s["dxm"], a["xin"], self.P(pre + f"{st}_norm_1.weight"), self.P(pre + f"{st}_norm_1.bias"),
                                    mod[:, mo : mo + D], nt, a["mean1"], a["rstd1"], dx[st], nx_, dmod[:, mo : mo + D],
                                    dmod[:, mo + D : mo + 2 * D], a["dwb"][0], **nxt)
                fold_norm(a["dwb"][0], pre + f"{st}_norm_1.weight")
                dx[st] = nx_
            getattr(wgrad, "flush", lambda: None)()  # (grouped weight gradients: whatever of this block is still pending)
            if self.reducer is not None:
                self.reducer.ready(*self.layer_ranges[bi], extra_events=(side.record_event(),))
        return dx["input"], dx["context"]

    # ------------------------------------------------------------------ single-stream stage
    def _sstage_fwd(self, blocks, lat_in: Tensor, T: int, pos: Tensor, kb: Tensor, out: Tensor) -> None:
        """MMDiTSingleStreamBlocks on the concatenated latents lat_in [B*T, D] (context rows first); materialised output -> out"""
        d, w, sh = self.d, self.ws, self.sh
        B = self.geo[0]
        D, Hh = d.inner_dim, d.num_heads
        cos, sin = self._tabs
        rot = sum(d.rope_axes_dim)
        mod = w["mod"]
        Tp = kb.shape[1]
        pend = None
        for bi in blocks:
            a, pre, mo = w["blk"][bi], self.prefixes[bi], self.layout.mod_off[bi]
            nw, nb = self.P(pre + "norm.weight"), self.P(pre + "norm.bias")
            if pend is None:
                xcur = lat_in
                ops.ln_modulate_fwd(xcur, nw, nb, mod[:, mo : mo + D], mod[:, mo + D : mo + 2 * D], T, 1e-5, a["m"], a["mean"], a["rstd"])
            else:
                xcur = a["x0"]
                ops.ln_modulate_fwd(pend[0], nw, nb, mod[:, mo : mo + D], mod[:, mo + D : mo + 2 * D], T, 1e-5, a["m"], a["mean"],
                                    a["rstd"], t=pend[1], gate=pend[2], x_out=xcur)
            a["xin"] = xcur
            ops.gemm_nt(a["m"], sh[pre + "attention.qkv.weight|f"], a["qkv"])
            ops.qk_norm_rope_fwd(a["qkv"], self.P(pre + "attention.qk_norm.query_norm.scale"),
                                 self.P(pre + "attention.qk_norm.key_norm.scale"), cos, sin, a["q"], a["k"], a["v"], a["rrms"], B, T, Hh,
                                 64, rot, pos=pos, n_off=0)
            ops.attn_fwd_ex(a["q"], a["k"], a["v"], a["ao"], a["lse"], B, Hh, Tp, Tp, 64, 64**-0.5, kb)
            ops.copy_rows3d(a["ao"], Tp * D, D, a["a"], T * D, D, B, T, D)
            ops.gemm_nt(a["a"], sh[pre + "attention.proj_out.weight|f"], a["ta"])
            if not ops.gemm_nt_swiglu(a["m"], sh[pre + "mlp.0.weight|g"], a["u"] if self._train else None, a["h"]):
                ops.gemm_nt(a["m"], sh[pre + "mlp.0.weight|f"], a["u"])
                ops.swiglu_fwd(a["u"], a["h"])
            ops.gemm_nt(a["h"], sh[pre + "mlp.2.weight|f"], a["t"], resid=a["ta"])  # attention + MLP (mmdit.py:524-527)
            pend = (xcur, a["t"], mod[:, mo + 2 * D : mo + 3 * D])
        ops.gated_residual_fwd(pend[0], pend[1], pend[2], T, out)

    def _sstage_bwd(self, blocks, dlat: Tensor, T: int, pos: Tensor, kb: Tensor, hooks) -> Tensor:
        d, w, sh = self.d, self.ws, self.sh
        wgrad, fold_norm, side = hooks
        B = self.geo[0]
        D, Hh = d.inner_dim, d.num_heads
        cos, sin = self._tabs
        rot = sum(d.rope_axes_dim)
        mod, dmod = w["mod"], w["dmod32"]
        Tp = kb.shape[1]
        s = w["s_l"]
        dao, dq, dk, dv = w["dao_d"], w["dq_d"], w["dk_d"], w["dv_d"]
        blocks = list(blocks)
        last = blocks[-1]
        ml = self.layout.mod_off[last]
        ops.gate_bwd(dlat, w["blk"][last]["t"], mod[:, ml + 2 * D : ml + 3 * D], T, w["blk"][last]["wg"]["dt"],
                     dmod[:, ml + 2 * D : ml + 3 * D])
        dx = dlat
        for j in reversed(range(len(blocks))):
            bi = blocks[j]
            a, pre, mo = w["blk"][bi], self.prefixes[bi], self.layout.mod_off[bi]
            g = a["wg"]
            wgrad(g["dt"], a["h"], pre + "mlp.2.weight")
            wgrad(g["dt"], a["a"], pre + "attention.proj_out.weight")
            ops.mlp_swiglu_bwd(g["dt"], sh[pre + "mlp.2.weight|t"], a["m"], sh[pre + "mlp.0.weight|g"], a["u"], s["dh"], g["du"])
            wgrad(g["du"], a["m"], pre + "mlp.0.weight")
            ops.gemm_nt(g["du"], sh[pre + "mlp.0.weight|t"], s["dxm"])
            ops.gemm_nt(g["dt"], sh[pre + "attention.proj_out.weight|t"], s["da"])
            dao.zero_()
            ops.copy_rows3d(s["da"], T * D, D, dao, Tp * D, D, B, T, D)
            ops.attn_bwd_ex(a["q"], a["k"], a["v"], a["ao"], dao, a["lse"], dq, dk, dv, B, Hh, Tp, Tp, 64, 64**-0.5, kb)
            ops.qk_norm_rope_bwd(dq, dk, dv, a["qkv"], self.P(pre + "attention.qk_norm.query_norm.scale"),
                                 self.P(pre + "attention.qk_norm.key_norm.scale"), cos, sin, a["rrms"], g["dqkv"],
                                 self.G(pre + "attention.qk_norm.query_norm.scale"), B, T, Hh, 64, rot, pos=pos, n_off=0)
            wgrad(g["dqkv"], a["m"], pre + "attention.qkv.weight")
            ops.gemm_nt(g["dqkv"], sh[pre + "attention.qkv.weight|t"], s["dxm2"], resid=s["dxm"])
            nxt = {}
            if j > 0:
                bp = blocks[j - 1]
                mp = self.layout.mod_off[bp]
                nxt = dict(gate_t=w["blk"][bp]["t"], gate=mod[:, mp + 2 * D : mp + 3 * D], dt=w["blk"][bp]["wg"]["dt"],
                           dgate=dmod[:, mp + 2 * D : mp + 3 * D])
            nx_ = s["dxb"] if dx.data_ptr() == s["dxa"].data_ptr() else s["dxa"]
            ops.ln_modulate_bwd(s["dxm2"], a["xin"], self.P(pre + "norm.weight"), self.P(pre + "norm.bias"), mod[:, mo : mo + D], T,
                                a["mean"], a["rstd"], dx, nx_, dmod[:, mo : mo + D], dmod[:, mo + D : mo + 2 * D], a["dwb"][0], **nxt)
            fold_norm(a["dwb"][0], pre + "norm.weight")
            dx = nx_
            getattr(wgrad, "flush", lambda: None)()  # (grouped weight gradients: whatever of this block is still pending)
            if self.reducer is not None:
                self.reducer.ready(*self.layer_ranges[bi], extra_events=(side.record_event(),))
        return dx

    # ------------------------------------------------------------------ forward
    def forward(self, x: Tensor, t: Tensor, y_eff: Tensor | None = None, train: bool = True, refresh: bool = True) -> Tensor:
        d = self.d
        B, C, H, W = x.shape
        assert C == d.input_channels and x.dtype == torch.float32 and x.is_cuda and self.context is not None and self.route is not None
        ctx, keep = self.context
        route = self.route
        Lc, k = ctx.shape[1], route.k
        self._alloc(B, H, W, train, Lc, k)
        if refresh:
            self.refresh_shadows(force=train)
        w, sh = self.ws, self.sh
        _, _, _, gh, gw, N, M, Bp, Fo = self.geo
        D, L, Cd = d.inner_dim, d.depth, d.context_dim
        ne, nd, ns = d.encoder_depth, d.deep_layers_depth, d.n_single_stream_blocks
        nj = nd - ns
        Td = Lc + k
        self._train, self._yeff, self._Lc, self._route = train, None, Lc, route
        self._tabs = self._rope[(Lc, gh, gw)]
        mod = self._stem_fwd(x, t, None)
        w["ctxP"][:, :Cd].copy_(ctx.reshape(B * Lc, Cd))
        ops.gemm_nt(w["ctxP"], sh["context_embed.weight|f"], w["c0"], M=B * Lc, N=D, K=w["ctxP"].shape[1])
        for kb in (w["kb_f"], w["kb_d"]):
            if keep is None:
                kb[:, :Lc].zero_()
            else:
                kb[:, :Lc].zero_().masked_fill_(~keep.to(device=kb.device, dtype=torch.bool), float("-inf"))
        cat, ccat = w["cat"], w["ccat"]
        # encoder: image output -> right half of cat, context output -> c_enc (and the right half of ccat)
        self._jstage_fwd(range(0, ne), w["x"][0], w["c0"], N, None, w["kb_f"], cat[:, D:], w["c_enc"])
        ops.copy_rows3d(w["c_enc"], 0, D, ccat[:, D:], 0, 2 * D, 1, B * Lc, D)
        mask = self.P("mask_token").view(D)
        if route.skip_deep:
            ops.restore_tokens(w["xd_out"], route.inv, mask, cat[:, :D], B, N, k, D)
            ops.copy_rows3d(w["c_enc"], 0, D, ccat, 0, 2 * D, 1, B * Lc, D)  # the deep layers did not touch the context
        else:
            ops.gather_tokens(cat[:, D:], route.idx, w["xd0"], B, N, k, D)
            pos_img = route.idx.view(-1)
            xs, cs = w["xd0"], w["c_enc"]
            if nj:
                if ns:
                    self._jstage_fwd(range(ne, ne + nj), xs, cs, k, pos_img, w["kb_d"], w["xj"], w["cj"])
                    xs, cs = w["xj"], w["cj"]
                else:
                    self._jstage_fwd(range(ne, ne + nj), xs, cs, k, pos_img, w["kb_d"], w["xd_out"], ccat[:, :D])
            if ns:
                lat0 = w["lat0"]
                ops.copy_rows3d(cs, Lc * D, D, lat0, Td * D, D, B, Lc, D)
                ops.copy_rows3d(xs, k * D, D, lat0[Lc:], Td * D, D, B, k, D)
                self._sstage_fwd(range(ne + nj, ne + nd), lat0, Td, route.pos_lat, w["kb_d"], w["lat_out"])
                ops.copy_rows3d(w["lat_out"][Lc:], Td * D, D, w["xd_out"], k * D, D, B, k, D)
                ops.copy_rows3d(w["lat_out"], Td * D, D, ccat, Lc * 2 * D, 2 * D, B, Lc, D)
            ops.restore_tokens(w["xd_out"], route.inv, mask, cat[:, :D], B, N, k, D)
        ops.gemm_nt(cat, sh["fuse.weight|f"], w["xfuse"], M=M, N=D, K=2 * D)
        ops.gemm_nt(ccat, sh["fuse_context.weight|f"], w["cfuse"], M=B * Lc, N=D, K=2 * D)
        self._jstage_fwd(range(ne + nd, L), w["xfuse"], w["cfuse"], N, None, w["kb_f"], w["xdec"], None)
        mo = self.layout.mod_rows - 2 * D
        ops.ln_modulate_fwd(w["xdec"], None, None, mod[:, mo : mo + D], mod[:, mo + D : mo + 2 * D], N, 1e-6, w["xf"], w["meanf"],
                            w["rstdf"])
        ops.gemm_nt(w["xf"], sh["last_layer.linear.weight|f"], w["otok"], bias=self.P("last_layer.linear.bias"), M=M, N=Fo, K=D)
        ops.unpatchify(w["otok"], w["pred"], d.patch_size)
        return w["pred"]

    def feature(self, kblk: int) -> Tensor:
        """image-stream output of encoder block k (``layers[k]``) of the last train-mode forward"""
        assert self._train and 0 <= kblk < self.d.encoder_depth
        B, N, D = self.geo[0], self.geo[5], self.d.inner_dim
        if kblk + 1 < self.d.encoder_depth:
            return self.ws["blk"][kblk + 1]["input"]["x0"].view(B, N, D)
        return self.ws["cat"].view(B, N, 2 * D)[:, :, D:]

    # ------------------------------------------------------------------ backward
    def backward(self, dpred: Tensor, dfeats: dict[int, Tensor] | None = None) -> None:
        assert self._train and self.grads is not None
        d, w, sh = self.d, self.ws, self.sh
        B, H, W, gh, gw, N, M, Bp, Fo = self.geo
        D, L, Cd = d.inner_dim, d.depth, d.context_dim
        ne, nd, ns = d.encoder_depth, d.deep_layers_depth, d.n_single_stream_blocks
        nj = nd - ns
        route, Lc = self._route, self._Lc
        k = route.k
        Td = Lc + k
        dfe = {kb_: g.reshape(-1, D).to(torch.bfloat16).contiguous() for kb_, g in (dfeats or {}).items()}
        mod, dmod = w["mod"], w["dmod32"]
        dmod[:B].zero_()
        Fo8 = _rup(Fo, 8)
        sN = w[f"s_x{N}"]

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
        mo = self.layout.mod_rows - 2 * D
        ops.ln_modulate_bwd(sN["dxm"], w["xdec"], None, None, mod[:, mo : mo + D], N, w["meanf"], w["rstdf"], None, w["dxdec"],
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

        # the linears of a block (per token stream) as ONE atomics-free launch (ops.WgradGroups) where the 384 x 192 tile divides
        # them; the stages call wgrad.flush() at every block end
        groups = ops.WgradGroups(w["tn_slab"], on_side, max_wgs=side_wgs) if w.get("tn_slab") is not None else None

        def wgrad(x_grad: Tensor, x_in: Tensor, gname: str) -> None:
            if groups is not None:
                groups.add(x_grad, x_in, self.G(gname))
            else:
                on_side(lambda: ops.gemm_tn(x_grad, x_in, self.G(gname), max_wgs=side_wgs))

        wgrad.flush = groups.flush if groups is not None else (lambda: None)

        def fold_norm(partial: Tensor, gname: str) -> None:
            ev = main.record_event()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                ops.reduce_rows_f32(partial, self.G(gname), B, 2 * D, clear=True)

        hooks = (wgrad, fold_norm, side)
        dxf, dcf = self._jstage_bwd(range(ne + nd, L), w["dxdec"], None, N, None, w["kb_f"], "f", {}, hooks)
        # fuse / fuse_context (main stream: dxf / dcf are chain scratch that the next stages reuse)
        ops.gemm_tn(dxf, w["cat"], self.G("fuse.weight"))
        ops.gemm_tn(dcf, w["ccat"], self.G("fuse_context.weight"))
        wt, wc = sh["fuse.weight|t"], sh["fuse_context.weight|t"]
        ops.gemm_nt(dxf, wt[:D], w["dleft"], M=M, N=D, K=D)
        ops.gemm_nt(dxf, wt[D:], w["dright"], M=M, N=D, K=D)
        ops.gemm_nt(dcf, wc[:D], w["dcl"], M=B * Lc, N=D, K=D)
        ops.gemm_nt(dcf, wc[D:], w["dcr"], M=B * Lc, N=D, K=D)
        ops.masked_colsum(w["dleft"], route.inv.view(-1), self.G("mask_token").view(D), M, D)
        if route.skip_deep:
            ops.add_bf16(w["dcr"], w["dcl"], w["dcr"])
            if self.reducer is not None:
                for bi in reversed(range(ne, ne + nd)):
                    self.reducer.ready(*self.layer_ranges[bi])
        else:
            pos_img = route.idx.view(-1)
            ops.gather_tokens(w["dleft"], route.idx, w["dxd"], B, N, k, D, keep=route.keep)
            gx, gc = w["dxd"], w["dcl"]
            if ns:
                dlat = w["dlat"]
                ops.copy_rows3d(gc, Lc * D, D, dlat, Td * D, D, B, Lc, D)
                ops.copy_rows3d(gx, k * D, D, dlat[Lc:], Td * D, D, B, k, D)
                dl = self._sstage_bwd(range(ne + nj, ne + nd), dlat, Td, route.pos_lat, w["kb_d"], hooks)
                ops.copy_rows3d(dl[Lc:], Td * D, D, w["dxj"], k * D, D, B, k, D)
                ops.copy_rows3d(dl, Td * D, D, w["dcj"], Lc * D, D, B, Lc, D)
                gx, gc = w["dxj"], w["dcj"]
            if nj:
                gx, gc = self._jstage_bwd(range(ne, ne + nj), gx, gc, k, pos_img, w["kb_d"], "d", {}, hooks)
            ops.scatter_tokens_add(gx, route.idx, w["dright"], B, N, k, D)
            ops.add_bf16(w["dcr"], gc, w["dcr"])
        dx0, dc0 = self._jstage_bwd(range(0, ne), w["dright"], w["dcr"], N, None, w["kb_f"], "f", dfe, hooks)
        main.wait_stream(side)
        ops.gemm_tn(dc0, w["ctxP"], self.G("context_embed.weight"), M=D, N=Cd)
        self._cond_bwd(dx0)


# ================================================================================================ plain MMDiT with single-stream blocks
@dataclass
class JointStackDims(JointDims):
    n_single_stream_blocks: int = 0

    def kinds(self) -> list[tuple[str, str]]:
        nj = self.depth - self.n_single_stream_blocks
        return [(f"layers.{i}.", "J" if i < nj else "S") for i in range(self.depth)]


class JointStackEngine(SprintJointEngine):
    """MMDiT(simple_dit=False, n_single_stream_blocks > 0) (mmdit.py:699-731, 789-851): joint blocks, then single-stream blocks on
    the concatenated [context ; image] latents; the image rows of the result feed the last layer.  Reuses the joint and
    single-stream stages of SprintJointEngine without the token routing."""

    route = None

    def _make_layout(self, d: JointStackDims) -> SprintJointLayout:  # type: ignore[override]
        self.kinds = d.kinds()
        return SprintJointLayout(d, sprint=False)

    def _extra_shadows(self, reg) -> None:
        reg("context_embed.weight", self.d.inner_dim, self.d.context_dim, dgrad=False)

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
        T = Lc + N
        Tp = _rup(T, 256)
        if Tp > 2048 or (B * N) % 64 or Lc < 1:
            raise NotImplementedError(f"joint MMDiT HIP path: context + image tokens <= 2048 (got {Lc} + {N}), image tokens a "
                                      "multiple of 64")
        M, Bp, Fo, F = B * N, _rup(B, 64), p * p * d.output_channels, d.mlp_ratio * D
        bf, f32 = torch.bfloat16, torch.float32
        Hh = d.num_heads

        def z(*shape, dtype=bf):
            with torch.inference_mode(False):
                return torch.zeros(*shape, device=dev, dtype=dtype)

        def zr(rows, *rest, dtype=bf):
            return z(_rup(rows, 64), *rest, dtype=dtype)[:rows]

        w: dict[str, object] = {"tokP": z(M, self._ki), "temb": z(Bp, d.frequency_embedding), "pre1": z(Bp, E), "h1": z(Bp, E),
                                "e": z(Bp, E, dtype=f32), "emb": z(Bp, E, dtype=f32), "se": z(Bp, E),
                                "mod": z(Bp, self.layout.mod_rows)}
        w["ctxP"] = zr(B * Lc, _rup(d.context_dim, 64))
        w["x"] = [z(M, D)]
        w["c0"] = zr(B * Lc, D)
        w["kb_f"] = z(B, Tp, dtype=f32)
        w["kb_f"][:, T:] = float("-inf")
        blk = []
        for _, kind in self.kinds:
            per: dict[str, object] = {"ao": z(B * Tp, D), "lse": z(B, Hh, Tp, dtype=f32), "q": z(B, Hh, Tp, 64), "k": z(B, Hh, Tp, 64),
                                      "v": z(B, Hh, Tp, 64)}
            if kind == "J":
                for st, nt in (("input", N), ("context", Lc)):
                    mt = B * nt
                    a = {"x0": zr(mt, D), "mean1": z(mt, dtype=f32), "rstd1": z(mt, dtype=f32), "xm1": zr(mt, D), "qkv": zr(mt, 3 * D),
                         "rrms": z(mt, 2, dtype=f32), "a": zr(mt, D), "t1": zr(mt, D), "x1": zr(mt, D), "mean2": z(mt, dtype=f32),
                         "rstd2": z(mt, dtype=f32), "xm2": zr(mt, D), "u": ops.mlp_u_buffer(zr, mt, D, F, train), "h": zr(mt, F), "t2": zr(mt, D)}
                    if train:
                        a["wg"] = {"dt2": zr(mt, D), "du": zr(mt, 2 * F), "dt1": zr(mt, D), "dqkv": zr(mt, 3 * D)}
                        a["dwb"] = z(2, B, 2, D, dtype=f32)
                    per[st] = a
            else:
                mt = B * T
                per.update({"x0": zr(mt, D), "mean": z(mt, dtype=f32), "rstd": z(mt, dtype=f32), "m": zr(mt, D), "qkv": zr(mt, 3 * D),
                            "rrms": z(mt, 2, dtype=f32), "a": zr(mt, D), "ta": zr(mt, D), "u": ops.mlp_u_buffer(zr, mt, D, F, train), "h": zr(mt, F),
                            "t": zr(mt, D)})
                if train:
                    per["wg"] = {"dt": zr(mt, D), "du": zr(mt, 2 * F), "dqkv": zr(mt, 3 * D)}
                    per["dwb"] = z(1, B, 2, D, dtype=f32)
            blk.append(per)
        w["blk"] = blk
        w["xj"], w["cj"] = z(M, D), zr(B * Lc, D)
        w["lat0"], w["lat_out"] = zr(B * T, D), zr(B * T, D)
        w["xl"] = z(M, D)
        w["meanf"], w["rstdf"], w["xf"] = z(M, dtype=f32), z(M, dtype=f32), z(M, D)
        w["otok"] = z(M, _rup(Fo, 8), dtype=f32)
        w["pred"] = z(B, d.output_channels, H, W, dtype=f32)
        if train:
            w["dO"] = z(M, self._ko)
            for name, mt in ((f"x{N}", M), ("c", B * Lc), ("l", B * T)):
                w["s_" + name] = {"dxa": zr(mt, D), "dxb": zr(mt, D), "dxm": zr(mt, D), "dxm2": zr(mt, D), "da": zr(mt, D),
                                  "dh": zr(mt, F)}
            w["dao_f"] = z(B * Tp, D)
            w["dq_f"], w["dk_f"], w["dv_f"] = (z(B, Hh, Tp, 64) for _ in range(3))
            for n_ in ("dao", "dq", "dk", "dv"):  # (the single-stream stage addresses its attention scratch as "_d")
                w[n_ + "_d"] = w[n_ + "_f"]
            w["dxl"], w["dlat"], w["dxj"], w["dcj"] = z(M, D), zr(B * T, D), z(M, D), zr(B * Lc, D)
            w["dmod"] = z(Bp, self.layout.mod_rows)
            w["dmod32"] = z(Bp, self.layout.mod_rows, dtype=f32)
            w["dse"], w["demb"], w["demb16"] = z(Bp, E, dtype=f32), z(Bp, E, dtype=f32), z(Bp, E)
            w["dh1"], w["dpre1"] = z(Bp, E, dtype=f32), z(Bp, E)
            w["scr_last"], w["scr_conv"] = z(_rup(Fo, 8), D, dtype=f32), z(D, self._ki, dtype=f32)
            # scratch of the bit-reproducible (and faster) small GEMMs / column sums of the conditioning backward (engine._cond_bwd)
            w["det_scr"] = torch.empty(max(2 * self.layout.mod_rows * d.embedding_dim, 1 << 22), device=self.dev, dtype=f32)
            if ops.WgradGroups.widths_ok(D, F) and tuning.on("DL_WGRAD_GROUP"):  # grouped weight gradients (ops.WgradGroups)
                w["tn_slab"] = torch.empty(ops.WgradGroups.slab_floats(D, F), device=self.dev, dtype=f32)
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
        D, L, Cd = d.inner_dim, d.depth, d.context_dim
        nj = L - d.n_single_stream_blocks
        T = Lc + N
        self._train, self._yeff, self._Lc = train, None, Lc
        self._tabs = self._rope[(Lc, gh, gw)]
        mod = self._stem_fwd(x, t, None)
        w["ctxP"][:, :Cd].copy_(ctx.reshape(B * Lc, Cd))
        ops.gemm_nt(w["ctxP"], sh["context_embed.weight|f"], w["c0"], M=B * Lc, N=D, K=w["ctxP"].shape[1])
        kb = w["kb_f"]
        if keep is None:
            kb[:, :Lc].zero_()
        else:
            kb[:, :Lc].zero_().masked_fill_(~keep.to(device=kb.device, dtype=torch.bool), float("-inf"))
        xs, cs = w["x"][0], w["c0"]
        if nj:
            self._jstage_fwd(range(0, nj), xs, cs, N, None, kb, w["xj"], w["cj"])
            xs, cs = w["xj"], w["cj"]
        lat0 = w["lat0"]
        ops.copy_rows3d(cs, Lc * D, D, lat0, T * D, D, B, Lc, D)
        ops.copy_rows3d(xs, N * D, D, lat0[Lc:], T * D, D, B, N, D)
        self._sstage_fwd(range(nj, L), lat0, T, None, kb, w["lat_out"])
        ops.copy_rows3d(w["lat_out"][Lc:], T * D, D, w["xl"], N * D, D, B, N, D)
        mo = self.layout.mod_rows - 2 * D
        ops.ln_modulate_fwd(w["xl"], None, None, mod[:, mo : mo + D], mod[:, mo + D : mo + 2 * D], N, 1e-6, w["xf"], w["meanf"],
                            w["rstdf"])
        ops.gemm_nt(w["xf"], sh["last_layer.linear.weight|f"], w["otok"], bias=self.P("last_layer.linear.bias"), M=M, N=Fo, K=D)
        ops.unpatchify(w["otok"], w["pred"], d.patch_size)
        return w["pred"]

    def feature(self, kblk: int) -> Tensor:
        nj = self.d.depth - self.d.n_single_stream_blocks
        assert self._train
        B, N, D = self.geo[0], self.geo[5], self.d.inner_dim
        if kblk + 1 < nj:
            return self.ws["blk"][kblk + 1]["input"]["x0"].view(B, N, D)
        if kblk + 1 == nj:
            return self.ws["xj"].view(B, N, D)
        raise NotImplementedError("forward hooks on single-stream blocks (the image rows live inside the concatenated latents)")

    def backward(self, dpred: Tensor, dfeats: dict[int, Tensor] | None = None) -> None:
        assert self._train and self.grads is not None
        d, w, sh = self.d, self.ws, self.sh
        B, H, W, gh, gw, N, M, Bp, Fo = self.geo
        D, L, Cd, Lc = d.inner_dim, d.depth, d.context_dim, self._Lc
        nj = L - d.n_single_stream_blocks
        T = Lc + N
        dfe = {kb_: g.reshape(-1, D).to(torch.bfloat16).contiguous() for kb_, g in (dfeats or {}).items()}
        mod, dmod = w["mod"], w["dmod32"]
        dmod[:B].zero_()
        Fo8 = _rup(Fo, 8)
        sN = w[f"s_x{N}"]
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
        mo = self.layout.mod_rows - 2 * D
        ops.ln_modulate_bwd(sN["dxm"], w["xl"], None, None, mod[:, mo : mo + D], N, w["meanf"], w["rstdf"], None, w["dxl"],
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

        # the linears of a block (per token stream) as ONE atomics-free launch (ops.WgradGroups) where the 384 x 192 tile divides
        # them; the stages call wgrad.flush() at every block end
        groups = ops.WgradGroups(w["tn_slab"], on_side, max_wgs=side_wgs) if w.get("tn_slab") is not None else None

        def wgrad(x_grad: Tensor, x_in: Tensor, gname: str) -> None:
            if groups is not None:
                groups.add(x_grad, x_in, self.G(gname))
            else:
                on_side(lambda: ops.gemm_tn(x_grad, x_in, self.G(gname), max_wgs=side_wgs))

        wgrad.flush = groups.flush if groups is not None else (lambda: None)

        def fold_norm(partial: Tensor, gname: str) -> None:
            ev = main.record_event()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                ops.reduce_rows_f32(partial, self.G(gname), B, 2 * D, clear=True)

        hooks = (wgrad, fold_norm, side)
        dlat = w["dlat"]
        dlat.zero_()  # the context rows of the last block's output feed nothing
        ops.copy_rows3d(w["dxl"], N * D, D, dlat[Lc:], T * D, D, B, N, D)
        dl = self._sstage_bwd(range(nj, L), dlat, T, None, w["kb_f"], hooks)
        ops.copy_rows3d(dl[Lc:], T * D, D, w["dxj"], N * D, D, B, N, D)
        ops.copy_rows3d(dl, T * D, D, w["dcj"], Lc * D, D, B, Lc, D)
        gx, gc = w["dxj"], w["dcj"]
        if nj:
            gx, gc = self._jstage_bwd(range(0, nj), gx, gc, N, None, w["kb_f"], "f", dfe, hooks)
        main.wait_stream(side)
        ops.gemm_tn(gc, w["ctxP"], self.G("context_embed.weight"), M=D, N=Cd)
        self._cond_bwd(gx)
