"""Host-side runtime of MMDiT with joint text-image blocks (simple_dit=False; reference networks/denoisers/mmdit.py:107-210,
312-439, 789-851): the DiT engine's arena / shadows / conditioning path, with two token streams per block.

Per block, each stream ("input" = image tokens, "context" = text tokens) runs the DiT sub-layer sequence with its own weights and
its own six modulation rows; the two meet only in the attention: the QK-norm + RoPE kernel of each stream writes its window of
the JOINT q / k / v buffers [B, H, Tp, 64] (rows [0, Lc) context, [Lc, Lc + N) image, padded to a multiple of 256), one
attention launch covers the joint sequence with the key-padding mask as an additive key bias, and the two halves of the result
are sliced back per stream.  The 3-axis RoPE table of the joint sequence (text (1..Lc, 0, 0), image (0, h, w)) is built once per
shape; the context stream of the LAST block feeds nothing (mmdit.py:838-842) and is skipped after the attention.
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import torch
from torch import Tensor

from . import ops, tuning
from .engine import DiTDims, DiTEngine, _rup

STREAMS = ("input", "context")


@dataclass
class JointDims(DiTDims):
    context_dim: int = 768

    def __post_init__(self) -> None:
        if not self.rope_axes_dim:
            self.rope_axes_dim = [self.head_dim // 3] * 3  # mmdit.py:659-664

    def validate(self) -> None:
        D, E = self.inner_dim, self.embedding_dim
        if self.head_dim != 64:
            raise NotImplementedError(f"HIP attention kernels are built for head_dim 64 (got {self.head_dim})")
        if D % 64 or E % 64 or self.frequency_embedding % 64 or D > 1024:
            raise NotImplementedError("inner_dim (<= 1024), embedding_dim and frequency_embedding must be multiples of 64")
        if len(self.rope_axes_dim) != 3 or any(a % 2 for a in self.rope_axes_dim) or sum(self.rope_axes_dim) % 8 or \
                sum(self.rope_axes_dim) > self.head_dim:
            raise NotImplementedError("joint blocks use a 3-axis RoPE (text position, row, col) with even axis widths whose sum is "
                                      f"a multiple of 8 and <= head_dim (got {self.rope_axes_dim})")
        if self.context_dim % 8:
            raise NotImplementedError("context embedding width must be a multiple of 8")
        if self.n_classes is not None:
            raise NotImplementedError("joint blocks are conditioned on a context embedder, not on class labels")


def joint_rope_tables(n_ctx: int, gh: int, gw: int, axes_dim: list[int], base: float) -> tuple[Tensor, Tensor]:
    """cos / sin [n_ctx + gh*gw, sum(axes)/2]: ids of mmdit.py:813-835 (text (t, 0, 0) with t = 1..n_ctx, image (0, h, w)) through
    nn.py:276-307 (angles in fp64, cast to fp32)"""
    pos = torch.zeros(n_ctx + gh * gw, 3, dtype=torch.float64)
    pos[:n_ctx, 0] = torch.arange(1, n_ctx + 1, dtype=torch.float64)
    pos[n_ctx:, 1] = torch.arange(gh, dtype=torch.float64).repeat_interleave(gw)
    pos[n_ctx:, 2] = torch.arange(gw, dtype=torch.float64).repeat(gh)
    cs, sn = [], []
    for a, d in enumerate(axes_dim):
        inv = 1.0 / (torch.tensor(float(base), dtype=torch.float64) ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
        ang = pos[:, a, None] * inv[None, :]
        cs.append(ang.cos().float())
        sn.append(ang.sin().float())
    return torch.cat(cs, 1).contiguous(), torch.cat(sn, 1).contiguous()


class JointLayout:
    """name -> (offset, shape) in the flat f32 arena: the stacked adaLN matrix ([input 6D | context 6D] per block, then the last
    layer's 2D rows) and its bias first, then stem / head, then the blocks"""

    def __init__(self, d: JointDims) -> None:
        D, E, p = d.inner_dim, d.embedding_dim, d.patch_size
        self.entries: dict[str, tuple[int, tuple[int, ...]]] = {}
        self.size = 0
        self.prefixes = [f"layers.{i}." for i in range(d.depth)]

        def add(name: str, shape: tuple[int, ...], align: int = 64) -> None:
            self.size = _rup(self.size, align)
            self.entries[name] = (self.size, shape)
            self.size += math.prod(shape)

        first = True
        for pre in self.prefixes:
            for st in STREAMS:
                add(pre + f"modulation_{st}.lin.weight", (6 * D, E), align=64 if first else 1)
                first = False
        add("last_layer.adaLN_modulation.1.weight", (2 * D, E), align=1)
        first = True
        for pre in self.prefixes:
            for st in STREAMS:
                add(pre + f"modulation_{st}.lin.bias", (6 * D,), align=64 if first else 1)
                first = False
        add("last_layer.adaLN_modulation.1.bias", (2 * D,), align=1)
        self.mod_rows = d.depth * 12 * D + 2 * D
        self.mod_w0, self.mod_b0 = "layers.0.modulation_input.lin.weight", "layers.0.modulation_input.lin.bias"
        add("time_embed.0.weight", (E, d.frequency_embedding))
        add("time_embed.0.bias", (E,))
        add("time_embed.2.weight", (E, E))
        add("time_embed.2.bias", (E,))
        add("conv_proj.weight", (D, d.input_channels, p, p))
        add("context_embed.weight", (D, d.context_dim))
        add("last_layer.linear.weight", (p * p * d.output_channels, D))
        add("last_layer.linear.bias", (p * p * d.output_channels,))
        self.block_first = []
        for pre in self.prefixes:
            self.block_first.append(pre + "input_norm_1.weight")
            for st in STREAMS:
                for n in (1, 2):
                    add(pre + f"{st}_norm_{n}.weight", (D,))
                    add(pre + f"{st}_norm_{n}.bias", (D,), align=1)  # [w; b] adjacent: one reduce writes both gradients
                add(pre + f"attention.qk_norm_{st}.query_norm.scale", (D,))
                add(pre + f"attention.qk_norm_{st}.key_norm.scale", (D,), align=1)
                add(pre + f"attention.qkv_{st}.weight", (3 * D, D))
                add(pre + f"attention.{st}_proj_out.weight", (D, D))
                add(pre + f"mlp_{st}.0.weight", (2 * d.mlp_ratio * D, D))
                add(pre + f"mlp_{st}.2.weight", (D, d.mlp_ratio * D))
        self.size = _rup(self.size, 64)

    def view(self, flat: Tensor, name: str) -> Tensor:
        off, shape = self.entries[name]
        return flat[off : off + math.prod(shape)].view(shape)


class JointEngine(DiTEngine):
    context: tuple[Tensor, Tensor | None] | None = None  # (embeddings [B, Lc, context_dim], keep bool [B, Lc] | None) of this step

    def _make_layout(self, d: JointDims) -> JointLayout:  # type: ignore[override]
        return JointLayout(d)

    def _extra_shadows(self, reg) -> None:
        reg("context_embed.weight", self.d.inner_dim, self.d.context_dim, dgrad=False)

    def _block_shadows(self, reg, pre: str) -> None:
        d, dev = self.d, self.dev
        D, F = d.inner_dim, d.mlp_ratio * d.inner_dim
        for st in STREAMS:
            reg(pre + f"attention.qkv_{st}.weight", 3 * D, D)
            reg(pre + f"attention.{st}_proj_out.weight", D, D)
            reg(pre + f"mlp_{st}.0.weight", 2 * F, D)
            self.sh[pre + f"mlp_{st}.0.weight|g"] = torch.zeros(2 * F, D, device=dev, dtype=torch.bfloat16)
            reg(pre + f"mlp_{st}.2.weight", D, F)

    # ------------------------------------------------------------------ workspace
    def _alloc(self, B: int, H: int, W: int, train: bool, Lc: int = 0) -> None:  # type: ignore[override]
        d, dev = self.d, self.dev
        key = (B, H, W, train, Lc)
        if key == self._ws_key:
            return
        if key in self._ws_cache:
            self.ws, self.geo = self._ws_cache[key]
            self._ws_key = key
            return
        D, E, p, L = d.inner_dim, d.embedding_dim, d.patch_size, d.depth
        gh, gw = H // p, W // p
        N = gh * gw
        T = Lc + N
        Tp = _rup(T, 256)
        if Tp > 2048 or (B * N) % 64 or Lc < 1:
            raise NotImplementedError(f"joint MMDiT HIP path: context + image tokens <= 2048 (got {Lc} + {N}), image tokens a "
                                      "multiple of 64")
        M, Bp, Fo, F = B * N, _rup(B, 64), p * p * d.output_channels, d.mlp_ratio * D
        bf, f32 = torch.bfloat16, torch.float32

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
        w["kbias"] = z(B, Tp, dtype=f32)
        w["kbias"][:, T:] = float("-inf")  # padded key rows
        w["q"], w["k"], w["v"] = (z(B, d.num_heads, Tp, 64) for _ in range(3))
        ntok = {"input": N, "context": Lc}
        blk = []
        for _ in range(L):
            per = {"ao": z(B * Tp, D), "lse": z(B, d.num_heads, Tp, dtype=f32)}
            for st in STREAMS:
                mt = B * ntok[st]
                a = {"x0": zr(mt, D), "mean1": zr(mt, dtype=f32), "rstd1": zr(mt, dtype=f32), "xm1": zr(mt, D), "qkv": zr(mt, 3 * D),
                     "rrms": zr(mt, 2, dtype=f32), "a": zr(mt, D), "t1": zr(mt, D), "x1": zr(mt, D), "mean2": zr(mt, dtype=f32),
                     "rstd2": zr(mt, dtype=f32), "xm2": zr(mt, D), "u": ops.mlp_u_buffer(zr, mt, D, F, train), "h": zr(mt, F), "t2": zr(mt, D)}
                if train:
                    a["wg"] = {"dt2": zr(mt, D), "du": zr(mt, 2 * F), "dt1": zr(mt, D), "dqkv": zr(mt, 3 * D)}
                    a["dwb"] = z(2, B, 2, D, dtype=f32)
                per[st] = a
            if train:  # q / k / v of every block are kept for the backward
                per["q"], per["k"], per["v"] = (z(B, d.num_heads, Tp, 64) for _ in range(3))
            blk.append(per)
        w["blk"] = blk
        w["xl"] = z(M, D)
        w["meanf"], w["rstdf"], w["xf"] = z(M, dtype=f32), z(M, dtype=f32), z(M, D)
        w["otok"] = z(M, _rup(Fo, 8), dtype=f32)
        w["pred"] = z(B, d.output_channels, H, W, dtype=f32)
        if train:
            w["dO"] = z(M, self._ko)
            for st in STREAMS:
                mt = B * ntok[st]
                w["s_" + st] = {"dxa": zr(mt, D), "dxb": zr(mt, D), "dxm": zr(mt, D), "da": zr(mt, D), "dh": zr(mt, F)}
            w["dao"] = z(B * Tp, D)
            w["dq"], w["dk"], w["dv"] = (z(B, d.num_heads, Tp, 64) for _ in range(3))
            w["dmod"] = z(Bp, self.layout.mod_rows)
            w["dmod32"] = z(Bp, self.layout.mod_rows, dtype=f32)
            w["dse"], w["demb"], w["demb16"] = z(Bp, E, dtype=f32), z(Bp, E, dtype=f32), z(Bp, E)
            w["dh1"], w["dpre1"] = z(Bp, E, dtype=f32), z(Bp, E)
            w["scr_last"], w["scr_conv"] = z(_rup(Fo, 8), D, dtype=f32), z(D, self._ki, dtype=f32)
            # scratch of the bit-reproducible (and faster) small GEMMs / column sums of the conditioning backward (engine._cond_bwd)
            w["det_scr"] = torch.empty(max(2 * self.layout.mod_rows * d.embedding_dim, 1 << 22), device=self.dev, dtype=f32)
            if ops.WgradGroups.widths_ok(D, F) and tuning.on("DL_WGRAD_GROUP"):
                w["tn_slab"] = torch.empty(ops.WgradGroups.slab_floats(D, F), device=self.dev, dtype=f32)
        self.ws, self._ws_key = w, key
        self.geo = (B, H, W, gh, gw, N, M, Bp, Fo)
        self.jgeo = (Lc, T, Tp)
        if len(self._ws_cache) >= 8:
            self._ws_cache.pop(next(iter(self._ws_cache)))
        self._ws_cache[key] = (w, self.geo)
        if (Lc, gh, gw) not in self._rope:
            c, s = joint_rope_tables(Lc, gh, gw, d.rope_axes_dim, d.rope_base)
            self._rope[(Lc, gh, gw)] = (c.to(dev), s.to(dev))

    def _streams(self, Lc: int, N: int):
        """(name, tokens, row offset in the joint sequence, column offset of its six modulation rows inside a block's 12D)"""
        D = self.d.inner_dim
        return (("input", N, Lc, 0), ("context", Lc, 0, 6 * D))

    # ------------------------------------------------------------------ forward
    def forward(self, x: Tensor, t: Tensor, y_eff: Tensor | None = None, train: bool = True, refresh: bool = True) -> Tensor:
        d = self.d
        B, C, H, W = x.shape
        assert C == d.input_channels and x.dtype == torch.float32 and x.is_cuda and self.context is not None
        ctx, keep = self.context
        Lc = ctx.shape[1]
        self._alloc(B, H, W, train, Lc)
        Lc, T, Tp = self.jgeo = (Lc, Lc + self.geo[5], _rup(Lc + self.geo[5], 256))  # (a cached workspace does not carry it)
        if refresh:
            self.refresh_shadows(force=train)
        w, sh = self.ws, self.sh
        _, _, _, gh, gw, N, M, Bp, Fo = self.geo
        D, L, Hh, Cd = d.inner_dim, d.depth, d.num_heads, d.context_dim
        cos, sin = self._rope[(Lc, gh, gw)]
        rot = sum(d.rope_axes_dim)
        hr = rot // 2
        self._train, self._yeff = train, None
        mod = self._stem_fwd(x, t, None)
        # context stream input: embeddings -> context_embed (mmdit.py:808-809); key-padding mask -> additive key bias
        w["ctxP"][:, :Cd].copy_(ctx.reshape(B * Lc, Cd))
        ops.gemm_nt(w["ctxP"], sh["context_embed.weight|f"], w["c0"], M=B * Lc, N=D, K=w["ctxP"].shape[1])
        kb = w["kbias"]
        if keep is None:
            kb[:, :Lc].zero_()
        else:
            kb[:, :Lc].zero_().masked_fill_(~keep.to(device=kb.device, dtype=torch.bool), float("-inf"))
        streams = self._streams(Lc, N)
        cur = {"input": w["x"][0], "context": w["c0"]}
        pend: dict[str, tuple | None] = {"input": None, "context": None}
        for i in range(L):
            per, pre, base = w["blk"][i], self.prefixes[i], i * 12 * D
            q, k, v = (per["q"], per["k"], per["v"]) if train else (w["q"], w["k"], w["v"])
            for st, nt, off, mc in streams:
                a, mo = per[st], base + mc
                n1w, n1b = self.P(pre + f"{st}_norm_1.weight"), self.P(pre + f"{st}_norm_1.bias")
                if pend[st] is None:
                    xcur = cur[st]
                    ops.ln_modulate_fwd(xcur, n1w, n1b, mod[:, mo : mo + D], mod[:, mo + D : mo + 2 * D], nt, 1e-5, a["xm1"],
                                        a["mean1"], a["rstd1"])
                else:
                    xcur = a["x0"]
                    pd = pend[st]
                    ops.ln_modulate_fwd(pd[0], n1w, n1b, mod[:, mo : mo + D], mod[:, mo + D : mo + 2 * D], nt, 1e-5, a["xm1"],
                                        a["mean1"], a["rstd1"], t=pd[1], gate=pd[2], x_out=xcur)
                a["xin"] = xcur
                ops.gemm_nt(a["xm1"], sh[pre + f"attention.qkv_{st}.weight|f"], a["qkv"])
                ops.qk_norm_rope_fwd(a["qkv"], self.P(pre + f"attention.qk_norm_{st}.query_norm.scale"),
                                     self.P(pre + f"attention.qk_norm_{st}.key_norm.scale"), cos[off:], sin[off:], q, k, v, a["rrms"],
                                     B, nt, Hh, 64, rot, n_off=off)
            ops.attn_fwd_ex(q, k, v, per["ao"], per["lse"], B, Hh, Tp, Tp, 64, 64**-0.5, kb)
            for st, nt, off, mc in streams:
                if st == "context" and i == L - 1:  # feeds nothing (the reference computes and discards it)
                    continue
                a, mo = per[st], base + mc
                ops.copy_rows3d(per["ao"][off:], Tp * D, D, a["a"], nt * D, D, B, nt, D)
                ops.gemm_nt(a["a"], sh[pre + f"attention.{st}_proj_out.weight|f"], a["t1"])
                ops.ln_modulate_fwd(a["xin"], self.P(pre + f"{st}_norm_2.weight"), self.P(pre + f"{st}_norm_2.bias"),
                                    mod[:, mo + 3 * D : mo + 4 * D], mod[:, mo + 4 * D : mo + 5 * D], nt, 1e-5, a["xm2"], a["mean2"],
                                    a["rstd2"], t=a["t1"], gate=mod[:, mo + 2 * D : mo + 3 * D], x_out=a["x1"])
                if not ops.gemm_nt_swiglu(a["xm2"], sh[pre + f"mlp_{st}.0.weight|g"], a["u"] if train else None, a["h"]):
                    ops.gemm_nt(a["xm2"], sh[pre + f"mlp_{st}.0.weight|f"], a["u"])
                    ops.swiglu_fwd(a["u"], a["h"])
                ops.gemm_nt(a["h"], sh[pre + f"mlp_{st}.2.weight|f"], a["t2"])
                pend[st] = (a["x1"], a["t2"], mod[:, mo + 5 * D : mo + 6 * D])
        mo = L * 12 * D
        pd = pend["input"]
        ops.ln_modulate_fwd(pd[0], None, None, mod[:, mo : mo + D], mod[:, mo + D : mo + 2 * D], N, 1e-6, w["xf"], w["meanf"],
                            w["rstdf"], t=pd[1], gate=pd[2], x_out=w["xl"])
        ops.gemm_nt(w["xf"], sh["last_layer.linear.weight|f"], w["otok"], bias=self.P("last_layer.linear.bias"), M=M, N=Fo, K=D)
        ops.unpatchify(w["otok"], w["pred"], d.patch_size)
        return w["pred"]

    def feature(self, kblk: int) -> Tensor:
        """image-stream output of block k of the last train-mode forward (what a hook on ``layers[k]`` sees as output[0])"""
        assert self._train
        B, N, D = self.geo[0], self.geo[5], self.d.inner_dim
        if kblk + 1 < self.d.depth:
            return self.ws["blk"][kblk + 1]["input"]["x0"].view(B, N, D)
        return self.ws["xl"].view(B, N, D)

    # ------------------------------------------------------------------ backward
    def backward(self, dpred: Tensor, dfeats: dict[int, Tensor] | None = None) -> None:
        assert self._train and self.grads is not None
        d, w, sh = self.d, self.ws, self.sh
        B, H, W, gh, gw, N, M, Bp, Fo = self.geo
        Lc, T, Tp = self.jgeo
        D, L, Hh, Cd = d.inner_dim, d.depth, d.num_heads, d.context_dim
        dfe = {kb: g.reshape(-1, D).to(torch.bfloat16).contiguous() for kb, g in (dfeats or {}).items()}
        cos, sin = self._rope[(Lc, gh, gw)]
        rot = sum(d.rope_axes_dim)
        mod, dmod = w["mod"], w["dmod32"]
        dmod[:B].zero_()
        Fo8 = _rup(Fo, 8)
        streams = self._streams(Lc, N)
        si, sc = w["s_input"], w["s_context"]

        ops.patchify(dpred, w["dO"], d.patch_size, ops.PATCH_PPC)
        gl = self.G("last_layer.linear.weight")
        if Fo == Fo8:
            ops.gemm_tn(w["dO"], w["xf"], gl, M=Fo, N=D)
        else:
            w["scr_last"].zero_()
            ops.gemm_tn(w["dO"], w["xf"], w["scr_last"], M=Fo8, N=D)
            ops.reduce_rows_f32(w["scr_last"], gl, 1, Fo * D)
        ops.colsum(w["dO"], self.G("last_layer.linear.bias"), M, Fo)
        ops.gemm_nt(w["dO"], sh["last_layer.linear.weight|t"], si["dxm"], M=M, N=D, K=self._ko)
        mo = L * 12 * D
        ml = (L - 1) * 12 * D
        al = w["blk"][L - 1]["input"]
        ops.ln_modulate_bwd(si["dxm"], w["xl"], None, None, mod[:, mo : mo + D], N, w["meanf"], w["rstdf"], dfe.get(L - 1), si["dxa"],
                            dmod[:, mo : mo + D], dmod[:, mo + D : mo + 2 * D], None, gate_t=al["t2"],
                            gate=mod[:, ml + 5 * D : ml + 6 * D], dt=al["wg"]["dt2"], dgate=dmod[:, ml + 5 * D : ml + 6 * D])

        main = torch.cuda.current_stream()
        side = self._side_stream()
        side.wait_stream(main)
        side_wgs = tuning.integer("DL_SIDE_WGS", 128)

        def on_side(fn) -> None:
            ev = main.record_event()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                fn()

        # the four linears of a stream as ONE atomics-free launch once its last operand (dqkv) exists (ops.WgradGroups); widths the
        # 384 x 192 tile does not divide keep one atomic launch per linear
        groups = ops.WgradGroups(w["tn_slab"], on_side, max_wgs=side_wgs) if w.get("tn_slab") is not None else None

        def wgrad(x_grad: Tensor, x_in: Tensor, gname: str) -> None:
            if groups is not None:
                groups.add(x_grad, x_in, self.G(gname))
            else:
                on_side(lambda: ops.gemm_tn(x_grad, x_in, self.G(gname), max_wgs=side_wgs))

        def fold_norm(partial: Tensor, gname: str) -> None:
            ev = main.record_event()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                ops.reduce_rows_f32(partial, self.G(gname), B, 2 * D, clear=True)

        dx: dict[str, Tensor | None] = {"input": si["dxa"], "context": None}  # residual-stream gradients (None: zero so far)
        scr = {"input": si, "context": sc}

        def other(s, cur):
            return s["dxb"] if (cur is not None and cur.data_ptr() == s["dxa"].data_ptr()) else s["dxa"]

        for i in reversed(range(L)):
            per, pre, base = w["blk"][i], self.prefixes[i], i * 12 * D
            w["dao"].zero_()
            for st, nt, off, mc in streams:
                if st == "context" and i == L - 1:
                    continue
                a, mo, s, g = per[st], base + mc, scr[st], per[st]["wg"]
                # MLP branch (dt2 / dgate were produced by the LayerNorm backward that precedes this block in the chain)
                wgrad(g["dt2"], a["h"], pre + f"mlp_{st}.2.weight")
                ops.mlp_swiglu_bwd(g["dt2"], sh[pre + f"mlp_{st}.2.weight|t"], a["xm2"], sh[pre + f"mlp_{st}.0.weight|g"], a["u"], s["dh"], g["du"])
                wgrad(g["du"], a["xm2"], pre + f"mlp_{st}.0.weight")
                ops.gemm_nt(g["du"], sh[pre + f"mlp_{st}.0.weight|t"], s["dxm"])
                nx = other(s, dx[st])
                ops.ln_modulate_bwd(s["dxm"], a["x1"], self.P(pre + f"{st}_norm_2.weight"), self.P(pre + f"{st}_norm_2.bias"),
                                    mod[:, mo + 3 * D : mo + 4 * D], nt, a["mean2"], a["rstd2"], dx[st], nx,
                                    dmod[:, mo + 3 * D : mo + 4 * D], dmod[:, mo + 4 * D : mo + 5 * D], a["dwb"][1], gate_t=a["t1"],
                                    gate=mod[:, mo + 2 * D : mo + 3 * D], dt=g["dt1"], dgate=dmod[:, mo + 2 * D : mo + 3 * D])
                fold_norm(a["dwb"][1], pre + f"{st}_norm_2.weight")
                dx[st] = nx
                wgrad(g["dt1"], a["a"], pre + f"attention.{st}_proj_out.weight")
                ops.gemm_nt(g["dt1"], sh[pre + f"attention.{st}_proj_out.weight|t"], s["da"])
                ops.copy_rows3d(s["da"], nt * D, D, w["dao"][off:], Tp * D, D, B, nt, D)
            ops.attn_bwd_ex(per["q"], per["k"], per["v"], per["ao"], w["dao"], per["lse"], w["dq"], w["dk"], w["dv"], B, Hh, Tp, Tp,
                            64, 64**-0.5, w["kbias"])
            for st, nt, off, mc in streams:
                a, mo, s, g = per[st], base + mc, scr[st], per[st]["wg"]
                ops.qk_norm_rope_bwd(w["dq"], w["dk"], w["dv"], a["qkv"], self.P(pre + f"attention.qk_norm_{st}.query_norm.scale"),
                                     self.P(pre + f"attention.qk_norm_{st}.key_norm.scale"), cos[off:], sin[off:], a["rrms"],
                                     g["dqkv"], self.G(pre + f"attention.qk_norm_{st}.query_norm.scale"), B, nt, Hh, 64, rot,
                                     n_off=off)
                wgrad(g["dqkv"], a["xm1"], pre + f"attention.qkv_{st}.weight")
                ops.gemm_nt(g["dqkv"], sh[pre + f"attention.qkv_{st}.weight|t"], s["dxm"])
                if st == "input" and i - 1 in dfe and dx[st] is not None:
                    ops.add_bf16(dx[st], dfe[i - 1], dx[st])
                nxt = {}
                if i > 0:
                    mp = (i - 1) * 12 * D + mc
                    ap = w["blk"][i - 1][st]
                    nxt = dict(gate_t=ap["t2"], gate=mod[:, mp + 5 * D : mp + 6 * D], dt=ap["wg"]["dt2"],
                               dgate=dmod[:, mp + 5 * D : mp + 6 * D])
                nx = other(s, dx[st])
                ops.ln_modulate_bwd(s["dxm"], a["xin"], self.P(pre + f"{st}_norm_1.weight"), self.P(pre + f"{st}_norm_1.bias"),
                                    mod[:, mo : mo + D], nt, a["mean1"], a["rstd1"], dx[st], nx, dmod[:, mo : mo + D],
                                    dmod[:, mo + D : mo + 2 * D], a["dwb"][0], **nxt)
                fold_norm(a["dwb"][0], pre + f"{st}_norm_1.weight")
                dx[st] = nx
            if groups is not None:
                groups.flush()
            if self.reducer is not None:
                self.reducer.ready(*self.layer_ranges[i], extra_events=(side.record_event(),))
        main.wait_stream(side)
        # context_embed (no gradient flows into the precomputed embeddings)
        ops.gemm_tn(dx["context"], w["ctxP"], self.G("context_embed.weight"), M=D, N=Cd)
        self._cond_bwd(dx["input"])
