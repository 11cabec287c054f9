"""fp32-class launch sequences of the UNet (`precision_type="no"`, the reference's default: training/trainers/common.py:76,105 --
BASELINE configuration 1, train_mnist_ddpm.yaml, inherits it).

`UNetEngineF32` is `unet_engine.UNetEngine` with every primitive replaced by its f32 form (csrc/f32.hip): the block orchestration
(`_res_fwd` / `_res_bwd`, `_resample_*`, `_attn_*`, the skip-connection bookkeeping of forward / backward) is inherited unchanged, so
the two regimes cannot drift apart structurally.  Differences:
  * activations are NHWC f32 rows; a 3x3 convolution (unet.py:187,208,594,745) is `dl_f32_im2col3x3` + `dl_f32_gemm` against the weight
    in its NATIVE [Co, Ci*9] layout (no shadows), its data gradient `dl_f32_gemm` + `dl_f32_col2im3x3`, its weight gradient
    `dl_f32_gemm` accumulated straight into the gradient arena; 1x1 convolutions and Linear layers are `dl_f32_gemm` on the arena;
  * GroupNorm32 / FiLM (nn.py:11-13, unet.py:215-237) are `dl_f32_gn_*` (one workgroup per (sample, group), per-sample partials of the
    affine gradients folded in a fixed order); the AttentionBlock (unet.py:296-322) is the strided batched `dl_f32_gemm` over
    materialised probabilities + `dl_f32_softmax_*`, like the fp32 DiT;
  * nothing runs on a side stream and nothing is accumulated through atomics: a step is bit-reproducible.
"""

from __future__ import annotations

import torch
from torch import Tensor

from . import ops
from .unet_engine import UNetDims, UNetEngine, UNetLayout, _rup, build_plan


class _F32Ops:
    """the entry points `UNetEngine`'s block orchestration calls through `self.o`, in f32"""

    def __init__(self, eng: "UNetEngineF32") -> None:
        self.eng = eng

    @staticmethod
    def expand2x2(x, out, B, Hi, Wi, C, scale):
        ops.f32_resample2x2(x, out, B, Hi, Wi, C, scale, 1)

    @staticmethod
    def reduce2x2(x, out, B, Ho, Wo, C, scale):
        ops.f32_resample2x2(x, out, B, Ho, Wo, C, scale, 0)

    @staticmethod
    def resample2x2_pair(x0, out0, x1, out1, B, Hs, Ws, C, scale, expand):
        ops.f32_resample2x2(x0, out0, B, Hs, Ws, C, scale, 1 if expand else 0)
        ops.f32_resample2x2(x1, out1, B, Hs, Ws, C, scale, 1 if expand else 0)

    @staticmethod
    def pick2x2(x, out, B, Ho, Wo, C):
        ops.f32_resample2x2(x, out, B, Ho, Wo, C, 1.0, 2)

    @staticmethod
    def stuff2x2(dy, out, B, Hi, Wi, C):
        ops.f32_resample2x2(dy, out, B, Hi, Wi, C, 1.0, 3)

    rowbias_add = staticmethod(ops.f32_rowbias_add)
    rowbias_bwd = staticmethod(ops.f32_rowbias_bwd)
    nchw_to_nhwc = staticmethod(ops.f32_nchw_to_nhwc)
    nhwc_to_nchw = staticmethod(ops.f32_nhwc_to_nchw)
    copy2d_bf16 = staticmethod(ops.f32_copy2d)  # (the orchestration's name for "strided 2-D copy")

    # AttentionBlock core (unet.py:311-318): any token count / head width (ops.f32_attn_fwd / f32_attn_bwd)
    attn_small_fwd = staticmethod(ops.f32_attn_fwd)

    def attn_small_bwd(self, q, k, v, dout, probs, dq, dk, dv, B, n, H, dh):
        dP = self.eng._scr("attn_dp", B * H * n * n, torch.float32).view(B, H, n, n)
        ops.f32_attn_bwd(q, k, v, dout, probs, dP, dq, dk, dv, B, n, H, dh)


class UNetEngineF32(UNetEngine):
    precision = "fp32"

    def __init__(self, dims: UNetDims, device: torch.device | str = "cuda") -> None:
        dims.validate()
        self.d = dims
        self.dev = torch.device(device)
        self.plan = build_plan(dims)
        self.layout = UNetLayout(dims, self.plan)
        self.params: Tensor | None = None
        self.grads: Tensor | None = None
        self.manual_version = 0
        self.param_version = 0
        self.reducer = None
        self._scratch: dict[str, Tensor] = {}
        self._saved: dict | None = None
        self.o = _F32Ops(self)
        self.sh = {}

    # ------------------------------------------------------------------ parameters: the GEMMs read the f32 arena directly
    def bind(self, params: Tensor, grads: Tensor | None) -> None:
        assert params.dtype == torch.float32 and params.numel() == self.layout.size and params.is_cuda
        self.params, self.grads = params, grads

    def refresh_shadows(self, force: bool = False) -> None:
        pass

    @property
    def _use_side(self) -> bool:
        return False

    def _new(self, *shape: int, dtype=torch.float32, zero: bool = False) -> Tensor:
        if dtype == torch.bfloat16:  # (buffers the inherited orchestration asks for by the bf16 default)
            dtype = torch.float32
        with torch.inference_mode(False):
            return (torch.zeros if zero else torch.empty)(*shape, device=self.dev, dtype=dtype)

    def _scr(self, key: str, numel: int, dtype=torch.float32) -> Tensor:
        return super()._scr(key, numel, torch.float32)

    def _wscr(self, out_elems: int) -> Tensor:
        """split-K scratch of dl_f32_gemm for products with few output tiles and a long contraction (weight gradients over all
        pixels): up to 16 partial images"""
        return self._scr("f32_splitk", 16 * max(out_elems, 1 << 16))

    def _fold(self, partial: Tensor, g: Tensor, B: int, n: int) -> None:
        ops.reduce_rows_batched_f32(partial, 0, g, 0, 1, B, n)

    # ------------------------------------------------------------------ primitives
    def _conv3(self, x: Tensor, B: int, H: int, W: int, ci: int, name: str, co: int, resid: Tensor | None = None) -> Tensor:
        M = B * H * W
        cols = self._scr("cols", M * 9 * ci).view(M, 9 * ci)
        ops.f32_im2col3x3(x, cols, B, H, W, ci)
        out = self._new(M, co)
        # (deep levels: few output tiles over a 9 ci contraction -- split-K partials where the product would not fill the chip)
        ops.f32_linear(cols, self.P(name).view(co, 9 * ci), out, bias=self.P(name[:-6] + "bias"),
                       scratch=self._wscr(M * co) if M * co <= (1 << 23) else None)
        if resid is not None:  # x + h of the ResBlock (unet.py:237)
            ops.f32_add(out, resid, out)
        return out

    def _conv3_bwd(self, dy: Tensor, x: Tensor, B: int, H: int, W: int, ci: int, name: str, co: int,
                   need_dx: bool = True) -> Tensor | None:
        M = B * H * W
        w = self.P(name).view(co, 9 * ci)
        dyv = dy[:, :co]  # (the head's gradient buffer is padded to 8 columns by the inherited backward)
        cols = self._scr("cols", M * 9 * ci).view(M, 9 * ci)
        ops.f32_im2col3x3(x, cols, B, H, W, ci)
        ops.colsum(dyv, self.Gr(name[:-6] + "bias"), M, co, scratch=self._scr("colsum", 512 * max(co, 8)))
        ops.f32_gemm(dyv, cols, self.Gr(name).view(co, 9 * ci), co, 9 * ci, M, lda=dy.stride(0), ldb=9 * ci, ldc=9 * ci, ta=True,
                     tb=True, accumulate=True, scratch=self._wscr(co * 9 * ci))
        if not need_dx:
            return None
        dcols = cols  # (the im2col matrix is dead after the weight gradient: reuse its storage)
        ops.f32_gemm(dyv, w, dcols, M, 9 * ci, co, lda=dy.stride(0), ldb=9 * ci, ldc=9 * ci, tb=True,
                     scratch=self._wscr(M * 9 * ci) if M * 9 * ci <= (1 << 23) else None)
        dx = self._new(M, ci)
        ops.f32_col2im3x3(dcols, dx, B, H, W, ci)
        return dx

    def _lin_fwd_pair(self, x0: Tensor, name0: str, co0: int, x1: Tensor, name1: str, co1: int, ci: int):
        return self._lin_fwd(x0, name0, co0, ci), self._lin_fwd(x1, name1, co1, ci)  # (the paired launch is a bf16 kernel)

    def _lin_bwd_pair(self, dy0: Tensor, x0: Tensor, name0: str, co0: int, dy1: Tensor, x1: Tensor, name1: str, co1: int, ci: int):
        return self._lin_bwd(dy0, x0, name0, co0, ci), self._lin_bwd(dy1, x1, name1, co1, ci)

    def _lin_fwd(self, x: Tensor, name: str, co: int, ci: int, resid: Tensor | None = None, out: Tensor | None = None) -> Tensor:
        M = x.shape[0]
        out = self._new(M, co) if out is None else out
        ops.f32_gemm(x, self.P(name).view(co, ci), out, M, co, ci, lda=x.stride(0), ldb=ci, ldc=out.stride(0),
                     bias=self.P(name[:-6] + "bias"))
        if resid is not None:
            ops.f32_add(out, resid, out)
        return out

    def _lin_bwd(self, dy: Tensor, x: Tensor, name: str, co: int, ci: int, need_dx: bool = True) -> Tensor | None:
        M = dy.shape[0]
        ops.colsum(dy, self.Gr(name[:-6] + "bias"), M, co, scratch=self._scr("colsum", 512 * max(co, 8)))
        ops.f32_gemm(dy, x, self.Gr(name).view(co, ci), co, ci, M, lda=dy.stride(0), ldb=x.stride(0), ldc=ci, ta=True, tb=True,
                     accumulate=True, scratch=self._wscr(co * ci))
        if not need_dx:
            return None
        dx = self._new(M, ci)
        ops.f32_gemm(dy, self.P(name).view(co, ci), dx, M, ci, co, lda=dy.stride(0), ldb=ci, ldc=ci, tb=True)
        return dx

    def _gn(self, x: Tensor, B: int, HW: int, C: int, wname: str, film=None, silu: bool = True, stats: Tensor | None = None):
        if stats is None:
            stats = self._new(B, self.G, 2)
            ops.f32_gn_stats(x, stats, B, HW, C, self.G)
        out = self._new(B * HW, C)
        fs, fh = film if film is not None else (None, None)
        ops.f32_gn_apply_fwd(x, stats, self.P(wname + "weight"), self.P(wname + "bias"), fs, fh, silu, out, B, HW, C, self.G)
        return out, stats

    def _gn_bwd(self, dout: Tensor, x: Tensor, stats: Tensor, B: int, HW: int, C: int, wname: str, film=None, dfilm=None,
                silu: bool = True, dres: Tensor | None = None) -> Tensor:
        dx = self._new(B * HW, C)
        fs, fh = film if film is not None else (None, None)
        dfs, dfh = dfilm if dfilm is not None else (None, None)
        part = self._scr("gn_part", 2 * B * C).view(2, B, C)
        ops.f32_gn_bwd(dout, x, stats, self.P(wname + "weight"), self.P(wname + "bias"), fs, fh, silu, dres, dx, part[0], part[1], dfs,
                       dfh, B, HW, C, self.G)
        self._fold(part[0], self.Gr(wname + "weight"), B, C)
        self._fold(part[1], self.Gr(wname + "bias"), B, C)
        return dx

    def _add(self, a: Tensor | None, b: Tensor | None) -> Tensor | None:
        if a is None:
            return b
        if b is None:
            return a
        out = self._new(*a.shape)
        ops.f32_add(a, b, out)
        return out

    # ------------------------------------------------------------------ conditioning (unet.py:832-838; nn.py:106-114, 149-164)
    def _emb_matrix(self, flat: Tensor) -> tuple[Tensor, Tensor]:
        lay, te = self.layout, 4 * self.d.model_channels
        R = lay.emb_rows
        w0, b0 = lay.entries[lay.emb_w0][0], lay.entries[lay.emb_b0][0]
        return flat[w0 : w0 + R * te].view(R, te), flat[b0 : b0 + R]

    def _cond_fwd(self, t: Tensor, y_eff: Tensor | None, B: int):
        d = self.d
        mc, te = d.model_channels, 4 * d.model_channels
        temb, pre1, h1, e = self._new(B, mc), self._new(B, te), self._new(B, te), self._new(B, te)
        ops.f32_timestep_embedding(t, temb)
        ops.f32_linear(temb, self.P("time_embed.0.weight"), h1, bias=self.P("time_embed.0.bias"), act=ops.ACT_SILU, pre_out=pre1)
        ops.f32_linear(h1, self.P("time_embed.2.weight"), e, bias=self.P("time_embed.2.bias"))
        table = self.P("label_embed.embedding.weight") if d.n_classes is not None else None
        emb, se = self._new(B, te), self._new(B, te)
        ops.f32_cond_combine_fwd(e, table, y_eff if table is not None else None, emb, se)
        w, bias = self._emb_matrix(self.params)
        eo = self._new(B, _rup(self.layout.emb_rows, 64))  # (row stride of the inherited backward's gradient buffer)
        ops.f32_gemm(se, w, eo, B, self.layout.emb_rows, te, lda=te, ldb=te, ldc=eo.stride(0), bias=bias)
        return eo, dict(temb=temb, pre1=pre1, h1=h1, emb=emb, se=se)

    def _cond_bwd(self, deo: Tensor, s: dict, B: int) -> None:
        d = self.d
        mc, te = d.model_channels, 4 * d.model_channels
        R = self.layout.emb_rows
        w, _ = self._emb_matrix(self.params)
        gw, gb = self._emb_matrix(self.grads)
        scr = self._wscr(max(R * te, B * te))
        ops.f32_gemm(deo, s["se"], gw, R, te, B, lda=deo.stride(0), ldb=te, ldc=te, ta=True, tb=True, accumulate=True)
        ops.colsum(deo[:B], gb, B, R, scratch=scr)
        dse = self._new(B, te)
        ops.f32_gemm(deo, w, dse, B, te, R, lda=deo.stride(0), ldb=te, ldc=te, tb=True, scratch=scr)
        table = d.n_classes is not None
        demb = self._new(B, te)
        ops.f32_cond_combine_bwd(dse, s["emb"], s["y"] if table else None, demb, self.Gr("label_embed.embedding.weight") if table else None)
        ops.colsum(demb, self.Gr("time_embed.2.bias"), B, te, scratch=scr)
        ops.f32_linear_wgrad(demb, s["h1"], self.Gr("time_embed.2.weight"))
        dh1 = self._new(B, te)
        ops.f32_linear_dgrad(demb, self.P("time_embed.2.weight"), dh1)
        dpre1 = self._new(B, te)
        ops.f32_silu_bwd(dh1, s["pre1"], dpre1)
        ops.f32_linear_wgrad(dpre1, s["temb"], self.Gr("time_embed.0.weight"))
        ops.colsum(dpre1, self.Gr("time_embed.0.bias"), B, te, scratch=scr)
