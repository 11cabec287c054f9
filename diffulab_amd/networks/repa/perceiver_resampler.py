"""Perceiver resampler of the REPA loss on the HIP path -- drop-in for
``diffulab.networks.repa.perceiver_resampler.PerceiverResampler`` (perceiver_resampler.py:172-252): same constructor kwargs, same
``state_dict`` keys (``latents``, ``layers.{i}.0.{norm_x,norm_latents,to_q,to_kv,to_out}``, ``layers.{i}.1.{0,1,3}``, ``norm``).

The ``nn.Module`` tree only owns parameters; forward and backward of the whole stack are explicit launch sequences over the C ABI
inside one autograd node:

  LayerNorm            dl_ln_modulate_fwd / _bwd with a zero modulation row (the backward also folds the residual add)
  to_q / to_kv / ...   bf16 MFMA GEMMs (dl_gemm_nt; residual add and exact-erf GELU fused into the epilogue), wgrads dl_gemm_tn
  head split + RoPE    dl_heads_split_rope: keys of x rows [0, n) rotated on the sqrt(n) x sqrt(n) grid, latent keys rows [n, n+m),
                       the key buffer padded to a multiple of 256 rows and the padding masked with a -inf key bias
  attention            dl_attn_fwd_ex / _bwd_ex (m latent queries against n + m keys)

Position ids: the reference builds un-batched ids when ``cos_sin`` is not given and then indexes the tables as batched
(utils/nn.py:342), which raises; this module implements the evident intent (one sqrt(n) x sqrt(n) grid shared by the batch) and the
parity fixture pins it through the reference module's own ``cos_sin`` argument (tests/golden/make_golden.py:gen_resampler).
The residual stream is bf16, accumulation f32 -- the same arithmetic contract as the DiT engine.
"""

from __future__ import annotations

import math

import torch
import torch.nn as nn
from torch import Tensor

from ... import ops
from ...engine import rope_grid_tables


def _rup(v: int, m: int) -> int:
    return (v + m - 1) // m * m


class _Attention(nn.Module):  # parameter holder: PerceiverAttention (perceiver_resampler.py:96-120)
    def __init__(self, dim: int, head_dim: int, num_heads: int) -> None:
        super().__init__()
        inner = head_dim * num_heads
        self.norm_x, self.norm_latents = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)


class _ResamplerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod: "PerceiverResampler", x: Tensor, *params: Tensor) -> Tensor:
        y, saved = mod._forward_launches(x)
        ctx.mod, ctx.saved, ctx.x_shape = mod, saved, x.shape
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        dx, grads = ctx.mod._backward_launches(ctx.saved, dy)
        ctx.saved = None
        return (None, dx.view(ctx.x_shape), *grads)


class PerceiverResampler(nn.Module):
    def __init__(self, dim: int, depth: int, head_dim: int = 64, num_heads: int = 8, ff_mult: int = 4,
                 rope_axes_dim: list[int] | None = None, num_latents: int = 256, rope_base: int = 10_000) -> None:
        super().__init__()
        if head_dim != 64:
            raise NotImplementedError("PerceiverResampler HIP path: head_dim must be 64 (attention kernels)")
        if num_latents % 256 or dim % 64 or dim > 1024 or int(dim * ff_mult) % 64:
            raise NotImplementedError("PerceiverResampler HIP path: num_latents % 256, dim % 64, dim <= 1024, dim*ff_mult % 64")
        self.latents = nn.Parameter(torch.randn(num_latents, dim))
        self.rope_base = rope_base
        if rope_axes_dim is None:
            rope_axes_dim = [head_dim // 2, head_dim // 2]
        if len(rope_axes_dim) != 2 or sum(rope_axes_dim) > head_dim or sum(rope_axes_dim) % 8:
            raise NotImplementedError("PerceiverResampler HIP path: 2-axis RoPE with sum(axes) % 8 == 0 and <= head_dim")
        hidden = int(dim * ff_mult)
        self.layers = nn.ModuleList([
            nn.ModuleList([_Attention(dim, head_dim, num_heads),
                           nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden, bias=False), nn.GELU(),
                                         nn.Linear(hidden, dim, bias=False))])
            for _ in range(depth)])
        self.rope_axes_dim = list(rope_axes_dim)
        self.norm = nn.LayerNorm(dim)
        self.dim, self.depth, self.head_dim, self.num_heads, self.hidden = dim, depth, head_dim, num_heads, hidden
        self._shadow_key: tuple | None = None

    # ------------------------------------------------------------------ bf16 weight shadows (one batched cast launch)
    def _weights(self) -> list[Tensor]:
        ws = []
        for attn, ff in self.layers:
            ws += [attn.to_q.weight, attn.to_kv.weight, attn.to_out.weight, ff[1].weight, ff[3].weight]
        return ws

    def _refresh_shadows(self, dev: torch.device) -> list[tuple[Tensor, Tensor]]:
        ws = self._weights()
        key = tuple(w.data_ptr() for w in ws)
        if key != self._shadow_key:
            bf = torch.bfloat16
            self._shadows = [(torch.empty(w.shape[0], w.shape[1], device=dev, dtype=bf),
                              torch.empty(w.shape[1], w.shape[0], device=dev, dtype=bf)) for w in ws]
            self._cast = ops.CastTable([(w.detach(), f, t, None) for w, (f, t) in zip(ws, self._shadows)])
            self._zmod = torch.zeros(1, self.dim, device=dev, dtype=bf)
            self._rope: dict[int, tuple[Tensor, Tensor]] = {}
            self._bias: dict[tuple[int, int], Tensor | None] = {}
            self._shadow_key = key
        self._cast.run()
        return self._shadows

    # ------------------------------------------------------------------ launch sequences
    def _ln_fwd(self, x: Tensor, norm: nn.LayerNorm) -> tuple[Tensor, Tensor, Tensor]:
        M = x.shape[0]
        out = torch.empty_like(x)
        mean, rstd = torch.empty(M, device=x.device), torch.empty(M, device=x.device)
        ops.ln_modulate_fwd(x, norm.weight.detach(), norm.bias.detach(), self._zmod, self._zmod, M, norm.eps, out, mean, rstd)
        return out, mean, rstd

    def _ln_bwd(self, dout: Tensor, x: Tensor, norm: nn.LayerNorm, mean: Tensor, rstd: Tensor, dres: Tensor | None,
                grads: dict[int, Tensor]) -> Tensor:
        D = self.dim
        dx = torch.empty_like(x)
        dwb = torch.zeros(1, 2, D, device=x.device)
        junk = torch.zeros(2, D, device=x.device)  # gradient of the (zero, constant) modulation row: discarded
        ops.ln_modulate_bwd(dout, x, norm.weight.detach(), norm.bias.detach(), self._zmod, x.shape[0], mean, rstd, dres, dx,
                            junk[0:1], junk[1:2], dwb)
        grads[id(norm.weight)], grads[id(norm.bias)] = dwb[0, 0], dwb[0, 1]
        return dx

    def _forward_launches(self, x: Tensor):
        if x.device.type != "cuda":
            raise RuntimeError("diffulab_amd.PerceiverResampler runs on an MI355X only (no CPU fallback)")
        B, n, D = x.shape
        g = math.isqrt(n)
        m, H, I, F = self.latents.shape[0], self.num_heads, self.num_heads * 64, self.hidden
        Nk = _rup(n + m, 256)
        if g * g != n or D != self.dim or (B * n) % 64 or Nk > 2048:
            raise NotImplementedError(f"PerceiverResampler HIP path: x [B, n, {self.dim}] with n a square, B*n % 64 == 0, "
                                      f"n + num_latents <= 2048 (got {tuple(x.shape)})")
        dev, bf = x.device, torch.bfloat16
        sh = self._refresh_shadows(dev)
        if n not in self._rope:
            cs, sn = rope_grid_tables(g, g, self.rope_axes_dim, float(self.rope_base))
            self._rope[n] = (cs.to(dev), sn.to(dev))
        cos, sin = self._rope[n]
        rot = sum(self.rope_axes_dim)
        if (B, n) not in self._bias:  # padded key rows [n + m, Nk) are masked out
            kb = None
            if Nk != n + m:
                kb = torch.zeros(B, Nk, device=dev)
                kb[:, n + m:] = float("-inf")
            self._bias[(B, n)] = kb
        kb = self._bias[(B, n)]
        Mx, Ml = B * n, B * m
        x2 = x.reshape(Mx, D)
        if x2.dtype != bf or not x2.is_contiguous():
            x2 = x2.to(bf).contiguous()
        lat = torch.empty(B, m, D, device=dev, dtype=bf)
        lat.copy_(self.latents.detach()[None])  # repeat 'n d -> b n d'
        lat = lat.view(Ml, D)
        scale = 64**-0.5
        layers = []
        for i, (attn, ff) in enumerate(self.layers):
            (wq, _), (wkv, _), (wo, _), (w1, _), (w2, _) = sh[5 * i: 5 * i + 5]
            xn, mx, rx = self._ln_fwd(x2, attn.norm_x)
            ln, ml, rl = self._ln_fwd(lat, attn.norm_latents)
            qf, kvx, kvl = (torch.empty(r, c, device=dev, dtype=bf) for r, c in ((Ml, I), (Mx, 2 * I), (Ml, 2 * I)))
            ops.gemm_nt(ln, wq, qf)
            ops.gemm_nt(xn, wkv, kvx)
            ops.gemm_nt(ln, wkv, kvl)
            q = torch.empty(B, H, m, 64, device=dev, dtype=bf)
            alloc = torch.zeros if kb is not None else torch.empty
            k, v = alloc(B, H, Nk, 64, device=dev, dtype=bf), alloc(B, H, Nk, 64, device=dev, dtype=bf)
            ops.heads_split_rope(qf, q, B, H, m, 0)
            ops.heads_split_rope(kvx[:, :I], k, B, H, n, 0, cos, sin, rot)  # RoPE on the keys that come from x only
            ops.heads_split_rope(kvl[:, :I], k, B, H, m, n)
            ops.heads_split_rope(kvx[:, I:], v, B, H, n, 0)
            ops.heads_split_rope(kvl[:, I:], v, B, H, m, n)
            att = torch.empty(Ml, I, device=dev, dtype=bf)
            lse = torch.empty(B, H, m, device=dev)
            ops.attn_fwd_ex(q, k, v, att, lse, B, H, m, Nk, 64, scale, kb)
            lat2 = torch.empty(Ml, D, device=dev, dtype=bf)
            ops.gemm_nt(att, wo, lat2, resid=lat)
            h, mh, rh = self._ln_fwd(lat2, ff[0])
            pre, gl = torch.empty(Ml, F, device=dev, dtype=bf), torch.empty(Ml, F, device=dev, dtype=bf)
            ops.gemm_nt(h, w1, gl, act=ops.ACT_GELU, pre_out=pre)
            lat3 = torch.empty(Ml, D, device=dev, dtype=bf)
            ops.gemm_nt(gl, w2, lat3, resid=lat2)
            layers.append((xn, mx, rx, ln, ml, rl, lat, q, k, v, att, lse, lat2, h, mh, rh, pre, gl))
            lat = lat3
        y, my, ry = self._ln_fwd(lat, self.norm)
        saved = (x2, layers, lat, my, ry, (B, n, m, Nk, kb, cos, sin, rot), sh)
        return y.view(B, m, D), saved

    def _backward_launches(self, saved, dy: Tensor):
        x2, layers, lat_f, my, ry, (B, n, m, Nk, kb, cos, sin, rot), sh = saved
        D, H, I, F = self.dim, self.num_heads, self.num_heads * 64, self.hidden
        Mx, Ml = B * n, B * m
        dev, bf, f32 = x2.device, torch.bfloat16, torch.float32
        scale = 64**-0.5
        grads: dict[int, Tensor] = {}
        dy2 = dy.reshape(Ml, D)
        if dy2.dtype != bf or not dy2.is_contiguous():
            dy2 = dy2.to(bf).contiguous()
        dlat = self._ln_bwd(dy2, lat_f, self.norm, my, ry, None, grads)
        dx = None
        # weight gradients through the partial-image forms (dl_gemm_tn_det: the atomics-free tiled kernel where its tile divides,
        # four token ranges of the widest matrix fill the chip)
        scr = ops.shared_scratch(dev, 4 * max(D * F, 2 * I * D))
        for i in reversed(range(self.depth)):
            attn, ff = self.layers[i]
            (_, tq), (_, tkv), (_, to), (_, t1), (_, t2) = sh[5 * i: 5 * i + 5]
            xn, mx, rx, ln, ml, rl, lat, q, k, v, att, lse, lat2, h, mh, rh, pre, gl = layers[i]
            # FeedForward: lat3 = lat2 + gelu(LN(lat2) W1^T) W2^T
            dgl = torch.empty(Ml, F, device=dev, dtype=f32)
            ops.gemm_nt(dlat, t2, dgl)
            dw2 = torch.zeros(D, F, device=dev)
            ops.gemm_tn(dlat, gl, dw2, scratch=scr)
            dpre = torch.empty(Ml, F, device=dev, dtype=bf)
            ops.gelu_bwd(dgl, pre, dpre)
            dw1 = torch.zeros(F, D, device=dev)
            ops.gemm_tn(dpre, h, dw1, scratch=scr)
            dh = torch.empty(Ml, D, device=dev, dtype=bf)
            ops.gemm_nt(dpre, t1, dh)
            del dgl, dpre
            dlat2 = self._ln_bwd(dh, lat2, ff[0], mh, rh, dlat, grads)
            # PerceiverAttention: lat2 = lat + softmax(q [k_x ; k_l]^T) [v_x ; v_l] Wout^T
            datt = torch.empty(Ml, I, device=dev, dtype=bf)
            ops.gemm_nt(dlat2, to, datt)
            dwo = torch.zeros(D, I, device=dev)
            ops.gemm_tn(dlat2, att, dwo, scratch=scr)
            dq = torch.empty(B, H, m, 64, device=dev, dtype=bf)
            dk, dv = torch.empty(B, H, Nk, 64, device=dev, dtype=bf), torch.empty(B, H, Nk, 64, device=dev, dtype=bf)
            ops.attn_bwd_ex(q, k, v, att, datt, lse, dq, dk, dv, B, H, m, Nk, 64, scale, kb)
            dqf, dkvl, dkvx = (torch.empty(r, c, device=dev, dtype=bf) for r, c in ((Ml, I), (Ml, 2 * I), (Mx, 2 * I)))
            ops.heads_merge_rope_bwd(dq, dqf, B, H, m, 0)
            ops.heads_merge_rope_bwd(dk, dkvl[:, :I], B, H, m, n)
            ops.heads_merge_rope_bwd(dv, dkvl[:, I:], B, H, m, n)
            ops.heads_merge_rope_bwd(dk, dkvx[:, :I], B, H, n, 0, cos, sin, rot)
            ops.heads_merge_rope_bwd(dv, dkvx[:, I:], B, H, n, 0)
            dwq, dwkv = torch.zeros(I, D, device=dev), torch.zeros(2 * I, D, device=dev)
            ops.gemm_tn(dqf, ln, dwq, scratch=scr)
            ops.gemm_tn(dkvx, xn, dwkv, scratch=scr)
            ops.gemm_tn(dkvl, ln, dwkv, scratch=scr)
            dln0, dln = torch.empty(Ml, D, device=dev, dtype=bf), torch.empty(Ml, D, device=dev, dtype=bf)
            ops.gemm_nt(dqf, tq, dln0)
            ops.gemm_nt(dkvl, tkv, dln, resid=dln0)
            dxn = torch.empty(Mx, D, device=dev, dtype=bf)
            ops.gemm_nt(dkvx, tkv, dxn)
            dlat = self._ln_bwd(dln, lat, attn.norm_latents, ml, rl, dlat2, grads)
            dx = self._ln_bwd(dxn, x2, attn.norm_x, mx, rx, dx, grads)
            for w, gw in ((attn.to_q.weight, dwq), (attn.to_kv.weight, dwkv), (attn.to_out.weight, dwo), (ff[1].weight, dw1),
                          (ff[3].weight, dw2)):
                grads[id(w)] = gw
        dlatents = torch.zeros(m * D, device=dev)
        ops.colsum(dlat.view(B, m * D), dlatents, B, m * D)
        grads[id(self.latents)] = dlatents.view(m, D)
        return dx, [grads[id(p)] for p in self.parameters()]

    def forward(self, x: Tensor, cos_sin: tuple[Tensor, Tensor] | None = None) -> Tensor:
        """x [B, n, dim] (n a square: the sqrt(n) x sqrt(n) token grid) -> latent tokens [B, num_latents, dim] (bf16)"""
        if cos_sin is not None:
            raise NotImplementedError("PerceiverResampler HIP path: the RoPE tables are built from the token grid (cos_sin=None)")
        return _ResamplerFn.apply(self, x, *self.parameters())
