"""Host-side runtime of the UNet denoiser (guided-diffusion style ``UNetModel``): parameter arena, bf16 weight shadows and
the forward / backward launch sequences over the C ABI.

Same MI355X-first choices as the DiT engine (engine.py): one flat f32 parameter arena + one flat gradient arena (one fused
AdamW launch, a handful of large RCCL all-reduces), the FiLM projections ``emb_layers.1`` of EVERY ResBlock stored back to
back so the conditioning of the whole network is ONE GEMM per step (silu(emb) is block-invariant), activations kept as NHWC
bf16 token rows so that 3x3 convolutions are im2col + MFMA GEMM with the residual add fused into the GEMM epilogue, and
1x1 convolutions are plain GEMMs.

Reference sites restated by the sequences below (``/root/reference/src/diffulab``): networks/denoisers/unet.py:593-745 (block
wiring), :832-853 (forward), :215-237 (ResBlock._forward), :296-322 (AttentionBlock._forward); networks/utils/nn.py:11-88
(GroupNorm32, Upsample, Downsample).  Only what ``configs/model/unet.yaml`` builds is covered: ``resblock_updown=True``,
``use_scale_shift_norm=True``, class-conditional or unconditional, self-attention blocks (no context embedder).
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
from torch import Tensor

from . import engine as _eng, tuning
from . import ops
from .engine import _rup


@dataclass
class UNetDims:
    image_size: tuple[int, int] = (32, 32)
    in_channels: int = 1
    model_channels: int = 128
    out_channels: int = 1
    num_res_blocks: int = 2
    attention_resolutions: tuple[int, ...] = (4, 8, 16)  # downsample FACTORS at which attention is inserted (unet.py:611)
    channel_mult: tuple[int, ...] = (1, 2, 4, 8)
    num_heads: int = 1
    use_scale_shift_norm: bool = False
    resblock_updown: bool = False
    conv_resample: bool = True  # read only when resblock_updown is False (unet.py:646,735)
    n_classes: int | None = None
    classifier_free: bool = False

    def validate(self) -> None:
        mc = self.model_channels
        if mc % 32:
            raise NotImplementedError("model_channels must be a multiple of 32 (GroupNorm32)")
        down = 2 ** (len(self.channel_mult) - 1)
        if self.image_size[0] % down or self.image_size[1] % down:
            raise NotImplementedError("image size must be divisible by 2^(levels-1)")
        ds = 1
        for level, mult in enumerate(self.channel_mult):
            if ds in self.attention_resolutions:
                if (mult * mc) % self.num_heads:
                    raise NotImplementedError("attention channels must be divisible by num_heads")
            if level != len(self.channel_mult) - 1:
                ds *= 2


@dataclass
class _Blk:
    kind: str  # "conv" | "res" | "attn" | "down" | "up" (the last two: Downsample / Upsample modules, nn.py:28-88)
    prefix: str
    cin: int = 0
    cout: int = 0
    up: bool = False
    down: bool = False
    emb_off: int = 0  # column offset of this ResBlock's [scale | shift] (or additive term) inside the stacked emb projection


@dataclass
class _Plan:
    input_blocks: list[list[_Blk]] = field(default_factory=list)
    middle: list[_Blk] = field(default_factory=list)
    output_blocks: list[list[_Blk]] = field(default_factory=list)
    final_ch: int = 0

    def all_blocks(self) -> list[_Blk]:
        return [b for grp in self.input_blocks + [self.middle] + self.output_blocks for b in grp]


def build_plan(d: UNetDims) -> _Plan:
    """module wiring of unet.py:593-745 without a context embedder"""
    mc = d.model_channels
    plan = _Plan()
    ch = int(d.channel_mult[0] * mc)
    plan.input_blocks.append([_Blk("conv", "input_blocks.0.0.", d.in_channels, ch)])
    chans, ds = [ch], 1
    for level, mult in enumerate(d.channel_mult):
        for _ in range(d.num_res_blocks):
            i = len(plan.input_blocks)
            layers = [_Blk("res", f"input_blocks.{i}.0.", ch, int(mult * mc))]
            ch = int(mult * mc)
            if ds in d.attention_resolutions:
                layers.append(_Blk("attn", f"input_blocks.{i}.1.", ch, ch))
            plan.input_blocks.append(layers)
            chans.append(ch)
        if level != len(d.channel_mult) - 1:
            i = len(plan.input_blocks)
            plan.input_blocks.append([_Blk("res", f"input_blocks.{i}.0.", ch, ch, down=True) if d.resblock_updown
                                      else _Blk("down", f"input_blocks.{i}.0.", ch, ch)])
            chans.append(ch)
            ds *= 2
    plan.middle = [_Blk("res", "middle_block.0.", ch, ch), _Blk("attn", "middle_block.1.", ch, ch),
                   _Blk("res", "middle_block.2.", ch, ch)]
    for level, mult in list(enumerate(d.channel_mult))[::-1]:
        for k in range(d.num_res_blocks + 1):
            ich = chans.pop()
            i = len(plan.output_blocks)
            layers = [_Blk("res", f"output_blocks.{i}.0.", ch + ich, int(mc * mult))]
            ch = int(mc * mult)
            if ds in d.attention_resolutions:
                layers.append(_Blk("attn", f"output_blocks.{i}.{len(layers)}.", ch, ch))
            if level and k == d.num_res_blocks:
                layers.append(_Blk("res", f"output_blocks.{i}.{len(layers)}.", ch, ch, up=True) if d.resblock_updown
                              else _Blk("up", f"output_blocks.{i}.{len(layers)}.", ch, ch))
                ds //= 2
            plan.output_blocks.append(layers)
    plan.final_ch = ch
    off = 0
    for b in plan.all_blocks():
        if b.kind == "res":
            b.emb_off = off
            off += emb_width(d, b)
    return plan


def emb_width(d: UNetDims, b: _Blk) -> int:
    """rows of emb_layers.1 (unet.py:161-166): [scale | shift] under FiLM, one additive term otherwise"""
    return 2 * b.cout if d.use_scale_shift_norm else b.cout


def resample_conv(b: _Blk) -> str:
    """parameter prefix of the 3x3 conv inside a Downsample (`op`, nn.py:79) / Upsample (`conv`, nn.py:47) module"""
    return b.prefix + ("op." if b.kind == "down" else "conv.")


class UNetLayout:
    """name -> (offset, shape) inside the flat f32 arena: the FiLM projections of all ResBlocks first (one [R, 4*mc] matrix and
    its bias), then everything else in module order; entries start on 256-byte boundaries."""

    def __init__(self, d: UNetDims, plan: _Plan) -> None:
        self.entries: dict[str, tuple[int, tuple[int, ...]]] = {}
        self.size = 0
        te = 4 * d.model_channels

        def add(name: str, shape: tuple[int, ...], align: int = 64) -> None:
            self.size = _rup(self.size, align)
            self.entries[name] = (self.size, shape)
            self.size += math.prod(shape)

        res = [b for b in plan.all_blocks() if b.kind == "res"]
        for i, b in enumerate(res):
            add(b.prefix + "emb_layers.1.weight", (emb_width(d, b), te), align=1 if i else 64)
        for i, b in enumerate(res):
            add(b.prefix + "emb_layers.1.bias", (emb_width(d, b),), align=1 if i else 64)
        self.emb_rows = sum(emb_width(d, b) for b in res)
        self.emb_w0, self.emb_b0 = res[0].prefix + "emb_layers.1.weight", res[0].prefix + "emb_layers.1.bias"
        add("time_embed.0.weight", (te, d.model_channels))
        add("time_embed.0.bias", (te,))
        add("time_embed.2.weight", (te, te))
        add("time_embed.2.bias", (te,))
        if d.n_classes is not None:
            add("label_embed.embedding.weight", (d.n_classes + (1 if d.classifier_free else 0), te))
        for b in plan.all_blocks():
            p = b.prefix
            if b.kind == "conv":
                add(p + "weight", (b.cout, b.cin, 3, 3))
                add(p + "bias", (b.cout,))
            elif b.kind == "res":
                add(p + "in_layers.0.weight", (b.cin,))
                add(p + "in_layers.0.bias", (b.cin,))
                add(p + "in_layers.2.weight", (b.cout, b.cin, 3, 3))
                add(p + "in_layers.2.bias", (b.cout,))
                add(p + "out_layers.0.weight", (b.cout,))
                add(p + "out_layers.0.bias", (b.cout,))
                add(p + "out_layers.3.weight", (b.cout, b.cout, 3, 3))
                add(p + "out_layers.3.bias", (b.cout,))
                if b.cin != b.cout:
                    add(p + "skip_connection.weight", (b.cout, b.cin, 1, 1))
                    add(p + "skip_connection.bias", (b.cout,))
            elif b.kind in ("down", "up"):
                if d.conv_resample:
                    add(resample_conv(b) + "weight", (b.cout, b.cin, 3, 3))
                    add(resample_conv(b) + "bias", (b.cout,))
            else:
                c = b.cin
                for n in ("norm_x", "norm_context"):
                    add(p + n + ".weight", (c,))
                    add(p + n + ".bias", (c,))
                add(p + "to_q.weight", (c, c, 1))
                add(p + "to_q.bias", (c,))
                add(p + "to_kv.weight", (2 * c, c, 1))
                add(p + "to_kv.bias", (2 * c,))
                add(p + "to_out.0.weight", (c, c, 1))
                add(p + "to_out.0.bias", (c,))
        add("out.0.weight", (plan.final_ch,))
        add("out.0.bias", (plan.final_ch,))
        add("out.2.weight", (d.out_channels, int(d.channel_mult[0] * d.model_channels), 3, 3))
        add("out.2.bias", (d.out_channels,))
        self.size = _rup(self.size, 64)

    def view(self, flat: Tensor, name: str) -> Tensor:
        off, shape = self.entries[name]
        return flat[off : off + math.prod(shape)].view(shape)


class UNetEngine:
    G = 32  # GroupNorm32
    precision = "bf16"

    def __init__(self, dims: UNetDims, device: torch.device | str = "cuda") -> None:
        dims.validate()
        self.d = dims
        self.dev = torch.device(device)
        self.plan = build_plan(dims)
        self.layout = UNetLayout(dims, self.plan)
        self.params: Tensor | None = None
        self.grads: Tensor | None = None
        self._shadow_key: tuple | None = None
        self.manual_version = 0
        self.param_version = 0  # see DiTEngine.refresh_shadows
        self.reducer = None  # optional training.dp.GradReducer
        self._scratch: dict[str, Tensor] = {}
        self._saved: dict | None = None
        self._zero = torch.zeros(64, device=self.dev, dtype=torch.bfloat16)  # out-of-image taps of the implicit-GEMM convs
        self.o = ops  # resampling / layout / attention entry points of the block orchestration (the fp32 engine swaps in f32 forms)
        self._build_shadows()

    # ------------------------------------------------------------------ parameters
    def bind(self, params: Tensor, grads: Tensor | None) -> None:
        assert params.dtype == torch.float32 and params.numel() == self.layout.size and params.is_cuda
        self.params, self.grads = params, grads
        self._shadow_key = None

    def P(self, name: str) -> Tensor:
        return self.layout.view(self.params, name)

    def Gr(self, name: str) -> Tensor:
        return self.layout.view(self.grads, name)

    def _build_shadows(self) -> None:
        d, dev = self.d, self.dev
        te = 4 * d.model_channels
        self.sh: dict[str, Tensor] = {}
        self._lin: list[tuple[str, tuple[int, int], bool, bool]] = []
        self._conv: list[tuple[str, int, int, bool]] = []

        def z(*shape: int) -> Tensor:
            return torch.zeros(*shape, device=dev, dtype=torch.bfloat16)

        def lin(name: str, R: int, C: int, fwd: bool = True, dgrad: bool = True) -> None:
            if fwd:
                self.sh[name + "|f"] = z(R, _rup(C, 64))
            if dgrad:
                self.sh[name + "|t"] = z(C, _rup(R, 64))
            self._lin.append((name, (R, C), fwd, dgrad))

        def conv(name: str, co: int, ci: int, dgrad: bool = True) -> None:
            self.sh[name + "|f"] = z(co, _rup(9 * ci, 64))
            self.sh[name + "|d"] = z(ci, _rup(9 * co, 64))  # tiny when unused (stem: ci = in_channels)
            self._conv.append((name, co, ci, dgrad))

        lin("@emb", self.layout.emb_rows, te)
        lin("time_embed.0.weight", te, d.model_channels, dgrad=False)
        lin("time_embed.2.weight", te, te)
        for b in self.plan.all_blocks():
            p = b.prefix
            if b.kind == "conv":
                conv(p + "weight", b.cout, b.cin, dgrad=False)
            elif b.kind == "res":
                conv(p + "in_layers.2.weight", b.cout, b.cin)
                conv(p + "out_layers.3.weight", b.cout, b.cout)
                if b.cin != b.cout:
                    lin(p + "skip_connection.weight", b.cout, b.cin)
            elif b.kind in ("down", "up"):
                if d.conv_resample:
                    conv(resample_conv(b) + "weight", b.cout, b.cin)
            else:
                c = b.cin
                lin(p + "to_q.weight", c, c)
                lin(p + "to_kv.weight", 2 * c, c)
                lin(p + "to_out.0.weight", c, c)
        conv("out.2.weight", d.out_channels, int(d.channel_mult[0] * d.model_channels))

    def refresh_shadows(self, force: bool = False) -> None:
        ver = 0 if self.params.is_inference() else self.params._version
        key = (self.params.data_ptr(), ver, self.manual_version, _eng._PARAM_EPOCH, self.param_version)
        if not force and key == self._shadow_key:
            return
        # every shadow of the network in TWO launches (one table of the linear weights, one of the 3x3 convolution weights; built once
        # per parameter arena) instead of one launch per weight: 107 launches of 3-130 us and their gaps on the main stream per step
        tkey = self.params.data_ptr()
        if getattr(self, "_cast_tables", (None,))[0] != tkey:
            lin_entries, conv_entries, loose = [], [], []
            for name, (R, C), fwd, dgrad in self._lin:
                if name == "@emb":
                    off = self.layout.entries[self.layout.emb_w0][0]
                    src = self.params[off : off + R * C].view(R, C)
                else:
                    src = self.P(name).view(R, C)
                lin_entries.append((src, self.sh[name + "|f"] if fwd else None, self.sh[name + "|t"] if dgrad else None, None))
            for name, co, ci, _ in self._conv:
                e = (self.P(name), self.sh[name + "|f"], self.sh[name + "|d"])
                (conv_entries if ops.ConvCastTable.accepts(*e) else loose).append(e)
            self._cast_tables = (tkey, ops.CastTable(lin_entries) if lin_entries else None,
                                 ops.ConvCastTable(conv_entries) if conv_entries else None, loose)
        _, lin_t, conv_t, loose = self._cast_tables
        if lin_t is not None:
            lin_t.run()
        if conv_t is not None:
            conv_t.run()
        for w_, wf_, wd_ in loose:  # (channel counts that are not multiples of 32: the 1-channel stem / head)
            ops.cast_conv3x3_weight(w_, wf_, wd_)
        self._shadow_key = key

    # ------------------------------------------------------------------ small helpers
    def _new(self, *shape: int, dtype=torch.bfloat16, zero: bool = False) -> Tensor:
        if torch.is_inference_mode_enabled():  # (samplers run under inference_mode; a training pass does not pay for the context)
            with torch.inference_mode(False):
                t = torch.empty(*shape, device=self.dev, dtype=dtype)
        else:
            t = torch.empty(*shape, device=self.dev, dtype=dtype)
        ops.keep(t)  # (a launch plan being recorded owns every buffer its calls name)
        return ops.zero_(t) if zero else t

    def _scr(self, key: str, numel: int, dtype=torch.bfloat16) -> Tensor:
        """grow-only scratch (im2col matrices, wgrad staging): reused by every conv, all launches are on one stream"""
        t = self._scratch.get(key)
        if t is None or t.numel() < numel or t.dtype != dtype:
            with torch.inference_mode(False):
                t = torch.empty(numel, device=self.dev, dtype=dtype)
            self._scratch[key] = t
        ops.keep(t)  # (a later, larger request replaces the entry: a recorded plan keeps the buffer it was recorded on)
        return t[:numel]

    def _splitk(self, M: int, n: int) -> Tensor | None:
        """f32 scratch (eight partial images) for the split-K path of the low-resolution convolutions (DL_UNET_SPLITK)"""
        if M > 16384 or not tuning.on("DL_UNET_SPLITK"):
            return None
        return self._scr("splitk", 8 * M * n, torch.float32)

    def _padded(self, x: Tensor, rows: int, cols: int) -> Tensor:
        """x [M, C] -> zero-padded [rows, cols] copy when the GEMM alignment (K % 64, reduction rows % 64) needs it"""
        if x.shape[0] == rows and x.shape[1] == cols:
            return x
        out = self._new(rows, cols, zero=True)
        ops.copy2d_bf16(x, out, x.shape[0], min(x.shape[1], cols))
        return out

    # ------------------------------------------------------------------ primitive forward / backward pairs
    def _conv3(self, x: Tensor, B: int, H: int, W: int, ci: int, name: str, co: int, resid: Tensor | None = None) -> Tensor:
        M = B * H * W
        out = self._new(M, _rup(co, 8), zero=bool(co % 8))
        bias = self.P(name[:-6] + "bias")
        if ops.conv3x3_nt(x, B, H, W, ci, self.sh[name + "|f"], out, co, bias, resid, self._zero, self._splitk(M, co)):
            return out  # implicit GEMM: no cols matrix
        Mp, ldk = _rup(M, 64), _rup(9 * ci, 64)
        cols = self._scr("cols", Mp * ldk).view(Mp, ldk)
        ops.im2col3x3(x, cols, B, H, W, ci)
        ops.gemm_nt(cols, self.sh[name + "|f"], out, bias=bias, resid=resid, M=M, N=co, K=ldk)
        return out

    def _off_chain(self, fn, *tensors: Tensor) -> None:
        """run `fn` (launches whose results nothing downstream in the backward reads: bias / weight gradients) on the side HIP
        stream, after everything issued so far on the main stream; `tensors` are the main-stream buffers it reads (kept from being
        recycled by the allocator until the side stream is past them).  DL_UNET_SIDE=0 keeps everything on one stream."""
        if not self._use_side:
            fn()
            return
        main = ops._s()
        side = self._side_stream()
        if not ops.is_recording():  # (a plan owns its buffers: nothing is recycled under it)
            for t in tensors:
                t.record_stream(side)
        ops.stream_wait(side.cuda_stream, main)  # (an event record + stream wait behind the C ABI: a recordable call)
        with torch.cuda.stream(side):
            fn()

    def _side_stream(self) -> "torch.cuda.Stream":
        if getattr(self, "_side", None) is None:
            mask = tuning.text("DL_SIDE_CU_MASK")  # (experiments: "b192" = the first 192 CUs, "i2" = every other CU)
            if mask:
                self._side = ops.masked_stream(mask, self.dev)
            elif tuning.on("DL_UNET_SIDE_LOW_PRIORITY"):
                self._side = ops.low_priority_stream(self.dev)
            else:
                self._side = torch.cuda.Stream(device=self.dev)
        return self._side

    @property
    def _use_side(self) -> bool:
        return tuning.on("DL_UNET_SIDE")

    def _bias_grad(self, dy: Tensor, bname: str, M: int, co: int) -> None:
        """bias gradient = column sum of dy.  Default: deferred -- the (dy, gradient) pair waits with the staged weight gradients and a
        stretch of the backward's column sums runs as ONE launch (_fold_staged; 105 launches of 5-40 us per step on the side stream
        otherwise).  DL_UNET_DET_COLSUM=1: at once, through row-slab partials + a fixed-order fold (bit-reproducible, two launches per
        bias: measured 2 % slower per step)."""
        if tuning.on("DL_UNET_DET_COLSUM"):
            ops.colsum(dy, self.Gr(bname), M, co, scratch=self._scr("colsum_part", 512 * co, torch.float32))
        elif tuning.on("DL_UNET_FOLD_BATCHED") and dy.dtype == torch.bfloat16:
            self.__dict__.setdefault("_colsum_pending", []).append((dy, self.Gr(bname), M, co))
        else:
            ops.colsum(dy, self.Gr(bname), M, co)

    def _wgrad_stage(self, name: str, ldk: int, co: int, ci: int, R: int, H: int, W: int) -> Tensor | None:
        """this convolution's weight-gradient stage, or None.  Default (DL_UNET_WGRAD_PARTS): the partial images [n_parts, ldk, co]
        f32 the R-splits of dl_conv3x3_wgrad_tn_parts STORE (no atomics, nothing to zero; the batched fold adds them in image order:
        bit-reproducible); shapes the implicit-GEMM kernel does not take (Ci % 128) and DL_UNET_WGRAD_PARTS=0: one [ldk, co] image
        the splits add into with f32 atomics (zero between backwards: the fold clears it)."""
        if not (tuning.on("DL_UNET_FOLD_BATCHED") and ops.ConvFoldTable.accepts(co, ci)):
            return None
        nparts = 0
        if tuning.on("DL_UNET_WGRAD_PARTS"):
            cache = self.__dict__.setdefault("_nparts", {})
            nparts = cache.get((H, W, ci, co, R))
            if nparts is None:
                nparts = cache[(H, W, ci, co, R)] = ops.conv3x3_wgrad_nparts(H, W, ci, co, R, tuning.integer("DL_UNET_WGRAD_WGS", 0))
        st = self.__dict__.setdefault("_stage", {})
        key = (name, nparts)
        g = st.get(key)
        if g is None:
            g = st[key] = self._new(nparts, ldk, co, dtype=torch.float32) if nparts else self._new(ldk, co, dtype=torch.float32, zero=True)
        ops.keep(g)
        self.__dict__.setdefault("_stage_pending", []).append(key)
        return g

    def _fold_staged(self, min_pending: int = 1) -> None:
        """every weight gradient staged since the last call into its parameter's gradient, one launch (which also clears the stage).
        Called after each block group of the backward once `min_pending` convolutions wait (the folds then run under the rest of the
        backward: only the last, small batch -- the high-resolution input blocks -- is left for the end) and once at the end."""
        pend = self.__dict__.get("_stage_pending")
        cs = self.__dict__.get("_colsum_pending")
        # only column sums wait: the end of a backward without staged convolutions -- or channel counts ConvFoldTable does not accept,
        # where nothing is ever staged and the pending (dy, gradient) pairs would keep every output gradient of the backward alive
        # until the end: flush them once 16 wait, whatever `min_pending` says
        if cs and not pend and (min_pending <= 1 or len(cs) >= 16):
            self._colsum_pending = []
            if self._use_side:
                with torch.cuda.stream(self._side_stream()):
                    ops.colsum_batched(cs)
            else:
                ops.colsum_batched(cs)
            return
        if not pend or len(pend) < min_pending:
            return
        key = (self.grads.data_ptr(), tuple(pend))
        tables = self.__dict__.setdefault("_fold_tables", {})
        table = tables.get(key)
        if table is None:
            if len(tables) > 64:  # (a re-bound gradient arena: the old tables hold dead pointers)
                tables.clear()
            table = tables[key] = ops.ConvFoldTable([(self._stage[k], self.Gr(k[0])) for k in pend])
        ops.keep(table)
        self._stage_pending = []
        self._colsum_pending = []
        if self._use_side:
            with torch.cuda.stream(self._side_stream()):
                if cs:
                    ops.colsum_batched(cs)
                table.run(clear=True)
        else:
            if cs:
                ops.colsum_batched(cs)
            table.run(clear=True)

    def _conv3_bwd(self, dy: Tensor, x: Tensor, B: int, H: int, W: int, ci: int, name: str, co: int,
                   need_dx: bool = True) -> Tensor | None:
        M = B * H * W
        Mp, ldk, co8 = _rup(M, 64), _rup(9 * ci, 64), _rup(co, 8)
        dyp = self._padded(dy, Mp, co8)

        def wgrad() -> None:  # bias + weight gradient: off the dependency chain of the backward
            # (DL_UNET_DET_COLSUM=1: row-slab partials + fixed-order fold instead of f32 atomics -- reproducible, but two launches per
            #  bias gradient: 35.4 vs 34.7 ms per MNIST-DDPM step, so the atomic form stays the default here)
            self._bias_grad(dy, name[:-6] + "bias", M, co)
            # the gradient lands transposed, [(tap, ci), co] f32 (9*Ci rows fit the 384-row wgrad tiles), in this convolution's slice of
            # a persistent staging arena; ONE launch at the end of the backward folds every slice into its [Co, Ci, 3, 3] gradient and
            # clears it (_fold_staged).  Channel counts off 32 (the first / last convolution): a temporary stage, folded here.
            g = self._wgrad_stage(name, ldk, co, ci, Mp, H, W)
            staged = g is not None
            if staged and g.dim() == 3:  # partial images: plain stores, folded in a fixed order
                ops.conv3x3_wgrad_tn_parts(x, B, H, W, ci, dyp, co8, g, self._zero, max_wgs=tuning.integer("DL_UNET_WGRAD_WGS", 0))
                return
            if not staged:
                g = self._new(ldk, co8, dtype=torch.float32, zero=True)
            if not ops.conv3x3_wgrad_tn(x, B, H, W, ci, dyp, co8, g, self._zero, max_wgs=tuning.integer("DL_UNET_WGRAD_WGS", 0)):
                cols = self._new(Mp, ldk)
                ops.im2col3x3(x, cols, B, H, W, ci)
                ops.gemm_tn(cols, dyp, g, M=ldk, N=co8)
            if not staged:
                ops.conv3x3_wgrad_fold(g, self.Gr(name))

        self._off_chain(wgrad, dy, dyp, x)
        if not need_dx:
            return None
        dx = self._new(M, ci)
        if ops.conv3x3_nt(dy, B, H, W, co, self.sh[name + "|d"], dx, ci, None, None, self._zero, self._splitk(M, ci)):
            return dx
        ldd = _rup(9 * co, 64)
        dcols = self._scr("cols", Mp * ldd).view(Mp, ldd)
        ops.im2col3x3(dy, dcols, B, H, W, co)
        ops.gemm_nt(dcols, self.sh[name + "|d"], dx, M=M, N=ci, K=ldd)
        return dx

    def _lin_fwd(self, x: Tensor, name: str, co: int, ci: int, resid: Tensor | None = None, out: Tensor | None = None) -> Tensor:
        M = x.shape[0]
        K = _rup(ci, 64)
        xp = self._padded(x, M, K)
        out = self._new(M, co) if out is None else out
        ops.gemm_nt(xp, self.sh[name + "|f"], out, bias=self.P(name[:-6] + "bias"), resid=resid, M=M, N=co, K=K)
        return out

    def _lin_fwd_pair(self, x0: Tensor, name0: str, co0: int, x1: Tensor, name1: str, co1: int, ci: int) -> tuple[Tensor, Tensor]:
        """two Linear forwards on different inputs as ONE launch where both are small (ops.gemm_nt_pair: the q and kv projections of an
        AttentionBlock at the low-resolution levels are 64 + 128 tiles on 256 CUs); bit-identical to two _lin_fwd calls"""
        if not tuning.on("DL_UNET_NT_PAIR"):
            return self._lin_fwd(x0, name0, co0, ci), self._lin_fwd(x1, name1, co1, ci)
        M, K = x0.shape[0], _rup(ci, 64)
        xp0, xp1 = self._padded(x0, M, K), self._padded(x1, M, K)
        o0, o1 = self._new(M, co0), self._new(M, co1)
        ops.gemm_nt_pair([(xp0, self.sh[name0 + "|f"], o0, self.P(name0[:-6] + "bias"), None, M, co0, K),
                          (xp1, self.sh[name1 + "|f"], o1, self.P(name1[:-6] + "bias"), None, M, co1, K)])
        return o0, o1

    def _lin_bwd_pair(self, dy0: Tensor, x0: Tensor, name0: str, co0: int, dy1: Tensor, x1: Tensor, name1: str, co1: int,
                      ci: int) -> tuple[Tensor, Tensor]:
        """the backward of _lin_fwd_pair: both weight gradients off the chain as before, the two data gradients as one launch"""
        if not tuning.on("DL_UNET_NT_PAIR"):
            return self._lin_bwd(dy0, x0, name0, co0, ci), self._lin_bwd(dy1, x1, name1, co1, ci)
        self._lin_bwd(dy0, x0, name0, co0, ci, need_dx=False)
        self._lin_bwd(dy1, x1, name1, co1, ci, need_dx=False)
        M = dy0.shape[0]
        K0, K1 = _rup(co0, 64), _rup(co1, 64)
        d0, d1 = self._new(M, ci), self._new(M, ci)
        ops.gemm_nt_pair([(self._padded(dy0, M, K0), self.sh[name0 + "|t"], d0, None, None, M, ci, K0),
                          (self._padded(dy1, M, K1), self.sh[name1 + "|t"], d1, None, None, M, ci, K1)])
        return d0, d1

    def _lin_bwd(self, dy: Tensor, x: Tensor, name: str, co: int, ci: int, need_dx: bool = True) -> Tensor | None:
        M = dy.shape[0]
        Mp = _rup(M, 64)
        dyp, xp = self._padded(dy, Mp, dy.shape[1]), self._padded(x, Mp, x.shape[1])

        def wgrad() -> None:
            # (DL_UNET_DET_COLSUM=1: row-slab partials + fixed-order fold instead of f32 atomics -- reproducible, but two launches per
            #  bias gradient: 35.4 vs 34.7 ms per MNIST-DDPM step, so the atomic form stays the default here)
            self._bias_grad(dy, name[:-6] + "bias", M, co)
            ops.gemm_tn(dyp, xp, self.Gr(name).view(co, ci), M=co, N=ci)

        self._off_chain(wgrad, dy, dyp, xp)
        if not need_dx:
            return None
        K = _rup(co, 64)
        dyk = self._padded(dy, M, K)
        dx = self._new(M, ci)
        ops.gemm_nt(dyk, self.sh[name + "|t"], dx, M=M, N=ci, K=K)
        return dx

    def _gn(self, x: Tensor, B: int, HW: int, C: int, wname: str, film=None, silu: bool = True, stats: Tensor | None = None):
        out = self._new(B * HW, C)
        fs, fh = film if film is not None else (None, None)
        if stats is None:
            stats = self._new(B, self.G, 2, dtype=torch.float32)
            ops.gn_fwd(x, stats, self.P(wname + "weight"), self.P(wname + "bias"), fs, fh, silu, out, B, HW, C, self.G)
            return out, stats
        ops.gn_apply_fwd(x, stats, self.P(wname + "weight"), self.P(wname + "bias"), fs, fh, silu, out, B, HW, C, self.G)
        return out, stats

    def _gn_bwd(self, dout: Tensor, x: Tensor, stats: Tensor, B: int, HW: int, C: int, wname: str, film=None, dfilm=None,
                silu: bool = True, dres: Tensor | None = None) -> Tensor:
        dx = self._new(B * HW, C)
        fs, fh = film if film is not None else (None, None)
        dfs, dfh = dfilm if dfilm is not None else (None, None)
        scr = self._scr("gn", 8 * B * 4 * C + B * self.G * 2, torch.float32)  # DL_GN_BWD_MAX_RANGES partial-sum slabs
        ops.gn_bwd(dout, x, stats, self.P(wname + "weight"), self.P(wname + "bias"), fs, fh, silu, dres, dx,
                   self.Gr(wname + "weight"), self.Gr(wname + "bias"), dfs, dfh, scr, B, HW, C, self.G)
        return dx

    def _add(self, a: Tensor | None, b: Tensor | None) -> Tensor | None:
        if a is None:
            return b
        if b is None:
            return a
        out = self._new(*a.shape)
        ops.add_bf16(a, b, out)
        return out

    # ------------------------------------------------------------------ blocks
    def _res_fwd(self, b: _Blk, x: Tensor, B: int, H: int, W: int, eo: Tensor, save: list | None):
        p = b.prefix
        h, st1 = self._gn(x, B, H * W, b.cin, p + "in_layers.0.")
        H2, W2, x2 = H, W, x
        if b.up or b.down:
            H2, W2 = (2 * H, 2 * W) if b.up else (H // 2, W // 2)
            hp, x2 = self._new(B * H2 * W2, b.cin), self._new(B * H2 * W2, b.cin)
            if b.up:
                self.o.resample2x2_pair(h, hp, x, x2, B, H, W, b.cin, 1.0, True)
            else:
                self.o.resample2x2_pair(h, hp, x, x2, B, H2, W2, b.cin, 0.25, False)
            h = hp
        h2 = self._conv3(h, B, H2, W2, b.cin, p + "in_layers.2.weight", b.cout)
        if self.d.use_scale_shift_norm:
            film = (eo[:, b.emb_off : b.emb_off + b.cout], eo[:, b.emb_off + b.cout : b.emb_off + 2 * b.cout])
        else:  # h + emb_out in front of the plain GroupNorm -> SiLU (unet.py:235-237)
            film, pre = None, h2
            h2 = self._new(B * H2 * W2, b.cout)
            self.o.rowbias_add(pre, eo[:, b.emb_off : b.emb_off + b.cout], h2, B, H2 * W2, b.cout)
        h3, st2 = self._gn(h2, B, H2 * W2, b.cout, p + "out_layers.0.", film=film)
        skip = x2 if b.cin == b.cout else self._lin_fwd(x2, p + "skip_connection.weight", b.cout, b.cin)
        out = self._conv3(h3, B, H2, W2, b.cout, p + "out_layers.3.weight", b.cout, resid=skip)  # x + h fused (unet.py:237)
        if save is not None:
            save.append((x, st1, h, h2, st2, h3, x2, H, W, H2, W2))
        return out, H2, W2

    def _res_bwd(self, b: _Blk, dout: Tensor, B: int, saved, eo: Tensor, deo: Tensor) -> Tensor:
        p = b.prefix
        x, st1, h, h2, st2, h3, x2, H, W, H2, W2 = saved
        dh3 = self._conv3_bwd(dout, h3, B, H2, W2, b.cout, p + "out_layers.3.weight", b.cout)
        dx2 = dout if b.cin == b.cout else self._lin_bwd(dout, x2, p + "skip_connection.weight", b.cout, b.cin)
        o = b.emb_off
        if self.d.use_scale_shift_norm:
            film = (eo[:, o : o + b.cout], eo[:, o + b.cout : o + 2 * b.cout])
            dfilm = (deo[:, o : o + b.cout], deo[:, o + b.cout : o + 2 * b.cout])
            dh2 = self._gn_bwd(dh3, h2, st2, B, H2 * W2, b.cout, p + "out_layers.0.", film=film, dfilm=dfilm)
        else:  # h2 is the conv output + emb_out: the GroupNorm gradient is both d(conv output) and, summed over pixels, d(emb_out)
            dh2 = self._gn_bwd(dh3, h2, st2, B, H2 * W2, b.cout, p + "out_layers.0.")
            self.o.rowbias_bwd(dh2, deo[:, o : o + b.cout], B, H2 * W2, b.cout)
        dh = self._conv3_bwd(dh2, h, B, H2, W2, b.cin, p + "in_layers.2.weight", b.cout)
        if b.up or b.down:
            dhp, dxs = self._new(B * H * W, b.cin), self._new(B * H * W, b.cin)
            if b.up:  # backward of nearest upsample: sum of the 2x2 window
                self.o.resample2x2_pair(dh, dhp, dx2, dxs, B, H, W, b.cin, 1.0, False)
            else:  # backward of avg-pool: broadcast / 4
                self.o.resample2x2_pair(dh, dhp, dx2, dxs, B, H2, W2, b.cin, 0.25, True)
            dh, dx2 = dhp, dxs
        return self._gn_bwd(dh, x, st1, B, H * W, b.cin, p + "in_layers.0.", dres=dx2)

    def _resample_fwd(self, b: _Blk, x: Tensor, B: int, H: int, W: int, save: list | None):
        """Downsample / Upsample modules (nn.py:28-88) of resblock_updown=False.  The stride-2 conv is the stride-1 implicit GEMM
        at the input resolution sampled at the even pixels (dl_pick2x2): 4x the needed MACs on the L-1 downsampling convs of a
        network, in exchange for one conv kernel family."""
        c, name = b.cin, resample_conv(b) + "weight"
        if b.kind == "down":
            H2, W2 = H // 2, W // 2
            out = self._new(B * H2 * W2, c)
            if self.d.conv_resample:
                self.o.pick2x2(self._conv3(x, B, H, W, c, name, c), out, B, H2, W2, c)
            else:
                self.o.reduce2x2(x, out, B, H2, W2, c, 0.25)
            xin = x
        else:
            H2, W2 = 2 * H, 2 * W
            xin = self._new(B * H2 * W2, c)
            self.o.expand2x2(x, xin, B, H, W, c, 1.0)
            out = self._conv3(xin, B, H2, W2, c, name, c) if self.d.conv_resample else xin
        if save is not None:
            save.append((xin if self.d.conv_resample else None, H, W))
        return out, H2, W2

    def _resample_bwd(self, b: _Blk, dout: Tensor, B: int, saved) -> Tensor:
        c, name = b.cin, resample_conv(b) + "weight"
        xin, H, W = saved
        dx = self._new(B * H * W, c)
        if b.kind == "down":
            if self.d.conv_resample:
                dfull = self._new(B * H * W, c)
                self.o.stuff2x2(dout, dfull, B, H // 2, W // 2, c)
                return self._conv3_bwd(dfull, xin, B, H, W, c, name, c)
            self.o.expand2x2(dout, dx, B, H // 2, W // 2, c, 0.25)
            return dx
        if self.d.conv_resample:
            dout = self._conv3_bwd(dout, xin, B, 2 * H, 2 * W, c, name, c)
        self.o.reduce2x2(dout, dx, B, H, W, c, 1.0)
        return dx

    # attention core: softmax(scale Q K^T) V per (sample, head), heads = column blocks of the token rows.  Up to 64 tokens (the
    # configurations of the reference: 8 x 8 maps and below) one workgroup per (sample, head) holds the whole problem
    # (dl_attn_small_*); larger maps -- attention at 16 x 16 / 32 x 32, any head width -- go through the exact-f32 batched GEMMs
    # + row softmax of the fp32 regime on f32 copies of q / kv (ops.f32_attn_*): general, and more precise than the bf16 regime asks
    def _attn_small_ok(self, n: int, dh: int) -> bool:
        return self.precision == "fp32" or (n <= 64 and dh % 8 == 0)

    def _attn_core_fwd(self, q: Tensor, kv: Tensor, att: Tensor, probs: Tensor, B: int, n: int, nh: int, c: int) -> None:
        if self._attn_small_ok(n, c // nh):
            self.o.attn_small_fwd(q, kv[:, :c], kv[:, c:], att, probs, B, n, nh, c // nh)
            return
        qf = self._scr("attn_qf", B * n * c, torch.float32).view(B * n, c)
        kvf = self._scr("attn_kvf", B * n * 2 * c, torch.float32).view(B * n, 2 * c)
        of = self._scr("attn_of", B * n * c, torch.float32).view(B * n, c)
        ops.cast_bf16_to_f32(q, qf)
        ops.cast_bf16_to_f32(kv, kvf)
        ops.f32_attn_fwd(qf, kvf[:, :c], kvf[:, c:], of, probs, B, n, nh, c // nh)
        ops.cast_f32_to_bf16(of, att)

    def _attn_core_bwd(self, q: Tensor, kv: Tensor, datt: Tensor, probs: Tensor, dq: Tensor, dkv: Tensor, B: int, n: int, nh: int,
                       c: int) -> None:
        if self._attn_small_ok(n, c // nh):
            self.o.attn_small_bwd(q, kv[:, :c], kv[:, c:], datt, probs, dq, dkv[:, :c], dkv[:, c:], B, n, nh, c // nh)
            return
        qf = self._scr("attn_qf", B * n * c, torch.float32).view(B * n, c)
        kvf = self._scr("attn_kvf", B * n * 2 * c, torch.float32).view(B * n, 2 * c)
        df = self._scr("attn_of", B * n * c, torch.float32).view(B * n, c)
        dqf = self._scr("attn_dqf", B * n * c, torch.float32).view(B * n, c)
        dkvf = self._scr("attn_dkvf", B * n * 2 * c, torch.float32).view(B * n, 2 * c)
        dP = self._scr("attn_dp", B * nh * n * n, torch.float32).view(B, nh, n, n)
        ops.cast_bf16_to_f32(q, qf)
        ops.cast_bf16_to_f32(kv, kvf)
        ops.cast_bf16_to_f32(datt, df)
        ops.f32_attn_bwd(qf, kvf[:, :c], kvf[:, c:], df, probs, dP, dqf, dkvf[:, :c], dkvf[:, c:], B, n, nh, c // nh)
        ops.cast_f32_to_bf16(dqf, dq)
        ops.cast_f32_to_bf16(dkvf, dkv)

    def _attn_fwd(self, b: _Blk, x: Tensor, B: int, H: int, W: int, save: list | None) -> Tensor:
        p, c, n = b.prefix, b.cin, H * W
        nh = self.d.num_heads
        nx, st = self._gn(x, B, n, c, p + "norm_x.", silu=False)
        nc, _ = self._gn(x, B, n, c, p + "norm_context.", silu=False, stats=st)  # context = x: same statistics
        q, kv = self._lin_fwd_pair(nx, p + "to_q.weight", c, nc, p + "to_kv.weight", 2 * c, c)
        att = self._new(B * n, c)
        probs = self._new(B, nh, n, n, dtype=torch.float32)
        self._attn_core_fwd(q, kv, att, probs, B, n, nh, c)
        out = self._lin_fwd(att, p + "to_out.0.weight", c, c, resid=x)
        if save is not None:
            save.append((x, st, nx, nc, q, kv, att, probs, H, W))
        return out

    def _attn_bwd(self, b: _Blk, dout: Tensor, B: int, saved) -> Tensor:
        p, c = b.prefix, b.cin
        nh = self.d.num_heads
        x, st, nx, nc, q, kv, att, probs, H, W = saved
        n = H * W
        datt = self._lin_bwd(dout, att, p + "to_out.0.weight", c, c)
        dq, dkv = self._new(B * n, c), self._new(B * n, 2 * c)
        self._attn_core_bwd(q, kv, datt, probs, dq, dkv, B, n, nh, c)
        dnx, dnc = self._lin_bwd_pair(dq, nx, p + "to_q.weight", c, dkv, nc, p + "to_kv.weight", 2 * c, c)
        dx = self._gn_bwd(dnx, x, st, B, n, c, p + "norm_x.", silu=False, dres=dout)
        return self._gn_bwd(dnc, x, st, B, n, c, p + "norm_context.", silu=False, dres=dx)

    def _group_fwd(self, blocks: list[_Blk], h: Tensor, B: int, H: int, W: int, eo: Tensor, save: list | None):
        for b in blocks:
            if b.kind == "conv":
                if save is not None:
                    save.append((h, H, W))
                h = self._conv3(h, B, H, W, b.cin, b.prefix + "weight", b.cout)
            elif b.kind == "res":
                h, H, W = self._res_fwd(b, h, B, H, W, eo, save)
            elif b.kind in ("down", "up"):
                h, H, W = self._resample_fwd(b, h, B, H, W, save)
            else:
                h = self._attn_fwd(b, h, B, H, W, save)
        return h, H, W

    def _group_bwd(self, blocks: list[_Blk], dh: Tensor, B: int, save: list, eo: Tensor, deo: Tensor) -> Tensor | None:
        for b in blocks[::-1]:
            s = save.pop()
            if b.kind == "conv":
                x, H, W = s
                return self._conv3_bwd(dh, x, B, H, W, b.cin, b.prefix + "weight", b.cout, need_dx=False)
            if b.kind == "res":
                dh = self._res_bwd(b, dh, B, s, eo, deo)
            elif b.kind in ("down", "up"):
                dh = self._resample_bwd(b, dh, B, s)
            else:
                dh = self._attn_bwd(b, dh, B, s)
        return dh

    def _cond_fwd(self, t: Tensor, y_eff: Tensor | None, B: int):
        """emb = time_embed(timestep_embedding(t)) + label_embed(y); returns the stacked FiLM / additive projections of every ResBlock
        (eo [B, >= emb_rows]) and the tensors the conditioning backward needs"""
        d = self.d
        mc, te = d.model_channels, 4 * d.model_channels
        Bp = _rup(B, 64)
        # conditioning: emb = time_embed(timestep_embedding(t)) + label_embed(y) ; every ResBlock consumes silu(emb)
        temb = self._new(Bp, _rup(mc, 64), zero=True)
        if mc % 64 == 0:
            ops.timestep_embedding(t, temb[:B])
        else:  # computed dense, then copied into the zero-padded GEMM operand
            dense = self._new(B, mc)
            ops.timestep_embedding(t, dense)
            ops.copy2d_bf16(dense, temb, B, mc)
        pre1, h1 = self._new(Bp, te, zero=True), self._new(Bp, te, zero=True)
        ops.gemm_nt(temb, self.sh["time_embed.0.weight|f"], h1, bias=self.P("time_embed.0.bias"), act=ops.ACT_SILU, pre_out=pre1,
                    M=B, N=te, K=_rup(mc, 64))
        e = self._new(B, te, dtype=torch.float32)
        ops.gemm_nt(h1, self.sh["time_embed.2.weight|f"], e, bias=self.P("time_embed.2.bias"), M=B, N=te, K=_rup(te, 64))
        table = self.P("label_embed.embedding.weight") if d.n_classes is not None else None
        emb = self._new(B, te, dtype=torch.float32)
        se = self._new(Bp, te, zero=True)
        ops.cond_combine_fwd(e, table, y_eff if table is not None else None, emb, se[:B])
        R = self.layout.emb_rows
        off = self.layout.entries[self.layout.emb_b0][0]
        eo = self._new(B, _rup(R, 64))  # (additive conditioning: R is a multiple of 32 only; the row stride stays GEMM-aligned)
        ops.gemm_nt(se, self.sh["@emb|f"], eo, bias=self.params[off : off + R], M=B, N=R, K=_rup(te, 64))

        return eo, dict(temb=temb, pre1=pre1, h1=h1, emb=emb, se=se)

    # ------------------------------------------------------------------ forward (unet.py:832-853)
    # ---- launch plans (ops.LaunchPlan): a training pass is recorded on its second run for a given shape and re-issued as a flat list
    #      of C calls afterwards.  The step of this network is ~716 launches; walking the engine's Python for them costs the host
    #      ~17 ms, more than the GPU needs for the step at the configured batch of 64.
    def _plans_on(self) -> bool:
        return type(self) is UNetEngine and tuning.on("DL_LAUNCH_PLAN")

    def forward(self, x: Tensor, t: Tensor, y_eff: Tensor | None, train: bool, refresh: bool = True) -> Tensor:
        """x f32 [B, in_channels, H, W], t f32 [B], y_eff int64 [B] or None -> prediction f32 [B, out_channels, H, W]"""
        if not (train and self._plans_on()) or ops.is_recording() or torch.is_inference_mode_enabled() or torch.cuda.is_current_stream_capturing():
            return self._forward_pass(x, t, y_eff, train, refresh)
        if refresh:
            self.refresh_shadows(force=True)  # (outside the plan: whether the shadows are stale is per-call state)
        key = (tuple(x.shape), x.dtype, t.dtype, None if y_eff is None else y_eff.dtype, ops._s(), self.params.data_ptr(),
               0 if self.grads is None else self.grads.data_ptr(), id(self.reducer))
        plans = self.__dict__.setdefault("_launch_plans", {})
        ent = plans.get(key)
        if ent is None or ent is False:  # first pass of this shape: eager (scratch buffers, tables and stages come into being)
            if ent is None:
                if len(plans) >= 4:  # (a plan holds a whole step's buffers: keep few)
                    plans.clear()
                plans[key] = 1
            self._cur_plan = None
            return self._forward_pass(x, t, y_eff, True, False)
        if ent == 1:  # second pass: record it
            lp = ops.LaunchPlan()
            lp.inputs = (x.detach().clone(), t.detach().clone(), None if y_eff is None else y_eff.detach().clone())
            try:
                with ops.recording(lp):
                    lp.out = self._forward_pass(*lp.inputs, True, False)
            except Exception:
                plans[key] = False
                raise
            lp.state = self._saved
            plans[key] = self._cur_plan = lp
            return lp.out
        lp = ent
        xs, ts, ys = lp.inputs
        xs.copy_(x)
        ts.copy_(t)
        if ys is not None:
            ys.copy_(y_eff)
        lp.replay()
        self._saved, self._cur_plan = lp.state, lp
        return lp.out

    def backward(self, dpred: Tensor, dfeats: dict | None = None) -> None:
        """dpred f32 [B, out_channels, H, W]; accumulates every parameter gradient into the gradient arena"""
        lp = self.__dict__.get("_cur_plan")
        if lp is None or lp.state is not self._saved or self._saved is None or ops.is_recording():
            return self._backward_pass(dpred, dfeats)
        assert not dfeats, "the UNet engine exposes no intermediate features"
        if lp.bwd is None:  # the backward that belongs to the recorded forward: record it on the forward's own buffers
            bp = ops.LaunchPlan()
            bp.inputs = dpred.detach().clone()
            try:
                with ops.recording(bp):
                    self._backward_pass(bp.inputs, None)
            except Exception:
                self.__dict__["_launch_plans"] = {k: (False if v is lp else v) for k, v in self.__dict__.get("_launch_plans", {}).items()}
                self._cur_plan = None
                raise
            lp.bwd = bp
            return
        lp.bwd.inputs.copy_(dpred)
        lp.bwd.replay()
        self._saved = None

    def _forward_pass(self, x: Tensor, t: Tensor, y_eff: Tensor | None, train: bool, refresh: bool = True) -> Tensor:
        d, plan = self.d, self.plan
        B, Cin, H, W = x.shape
        assert (H, W) == tuple(d.image_size) and Cin == d.in_channels
        if refresh:
            self.refresh_shadows(force=train)
        mc, te = d.model_channels, 4 * d.model_channels
        Bp = _rup(B, 64)
        save: list | None = [] if train else None

        eo, cs = self._cond_fwd(t, y_eff, B)

        h = self._new(B * H * W, Cin)
        self.o.nchw_to_nhwc(x, h, B, Cin, H * W)
        hs: list[tuple[Tensor, int]] = []
        for grp in plan.input_blocks:
            h, H, W = self._group_fwd(grp, h, B, H, W, eo, save)
            hs.append((h, grp[-1].cout))
        h, H, W = self._group_fwd(plan.middle, h, B, H, W, eo, save)
        ch = plan.middle[-1].cout
        for grp in plan.output_blocks:
            skip, ich = hs.pop()
            cat = self._new(B * H * W, ch + ich)  # torch.cat([h, hs.pop()], dim=1)
            self.o.copy2d_bf16(h, cat[:, :ch], B * H * W, ch)
            self.o.copy2d_bf16(skip, cat[:, ch:], B * H * W, ich)
            h, H, W = self._group_fwd(grp, cat, B, H, W, eo, save)
            ch = grp[-1].cout
        hf, stf = self._gn(h, B, H * W, ch, "out.0.")
        o = self._conv3(hf, B, H, W, ch, "out.2.weight", d.out_channels)
        pred = self._new(B, d.out_channels, H, W, dtype=torch.float32)
        self.o.nhwc_to_nchw(o, pred, B, d.out_channels, H * W)
        if train:
            self._saved = dict(B=B, H=H, W=W, save=save, eo=eo, y=y_eff, h=h, stf=stf, hf=hf, **cs)
        return pred

    # ------------------------------------------------------------------ backward
    def _backward_pass(self, dpred: Tensor, dfeats: dict | None = None) -> None:
        assert not dfeats, "the UNet engine exposes no intermediate features"
        s, d, plan = self._saved, self.d, self.plan
        assert s is not None, "backward without a train-mode forward"
        self._saved = None
        if self.__dict__.get("_stage_pending") or self.__dict__.get("_colsum_pending"):
            # a backward that raised half-way left staged gradients behind: drop the references, zero the stage
            self._stage_pending, self._colsum_pending = [], []
            for g in self.__dict__.get("_stage", {}).values():
                g.zero_()
        B, H, W, save, eo = s["B"], s["H"], s["W"], s["save"], s["eo"]
        mc, te = d.model_channels, 4 * d.model_channels
        Bp = _rup(B, 64)
        R = self.layout.emb_rows
        deo = self._new(Bp, _rup(R, 64), zero=True)
        co8 = _rup(d.out_channels, 8)
        do = self._new(B * H * W, co8, zero=True)
        self.o.nchw_to_nhwc(dpred, do, B, d.out_channels, H * W)
        ch0 = plan.final_ch
        dhf = self._conv3_bwd(do, s["hf"], B, H, W, ch0, "out.2.weight", d.out_channels)
        dh = self._gn_bwd(dhf, s["h"], s["stf"], B, H * W, ch0, "out.0.")
        dskips: list[Tensor] = []
        for grp in plan.output_blocks[::-1]:
            dcat = self._group_bwd(grp, dh, B, save, eo, deo)
            ctot = grp[0].cin
            M = dcat.shape[0]
            # channel split of the concat: [h | skip]; the skip width is what the matching input block produced
            ich = self._skip_widths[len(dskips)]
            ch = ctot - ich
            dh, dsk = self._new(M, ch), self._new(M, ich)
            self.o.copy2d_bf16(dcat[:, :ch], dh, M, ch)
            self.o.copy2d_bf16(dcat[:, ch:], dsk, M, ich)
            dskips.append(dsk)
            self._fold_staged(min_pending=8)
        dh = self._group_bwd(plan.middle, dh, B, save, eo, deo)
        for grp in plan.input_blocks[::-1]:
            dh = self._add(dh, dskips.pop())
            dh = self._group_bwd(grp, dh, B, save, eo, deo)
            self._fold_staged(min_pending=8)
        assert not save and not dskips

        self._cond_bwd(deo, s, B)
        self._fold_staged()
        if self._use_side:
            ops.stream_wait(ops._s(), self._side_stream().cuda_stream)
        if self.reducer is not None:
            red = self.reducer
            ops.rec(lambda: (red.ready(0, self.layout.size), red.finish()))

    def _cond_bwd(self, deo: Tensor, s: dict, B: int) -> None:
        d = self.d
        mc, te = d.model_channels, 4 * d.model_channels
        Bp = _rup(B, 64)
        R = self.layout.emb_rows
        # FiLM projections (one stacked GEMM pair), then the conditioning MLP
        w0, b0 = self.layout.entries[self.layout.emb_w0][0], self.layout.entries[self.layout.emb_b0][0]
        ops.gemm_tn(deo, s["se"], self.grads[w0 : w0 + R * te].view(R, te), M=R)
        ops.colsum(deo, self.grads[b0 : b0 + R], B, R)
        dse = self._new(B, te, dtype=torch.float32)
        ops.gemm_nt(deo, self.sh["@emb|t"], dse, M=B, N=te, K=_rup(R, 64))
        table = d.n_classes is not None
        demb, demb16 = self._new(B, te, dtype=torch.float32), self._new(Bp, te, zero=True)
        ops.cond_combine_bwd(dse, s["emb"], s["y"] if table else None, demb, demb16[:B],
                             self.Gr("label_embed.embedding.weight") if table else None)
        ops.colsum(demb, self.Gr("time_embed.2.bias"), B, te)
        ops.gemm_tn(demb16, s["h1"], self.Gr("time_embed.2.weight"))
        dh1 = self._new(B, te, dtype=torch.float32)
        ops.gemm_nt(demb16, self.sh["time_embed.2.weight|t"], dh1, M=B, N=te, K=_rup(te, 64))
        dpre1 = self._new(Bp, te, zero=True)
        ops.silu_bwd(dh1, s["pre1"][:B], dpre1[:B])
        ops.gemm_tn(dpre1, s["temb"], self.Gr("time_embed.0.weight"), M=te, N=mc)
        ops.colsum(dpre1, self.Gr("time_embed.0.bias"), B, te)

    @property
    def _skip_widths(self) -> list[int]:
        """skip widths in the order the output blocks' BACKWARD meets them (last output block first)"""
        w = [grp[-1].cout for grp in self.plan.input_blocks]  # hs in push order; output block k pops hs[-1-k]
        return w  # backward visits output blocks in reverse: block n-1 popped hs[0], ...
