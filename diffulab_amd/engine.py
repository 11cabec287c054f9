"""Host-side runtime of the DiT hot path: parameter arena, bf16 weight shadows, activation workspace and
the forward / backward launch sequences over the C ABI.

MI355X-first choices:
  * all parameters live in ONE flat f32 HBM buffer (and one flat gradient buffer): the optimizer is a single
    fused launch and the data-parallel reduction is a handful of large RCCL collectives on contiguous memory;
  * the adaLN linears of every block (and of the last layer) are stored back to back so the modulation of the
    whole network is ONE GEMM per step (the conditioning vector is layer-invariant);
  * every activation is preallocated (288 GB of HBM3E: keep, don't recompute) so a step performs no
    allocation and the launch sequence is static (hipGraph-capturable);
  * residual stream and GEMM operands are bf16, accumulation / norm statistics / gradients of parameters f32.

Reference sites restated by the sequences below: mmdit.py:853-928 (simple_dit_forward / forward),
mmdit.py:288-309 (DiTBlock), mmdit.py:75-104 (DiTAttention), mmdit.py:542-549 (ModulatedLastLayer).
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
from torch import Tensor

from . import ops, tuning


_PARAM_EPOCH = 0


def bump_param_epoch() -> None:
    """to be called by anything that writes parameters through a raw pointer (fused AdamW / EMA kernels):
    torch version counters do not see those writes, and the bf16 weight shadows are keyed on this epoch."""
    global _PARAM_EPOCH
    _PARAM_EPOCH += 1


def _must(ok: bool) -> None:
    """an entry point that may answer DL_ERR_UNSUPPORTED was chosen for a shape it must serve (no assert: -O would drop the call)"""
    if not ok:
        raise RuntimeError("a HIP kernel declined a shape the engine had selected it for")


def _rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class DiTDims:
    input_channels: int = 4
    output_channels: int = 4
    inner_dim: int = 384
    embedding_dim: int = 384
    num_heads: int = 6
    mlp_ratio: int = 4
    patch_size: int = 2
    depth: int = 12
    rope_base: float = 10_000.0
    frequency_embedding: int = 256
    n_classes: int | None = 1000
    classifier_free: bool = True
    rope_axes_dim: list[int] = field(default_factory=list)

    def __post_init__(self) -> None:
        if not self.rope_axes_dim:
            hd = self.inner_dim // self.num_heads
            self.rope_axes_dim = [hd // 2, hd // 2]

    @property
    def head_dim(self) -> int:
        return self.inner_dim // self.num_heads

    def validate(self) -> None:
        D, E = self.inner_dim, self.embedding_dim
        if self.head_dim != 64:
            raise NotImplementedError(f"HIP attention kernels are built for head_dim 64 (got {self.head_dim})")
        if D % 64 or E % 64 or self.frequency_embedding % 64:
            raise NotImplementedError("inner_dim, embedding_dim and frequency_embedding must be multiples of 64")
        if D > 1024:
            raise NotImplementedError("row kernels support inner_dim <= 1024")
        if sum(self.rope_axes_dim) > self.head_dim or sum(self.rope_axes_dim) % 8:
            raise NotImplementedError("rotary width must be a multiple of 8 and <= head_dim")
        if len(self.rope_axes_dim) != 2:
            raise NotImplementedError("simple_dit uses a 2-axis (row, col) RoPE")


def rope_grid_tables(gh: int, gw: int, axes_dim: list[int], base: float) -> tuple[Tensor, Tensor]:
    """cos/sin tables [gh*gw, sum(axes)/2] of the 2-D grid, angles in fp64 then cast to fp32 exactly like
    nn.py:276-307 applied to the meshgrid(indexing='ij') ids of mmdit.py:871-886 (row axis first)."""
    pos = (torch.arange(gh, dtype=torch.float64).repeat_interleave(gw), torch.arange(gw, dtype=torch.float64).repeat(gh))
    cs, sn = [], []
    for p, d in zip(pos, axes_dim):
        inv = 1.0 / (torch.tensor(float(base), dtype=torch.float64) ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
        ang = p[:, None] * inv[None, :]
        cs.append(ang.cos().float())
        sn.append(ang.sin().float())
    return torch.cat(cs, 1).contiguous(), torch.cat(sn, 1).contiguous()


class ParamLayout:
    """name -> (offset, shape) inside the flat f32 arena.  Order: adaLN weights of all blocks + last layer
    (one [L*6D+2D, E] matrix), their biases, then everything else; every entry starts on a 256-byte boundary."""

    def __init__(self, d: DiTDims, prefixes: list[str] | None = None, extra: tuple[tuple[str, tuple[int, ...]], ...] = ()) -> None:
        """prefixes: state_dict prefix of every adaLN-zero block in execution order (default ``layers.{i}.``);
        extra: further non-block parameters (SPRINT's mask token and fuse linear), placed with the stem / head entries"""
        D, E, p = d.inner_dim, d.embedding_dim, d.patch_size
        self.entries: dict[str, tuple[int, tuple[int, ...]]] = {}
        self.size = 0
        self.prefixes = prefixes = list(prefixes) if prefixes is not None else [f"layers.{i}." for i in range(d.depth)]
        nb = len(prefixes)

        def add(name: str, shape: tuple[int, ...], align: int = 64) -> None:
            self.size = _rup(self.size, align)
            self.entries[name] = (self.size, shape)
            self.size += math.prod(shape)

        # -- contiguous adaLN matrix / bias (rows are multiples of 64*... so no padding is inserted between them)
        for i, pre in enumerate(prefixes):
            add(pre + "modulation.lin.weight", (6 * D, E), align=1 if i else 64)
        add("last_layer.adaLN_modulation.1.weight", (2 * D, E), align=1)
        for i, pre in enumerate(prefixes):
            add(pre + "modulation.lin.bias", (6 * D,), align=1 if i else 64)
        add("last_layer.adaLN_modulation.1.bias", (2 * D,), align=1)
        self.mod_rows = nb * 6 * D + 2 * D
        self.mod_w0, self.mod_b0 = prefixes[0] + "modulation.lin.weight", prefixes[0] + "modulation.lin.bias"
        self.block_first = [pre + "norm_1.weight" for pre in prefixes]
        if d.n_classes is not None:
            add("label_embed.embedding.weight", (d.n_classes + (1 if d.classifier_free else 0), E))
        add("time_embed.0.weight", (E, d.frequency_embedding))
        add("time_embed.0.bias", (E,))
        add("time_embed.2.weight", (E, E))
        add("time_embed.2.bias", (E,))
        add("conv_proj.weight", (D, d.input_channels, p, p))
        add("last_layer.linear.weight", (p * p * d.output_channels, D))
        add("last_layer.linear.bias", (p * p * d.output_channels,))
        for name, shape in extra:
            add(name, shape)
        for pre in prefixes:
            add(pre + "norm_1.weight", (D,))
            add(pre + "norm_1.bias", (D,), align=1)  # [w; b] adjacent: one reduce writes both gradients
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


class DiTEngine:
    _conv_name = "conv_proj.weight"  # patch-embedding convolution of the (encoder) token stream

    def __init__(self, dims: DiTDims, device: torch.device | str = "cuda") -> None:
        dims.validate()
        self.d = dims
        self.dev = torch.device(device)
        self._ki = _rup(dims.input_channels * dims.patch_size**2, 64)   # K of the patch-embedding GEMM (zero-padded)
        self._ko = _rup(dims.output_channels * dims.patch_size**2, 64)  # K of the last linear's data gradient
        self.layout = self._make_layout(dims)
        self.prefixes = self.layout.prefixes
        self.params: Tensor | None = None
        self.grads: Tensor | None = None
        self._shadow_key: tuple | None = None
        self.manual_version = 0
        self.param_version = 0  # sum of the parameters' own version counters, set by the owning module before every forward
        self._ws_key: tuple | None = None
        self._ws_cache: dict[tuple, tuple] = {}
        self._cast_table = None
        self._cast_table_ptr = 0
        self._rope: dict[tuple[int, int], tuple[Tensor, Tensor]] = {}
        self.reducer = None  # optional training.dp.GradReducer: gets ready(lo, hi) as gradient ranges complete
        self._build_shadows()
        ent = self.layout.entries
        starts = [ent[n][0] for n in self.layout.block_first] + [self.layout.size]
        self.layer_ranges = [(starts[i], starts[i + 1]) for i in range(len(self.prefixes))]

    def _make_layout(self, dims: DiTDims) -> ParamLayout:
        return ParamLayout(dims)

    # ------------------------------------------------------------------ parameters
    def bind(self, params: Tensor, grads: Tensor | None) -> None:
        assert params.dtype == torch.float32 and params.numel() == self.layout.size and params.is_cuda
        self.params, self.grads = params, grads
        self._shadow_key = None
        self._blk_cache: dict[tuple, object] = {}
        self._pviews: dict[str, Tensor] = {}
        self._gviews: dict[str, Tensor] = {}

    def P(self, name: str) -> Tensor:
        v = self._pviews.get(name)  # (views of the arena are cached: creating one costs ~3 us and a block uses a dozen)
        if v is None:
            v = self._pviews[name] = self.layout.view(self.params, name)
        return v

    def G(self, name: str) -> Tensor:
        v = self._gviews.get(name)
        if v is None:
            v = self._gviews[name] = self.layout.view(self.grads, name)
        return v

    def _build_shadows(self) -> None:
        """bf16 copies consumed by the MFMA GEMMs: W [out, rup64(in)] for forward, W^T [in, rup64(out)] for dgrad."""
        d, dev = self.d, self.dev
        D, E = d.inner_dim, d.embedding_dim
        self.sh: dict[str, Tensor] = {}
        self._casts: list[tuple[str, tuple[int, int], str | None, str | None]] = []

        def reg(name: str, R: int, C: int, fwd: bool = True, dgrad: bool = True) -> None:
            f = t = None
            if fwd:
                f = name + "|f"
                self.sh[f] = torch.zeros(R, _rup(C, 64), device=dev, dtype=torch.bfloat16)
            if dgrad:
                t = name + "|t"
                self.sh[t] = torch.zeros(C, _rup(R, 64), device=dev, dtype=torch.bfloat16)
            self._casts.append((name, (R, C), f, t))

        self.mod_name = self.layout.mod_w0  # start of the stacked [mod_rows, E] matrix
        reg("@mod", self.layout.mod_rows, E)
        reg("time_embed.0.weight", E, d.frequency_embedding, dgrad=False)
        reg("time_embed.2.weight", E, E)
        reg(self._conv_name, D, d.input_channels * d.patch_size**2, dgrad=False)
        reg("last_layer.linear.weight", d.patch_size**2 * d.output_channels, D)
        self._extra_shadows(reg)
        for pre in self.prefixes:
            self._block_shadows(reg, pre)

    def _block_shadows(self, reg, pre: str) -> None:
        d, dev = self.d, self.dev
        D = d.inner_dim
        reg(pre + "attention.qkv.weight", 3 * D, D)
        reg(pre + "attention.proj_out.weight", D, D)
        reg(pre + "mlp_input.0.weight", 2 * d.mlp_ratio * D, D)
        self.sh[pre + "mlp_input.0.weight|g"] = torch.zeros(2 * d.mlp_ratio * D, D, device=dev, dtype=torch.bfloat16)
        reg(pre + "mlp_input.2.weight", D, d.mlp_ratio * D)

    def _extra_shadows(self, reg) -> None:  # hook: shadows of further linears (subclasses)
        pass

    def _src(self, name: str, shape: tuple[int, int]) -> Tensor:
        if name == "@mod":
            off = self.layout.entries[self.mod_name][0]
            return self.params[off : off + shape[0] * shape[1]].view(shape)
        return self.P(name).view(shape)

    def refresh_shadows(self, force: bool = False) -> None:
        ver = 0 if self.params.is_inference() else self.params._version
        # writes THROUGH a parameter (load_state_dict on a flattened model, a stock optimizer, p.copy_) bump that parameter's
        # version counter, not the arena's: the owning module passes the sum of those counters as param_version
        key = (self.params.data_ptr(), ver, self.manual_version, _PARAM_EPOCH, self.param_version)
        if not force and key == self._shadow_key:
            return
        if self._cast_table is None or self._cast_table_ptr != self.params.data_ptr():
            ent = []  # one device table for all shadows (incl. the row-permuted MLP-up copy of the fused SwiGLU kernel)
            for name, shape, f, t in self._casts:
                g = self.sh.get(name + "|g")
                ent.append((self._src(name, shape), self.sh[f] if f else None, self.sh[t] if t else None, g))
            self._cast_table, self._cast_table_ptr = ops.CastTable(ent), self.params.data_ptr()
        self._cast_table.run()
        self._shadow_key = key

    def _side_stream(self) -> "torch.cuda.Stream":
        if getattr(self, "_side", None) is None:
            mask = tuning.text("DL_SIDE_CU_MASK")  # e.g. "i4" = every 4th CU, "b128" = the first 128 CUs
            if mask:
                self._side = ops.masked_stream(mask, self.dev)
            elif tuning.on("DL_SIDE_LOW_PRIORITY"):
                self._side = ops.low_priority_stream(self.dev)
            else:
                self._side = torch.cuda.Stream(device=self.dev)
        return self._side

    def params_changed(self) -> None:
        """call after writing the parameter arena through a raw pointer (fused AdamW)."""
        self.manual_version += 1

    # ------------------------------------------------------------------ workspace
    def _alloc(self, B: int, H: int, W: int, train: bool) -> None:
        key = (B, H, W, train)
        if key == self._ws_key:
            return
        if key in self._ws_cache:  # workspaces are kept per shape: captured hipGraphs have their addresses baked in
            self.ws, self.geo = self._ws_cache[key]
            self._ws_key = key
            return
        d, dev = self.d, self.dev
        D, E, p, L = d.inner_dim, d.embedding_dim, d.patch_size, d.depth
        gh, gw = H // p, W // p
        N = gh * gw
        M = B * N
        # token grids the attention kernels do not take as they are (not a multiple of 64 up to 256 / of 256 beyond: non-square and
        # multi-aspect-ratio latents): q / k / v rows padded to a multiple of 256 per (sample, head), pad keys masked (dl_attn_*_ex)
        pad = ops.attn_needs_padding(N)
        Nd = _rup(N, 256) if pad else N
        if pad and type(self) is DiTEngine and Nd <= 2048 and M % 64:
            raise NotImplementedError(f"token grid {gh}x{gw} ({N} tokens) in the bf16 regime: batch * tokens must be a multiple of 64 (got "
                                      f"{B} * {N}); the fp32 regime (precision_type=\"no\" / set_precision(\"fp32\")) takes any grid and batch")
        if Nd > 2048 or (pad and type(self) is not DiTEngine):
            raise NotImplementedError(f"token grid {gh}x{gw}: the bf16 regime's attention takes up to 2048 tokens"
                                      + ("" if type(self) is DiTEngine else ", N % 64 == 0 up to 256 or N % 256 == 0 beyond, in this engine")
                                      + f" (got {N}); the fp32 regime (precision_type=\"no\" / set_precision(\"fp32\")) takes any grid")
        Bp = _rup(B, 64)
        Fo = p * p * d.output_channels
        bf, f32 = torch.bfloat16, torch.float32

        def z(*shape, dtype=bf):
            with torch.inference_mode(False):  # workspaces outlive an inference_mode() sampler loop
                return torch.zeros(*shape, device=dev, dtype=dtype)

        w: dict[str, object] = {}
        w["tokP"] = z(M, self._ki)                      # patchified input, K padded to 64
        w["temb"] = z(Bp, d.frequency_embedding)
        w["pre1"] = z(Bp, E)
        w["h1"] = z(Bp, E)
        w["e"] = z(Bp, E, dtype=f32)
        w["emb"] = z(Bp, E, dtype=f32)
        w["se"] = z(Bp, E)
        w["mod"] = z(Bp, self.layout.mod_rows)
        nl = L if train else 1                    # inference reuses one block's buffers for every layer
        # MLP backward with recomputed pre-activations (csrc/mlp_bwd.hip): the training forward then stores h only and no
        # [M, 2F] pre-activation buffer exists (403 MB per block at B = 256); shapes without that kernel keep u
        F_ = d.mlp_ratio * D
        rc_u = train and type(self) is DiTEngine and ops.mlp_recompute_ok(M, D, F_)
        w["mlp_recompute"] = rc_u
        w["x"] = [z(M, D) for _ in range((L + 1) if train else 2)]
        per = []
        for _ in range(nl):
            per.append({
                "mean1": z(M, dtype=f32), "rstd1": z(M, dtype=f32), "xm1": z(M, D), "qkv": z(M, 3 * D),
                "q": z(B, d.num_heads, Nd, 64), "k": z(B, d.num_heads, Nd, 64),
                "v": None if ops.v_in_place(N) else z(B, d.num_heads, Nd, 64),  # (N <= 256: V is read in place from qkv)
                "ao": z(B * Nd, D) if pad else None,
                "rrms": z(M, 2, dtype=f32), "a": z(M, D), "lse": z(B, d.num_heads, Nd, dtype=f32), "t1": z(M, D),
                "x1": z(M, D), "mean2": z(M, dtype=f32), "rstd2": z(M, dtype=f32), "xm2": z(M, D),
                "u": None if rc_u else z(M, 2 * d.mlp_ratio * D), "h": z(M, d.mlp_ratio * D), "t2": z(M, D),
            })
        w["layers"] = per
        # QK-norm on load (dl_gemm_nt_ssq + dl_attn_fwd_qkn): per-block sums of squares of the q / k rows, zeroed once per forward (one
        # buffer per block also in inference: the GEMM epilogues ADD into it)
        if pad:
            w["kb"] = z(B, Nd, dtype=f32)
            w["kb"][:, N:] = float("-inf")
        if self._qkn_on_load(M, N):
            w["ssq_all"] = z(L, M, 2, dtype=f32)
        w["meanf"], w["rstdf"] = z(M, dtype=f32), z(M, dtype=f32)
        w["xf"] = z(M, D)
        w["otok"] = z(M, _rup(Fo, 8), dtype=f32)
        w["pred"] = z(B, d.output_channels, H, W, dtype=f32)
        if train:
            w["dO"] = z(M, self._ko)
            w["dxa"], w["dxb"] = z(M, D), z(M, D)
            w["dxm"], w["da"] = z(M, D), z(M, D)
            w["dh"] = z(M, d.mlp_ratio * D)
            # per-block inputs of the weight-gradient GEMMs (consumed asynchronously on the side stream)
            w["wg"] = [{"dt2": z(M, D), "du": z(M, 2 * d.mlp_ratio * D), "dt1": z(M, D), "dqkv": z(M, 3 * D)}
                       for _ in range(L)]
            # N <= 256 and D <= 512: dQ / dK / dV leave the attention backward token-major inside the dqkv rows and the QK-norm
            # backward works in place (its per-workgroup scale-gradient partials land in qk_part): no dq / dk buffers
            # (from 32768 token rows: at the CIFAR config's 8192 rows the in-place pair is 5 % of a 6.7 ms step SLOWER)
            if ops.v_in_place(N) and D <= 512 and M >= 32768 and type(self) is DiTEngine and tuning.on("DL_QK_INPLACE"):
                w["qk_part"] = torch.empty(1024 * 2 * D, device=dev, dtype=f32)
                w["dq"] = w["dk"] = None
            else:
                w["dq"], w["dk"] = z(B, d.num_heads, Nd, 64), z(B, d.num_heads, Nd, 64)
            w["dv"] = None if ops.v_in_place(N) else z(B, d.num_heads, Nd, 64)
            if pad:
                w["dao"] = z(B * Nd, D)
            w["dmod"] = z(Bp, self.layout.mod_rows)                  # bf16 operand of the modulation GEMMs' backward
            w["dmod32"] = z(Bp, self.layout.mod_rows, dtype=f32)     # f32 accumulator the block kernels add into
            w["dwb"] = z(2 * L, B, 2, D, dtype=f32)  # per-norm partials: folded on the side stream, cleared while read
            w["dse"] = z(Bp, E, dtype=f32)
            w["demb"] = z(Bp, E, dtype=f32)
            w["demb16"] = z(Bp, E)
            w["dh1"] = z(Bp, E, dtype=f32)
            w["dpre1"] = z(Bp, E)
            # partial slabs of the grouped weight-gradient launch (dl_gemm_tn_group: the four linears of a block in one atomics-free
            # launch): 8 token ranges x (4 + 3 mlp_ratio) D^2 floats; one set, the side stream runs the blocks one after the other
            if self._grouped_wgrad(M):
                w["tn_slab"] = torch.empty(8 * (4 + 3 * d.mlp_ratio) * D * D, device=dev, dtype=f32)
            # scratch of the bit-reproducible small GEMMs / column sums of the head and the conditioning path (partial images per
            # split, folded in a fixed order: ops.gemm_tn / gemm_nt / colsum with scratch=): two images of the stacked adaLN
            # weight gradient is the largest user
            w["det_scr"] = torch.empty(max(2 * self.layout.mod_rows * E, 1 << 22), device=dev, dtype=f32)
            w["scr_last"] = z(_rup(Fo, 8), D, dtype=f32)
            w["scr_conv"] = z(D, self._ki, dtype=f32)
        self.ws, self._ws_key = w, key
        self.geo = (B, H, W, gh, gw, N, M, Bp, Fo)
        if len(self._ws_cache) >= 8:  # bound the cache: drop the oldest shape (its graphs are dropped by the module too)
            self._ws_cache.pop(next(iter(self._ws_cache)))
        self._ws_cache[key] = (w, self.geo)
        if (gh, gw) not in self._rope:
            c, s = rope_grid_tables(gh, gw, d.rope_axes_dim, d.rope_base)
            self._rope[(gh, gw)] = (c.to(dev), s.to(dev))

    def _stem_fwd(self, x: Tensor, t: Tensor, y_eff: Tensor | None, x0: Tensor | None = None, patch_gemm: bool = True) -> Tensor:
        """patch embedding into ``x0`` (default ws["x"][0]) and the conditioning path: time MLP (+ label row) -> SiLU -> the
        stacked adaLN GEMM of every block and of the last layer.  Returns the modulation matrix bf16 [Bp, mod_rows].
        patch_gemm=False leaves the patch-embedding GEMM to the caller (row-complete path: it needs the modulation rows)."""
        d, w, sh = self.d, self.ws, self.sh
        B, _, _, _, _, _, M, _, _ = self.geo
        D, E = d.inner_dim, d.embedding_dim
        ops.patchify(x, w["tokP"], d.patch_size, ops.PATCH_CPP)
        if patch_gemm:
            ops.gemm_nt(w["tokP"], sh[self._conv_name + "|f"], w["x"][0] if x0 is None else x0, M=M, N=D, K=self._ki)
        ops.timestep_embedding(t, w["temb"][:B])
        ops.gemm_nt(w["temb"], sh["time_embed.0.weight|f"], w["h1"], bias=self.P("time_embed.0.bias"), act=ops.ACT_SILU,
                    pre_out=w["pre1"], M=B, N=E, K=d.frequency_embedding)
        ops.gemm_nt(w["h1"], sh["time_embed.2.weight|f"], w["e"], bias=self.P("time_embed.2.bias"), M=B, N=E, K=E)
        table = self.P("label_embed.embedding.weight") if d.n_classes is not None else None
        ops.cond_combine_fwd(w["e"][:B], table, y_eff if table is not None else None, w["emb"][:B], w["se"][:B])
        mod_bias = self.params[self.layout.entries[self.layout.mod_b0][0] :][: self.layout.mod_rows]
        ops.gemm_nt(w["se"], sh["@mod|f"], w["mod"], bias=mod_bias, M=B, N=self.layout.mod_rows, K=E)
        return w["mod"]

    def _main_wgs(self) -> int:
        """workgroup budget of the persistent main-chain kernels (0 = one per CU).  DL_DP_RESERVE_CUS = r > 0: with a gradient reducer
        attached the grids leave r compute units to RCCL's channel workgroups.  Measured with the stand-in of scripts/lab/occupied_cus.py
        (round 4): the headline shape's tile counts are whole multiples of 256, so 248 workgroups run an extra, nearly empty round --
        20.7 -> 24.1 ms per step WITHOUT any foreign workgroup, 26.2 vs 24.6 ms with 16 of them resident for half the step -- a loss;
        the shipped default is therefore 0 and the reducer's measured choice between overlapped buckets and one exchange after the
        backward (training/dp.py) stays the mechanism."""
        explicit = tuning.integer("DL_MAIN_WGS", 0)
        if explicit:
            return explicit
        reserve = tuning.integer("DL_DP_RESERVE_CUS", 0)
        if reserve and self.reducer is not None and getattr(self.reducer, "enabled", True):
            if getattr(self, "_cus", None) is None:
                self._cus = torch.cuda.get_device_properties(self.dev).multi_processor_count
            return max(8, (self._cus - reserve) & ~7)
        return 0

    # ------------------------------------------------------------------ native block driver (csrc/block.hip)
    def _native_blocks(self) -> bool:
        """one C call per block and direction (dl_dit_block_fwd / _bwd) instead of ~25 launches from Python; DL_NATIVE_BLOCK=0 is the
        A/B switch back to the Python-issued sequence (identical kernels, identical order)"""
        return type(self) is DiTEngine and tuning.on("DL_NATIVE_BLOCK") and not (self.geo is not None and ops.attn_needs_padding(self.geo[5]))

    def _block_args(self, i: int, train: bool):
        """the dl_dit_block_t of block i on the current workspace (cached: every pointer is fixed once arena and workspace exist)"""
        key = (self._ws_key, i, self.reducer is None, self._main_wgs())
        blk = self._blk_cache.get(key)
        if blk is not None:
            return blk
        from ._block import DitBlock

        d, w, sh = self.d, self.ws, self.sh
        B, _, _, gh, gw, N, M, Bp, Fo = self.geo
        D, L, F = d.inner_dim, d.depth, d.mlp_ratio * d.inner_dim
        pre, mo = f"layers.{i}.", i * 6 * D
        mod, xs = w["mod"], w["x"]
        a = w["layers"][i if train else 0]
        xin = xs[i] if train else xs[i & 1]
        cos, sin = self._rope[(gh, gw)]
        mrow = lambda off: mod.data_ptr() + 2 * off  # noqa: E731  (row view mod[:, off:]: bf16)
        blk = DitBlock()
        blk.B, blk.N, blk.D, blk.H, blk.F = B, N, D, d.num_heads, F
        blk.ld_mod = mod.stride(0)
        blk.ldw_d, blk.ldw_f = sh[pre + "attention.qkv.weight|f"].stride(0), sh[pre + "mlp_input.2.weight|f"].stride(0)
        blk.ldwt_d, blk.ldwt_f2 = sh[pre + "attention.proj_out.weight|t"].stride(0), sh[pre + "mlp_input.0.weight|t"].stride(0)
        blk.ldwt_3d = sh[pre + "attention.qkv.weight|t"].stride(0)
        blk.rot, blk.eps = sum(d.rope_axes_dim), 1e-5
        blk.max_workgroups = self._main_wgs()
        prev = None
        if i > 0:
            ap = w["layers"][(i - 1) if train else 0]
            prev = (ap["x1"], ap["t2"], mrow((i - 1) * 6 * D + 5 * D))
        blk.set(x_in=xin, pend_x=prev[0] if prev else None, pend_t=prev[1] if prev else None, pend_gate=prev[2] if prev else None,
                scale1=mrow(mo), shift1=mrow(mo + D), gate1=mrow(mo + 2 * D), scale2=mrow(mo + 3 * D), shift2=mrow(mo + 4 * D),
                gate2=mrow(mo + 5 * D),
                ln1_w=self.P(pre + "norm_1.weight"), ln1_b=self.P(pre + "norm_1.bias"), ln2_w=self.P(pre + "norm_2.weight"),
                ln2_b=self.P(pre + "norm_2.bias"), qn_scale=self.P(pre + "attention.qk_norm.query_norm.scale"),
                kn_scale=self.P(pre + "attention.qk_norm.key_norm.scale"),
                w_qkv=sh[pre + "attention.qkv.weight|f"], w_proj=sh[pre + "attention.proj_out.weight|f"],
                w_up=sh[pre + "mlp_input.0.weight|f"], w_up_perm=sh[pre + "mlp_input.0.weight|g"], w_down=sh[pre + "mlp_input.2.weight|f"],
                wt_qkv=sh[pre + "attention.qkv.weight|t"], wt_proj=sh[pre + "attention.proj_out.weight|t"],
                wt_up=sh[pre + "mlp_input.0.weight|t"], wt_down=sh[pre + "mlp_input.2.weight|t"], rope_cos=cos, rope_sin=sin,
                xm1=a["xm1"], mean1=a["mean1"], rstd1=a["rstd1"], qkv=a["qkv"], q=a["q"], k=a["k"], v=a["v"], rrms=a["rrms"], a=a["a"],
                lse=a["lse"], t1=a["t1"], x1=a["x1"], xm2=a["xm2"], mean2=a["mean2"], rstd2=a["rstd2"], u=a["u"], h=a["h"], t2=a["t2"],
                ssq=w["ssq_all"][i] if "ssq_all" in w else None)
        rows = self._row_gemms(M, N)
        blk.row_gemms = 0
        if rows:
            if i + 1 < L:
                nx, mn, nxt = w["layers"][(i + 1) if train else 0], (i + 1) * 6 * D, f"layers.{i + 1}."
                lnw, lnb, blk.next_eps = self.P(nxt + "norm_1.weight"), self.P(nxt + "norm_1.bias"), 1e-5
                xm_n, mu_n, rs_n = nx["xm1"], nx["mean1"], nx["rstd1"]
            else:
                mn, lnw, lnb, blk.next_eps = L * 6 * D, None, None, 1e-6
                xm_n, mu_n, rs_n = w["xf"], w["meanf"], w["rstdf"]
            blk.set(next_ln_w=lnw, next_ln_b=lnb, next_scale=mrow(mn), next_shift=mrow(mn + D),
                    next_x=xs[i + 1] if train else xs[(i + 1) & 1], next_xm=xm_n, next_mean=mu_n, next_rstd=rs_n)
            # bit 1: QK-norm + RoPE in the qkv epilogue (measured slightly slower inside the step: opt-in); bit 2: the LayerNorm-affine
            # partials of all blocks are folded by ONE launch at the end of the backward (not with a gradient reducer attached: it
            # wants every block's range final as soon as the block is done)
            blk.row_gemms = 1 | (2 if tuning.on("DL_ROW_GEMM_QK") else 0) | (4 if self.reducer is None else 0)
        if train and self.grads is not None:
            g, dmod = w["wg"][i], w["dmod32"]
            blk.ld_dmod = dmod.stride(0)
            drow = lambda off: dmod.data_ptr() + 4 * off  # noqa: E731
            pg = None
            if i > 0:
                mp = (i - 1) * 6 * D
                pg = (w["layers"][i - 1]["t2"], mrow(mp + 5 * D), w["wg"][i - 1]["dt2"], drow(mp + 5 * D))
            blk.set(dt2=g["dt2"], dx_in=w["dxa"], dx_mid=w["dxb"], dx_out=w["dxa"], dh=w["dh"], du=g["du"], dxm=w["dxm"], dt1=g["dt1"],
                    da=w["da"], dq=w["dq"], dk=w["dk"], dv=w["dv"], dqkv=g["dqkv"], dfeat=None,
                    dscale1=drow(mo), dshift1=drow(mo + D), dgate1=drow(mo + 2 * D), dscale2=drow(mo + 3 * D), dshift2=drow(mo + 4 * D),
                    dwb1=w["dwb"][2 * i], dwb2=w["dwb"][2 * i + 1],
                    prev_t2=pg[0] if pg else None, prev_gate2=pg[1] if pg else None, prev_dt2=pg[2] if pg else None,
                    prev_dgate2=pg[3] if pg else None,
                    g_qkv=self.G(pre + "attention.qkv.weight"), g_proj=self.G(pre + "attention.proj_out.weight"),
                    g_up=self.G(pre + "mlp_input.0.weight"), g_down=self.G(pre + "mlp_input.2.weight"),
                    g_ln1=self.G(pre + "norm_1.weight"), g_ln2=self.G(pre + "norm_2.weight"),
                    g_qk_scale=self.G(pre + "attention.qk_norm.query_norm.scale"), tn_slab=w.get("tn_slab"),
                    qk_partials=w.get("qk_part"))
            blk.tn_slab_floats = w["tn_slab"].numel() if "tn_slab" in w else 0
        self._blk_cache[key] = blk
        return blk

    # ------------------------------------------------------------------ forward
    def forward(self, x: Tensor, t: Tensor, y_eff: Tensor | None, train: bool = True, refresh: bool = True) -> Tensor:
        """x f32 [B,C,H,W]; t f32 [B] (flow: in [0,1]; ddpm: indices as floats, both fed unscaled like the
        reference); y_eff int64 [B] labels AFTER the classifier-free drop, or None.  Returns pred f32 [B,Co,H,W]
        (a workspace buffer: consume it before the next forward)."""
        d = self.d
        B, C, H, W = x.shape
        assert C == d.input_channels and x.dtype == torch.float32 and x.is_cuda
        self._alloc(B, H, W, train)
        if refresh:  # (False while a hipGraph of this forward is being captured: the replay path refreshes eagerly)
            self.refresh_shadows(force=train)  # a training forward always follows a parameter update
        w, sh = self.ws, self.sh
        _, _, _, gh, gw, N, M, Bp, Fo = self.geo
        D, E, L, Hh = d.inner_dim, d.embedding_dim, d.depth, d.num_heads
        cos, sin = self._rope[(gh, gw)]
        rot = sum(d.rope_axes_dim)
        self._train = train
        self._yeff = y_eff

        fused = self._row_gemms(M, N)
        if "ssq_all" in w:
            w["ssq_all"].zero_()
        mod = self._stem_fwd(x, t, y_eff, patch_gemm=not fused)
        xs = w["x"]
        if fused:
            if self._native_blocks():
                a0 = w["layers"][0]
                _must(ops.ln_modulate_gemm_fwd(w["tokP"], sh[self._conv_name + "|f"], None, None, self.P("layers.0.norm_1.weight"),
                                               self.P("layers.0.norm_1.bias"), mod[:, 0:D], mod[:, D : 2 * D], N, 1e-5, None, xs[0],
                                               a0["xm1"], a0["mean1"], a0["rstd1"], K=self._ki))
                for i in range(L):
                    ops.dit_block_fwd(self._block_args(i, train), train)
            else:
                self._forward_row_gemms(mod, train)
            ops.gemm_nt(w["xf"], sh["last_layer.linear.weight|f"], w["otok"], bias=self.P("last_layer.linear.bias"), M=M,
                        N=Fo, K=D)
            ops.unpatchify(w["otok"], w["pred"], d.patch_size)
            return w["pred"]

        # The gated residual of every sub-layer (x += gate * f(...)) is applied by the NEXT LayerNorm-modulate kernel, which
        # has to read the residual stream anyway: the projection / MLP-down GEMMs stay plain stores (fast 256x384 tiles)
        # and write t1 / t2, which the backward needs for the gate gradients.
        pend = None  # (x_base, t, gate) of the sub-layer whose residual add is still pending
        native = self._native_blocks()
        for i in range(L):
            if native:
                ops.dit_block_fwd(self._block_args(i, train), train)
                a = w["layers"][i if train else 0]
                pend = (a["x1"], a["t2"], mod[:, i * 6 * D + 5 * D : i * 6 * D + 6 * D])
                continue
            a = w["layers"][i if train else 0]
            xin = xs[i] if train else xs[i & 1]
            pre = f"layers.{i}."
            mo = i * 6 * D
            if pend is None:
                ops.ln_modulate_fwd(xin, self.P(pre + "norm_1.weight"), self.P(pre + "norm_1.bias"), mod[:, mo : mo + D],
                                    mod[:, mo + D : mo + 2 * D], N, 1e-5, a["xm1"], a["mean1"], a["rstd1"])
            else:
                ops.ln_modulate_fwd(pend[0], self.P(pre + "norm_1.weight"), self.P(pre + "norm_1.bias"), mod[:, mo : mo + D],
                                    mod[:, mo + D : mo + 2 * D], N, 1e-5, a["xm1"], a["mean1"], a["rstd1"], t=pend[1],
                                    gate=pend[2], x_out=xin)
            ops.gemm_nt(a["xm1"], sh[pre + "attention.qkv.weight|f"], a["qkv"])
            # up to 256 tokens the attention kernels address V (and dV) inside the token-major qkv (dqkv) rows: no V head split
            v_in_place = ops.v_in_place(N)
            ops.qk_norm_rope_fwd(a["qkv"], self.P(pre + "attention.qk_norm.query_norm.scale"),
                                 self.P(pre + "attention.qk_norm.key_norm.scale"), cos, sin, a["q"], a["k"],
                                 None if v_in_place else a["v"], a["rrms"], B, N, Hh, 64, rot)
            if a.get("ao") is not None:  # padded rows, masked pad keys (_alloc)
                Np = a["q"].shape[2]
                ops.attn_fwd_ex(a["q"], a["k"], a["v"], a["ao"], a["lse"], B, Hh, Np, Np, 64, 64**-0.5, w["kb"])
                ops.copy_rows3d(a["ao"], Np * D, D, a["a"], N * D, D, B, N, D)
            elif v_in_place:
                ops.attn_fwd_qkv(a["q"], a["k"], a["qkv"], a["a"], a["lse"], B, Hh, N, 64, 64**-0.5)
            else:
                ops.attn_fwd(a["q"], a["k"], a["v"], a["a"], a["lse"], B, Hh, N, 64, 64**-0.5)
            ops.gemm_nt(a["a"], sh[pre + "attention.proj_out.weight|f"], a["t1"])
            ops.ln_modulate_fwd(xin, self.P(pre + "norm_2.weight"), self.P(pre + "norm_2.bias"),
                                mod[:, mo + 3 * D : mo + 4 * D], mod[:, mo + 4 * D : mo + 5 * D], N, 1e-5, a["xm2"],
                                a["mean2"], a["rstd2"], t=a["t1"], gate=mod[:, mo + 2 * D : mo + 3 * D], x_out=a["x1"])
            if not ops.gemm_nt_swiglu(a["xm2"], sh[pre + "mlp_input.0.weight|g"], a["u"] if train else None, a["h"]):
                assert a["u"] is not None, "the recompute mode is only chosen for shapes the fused MLP-up kernel serves"
                ops.gemm_nt(a["xm2"], sh[pre + "mlp_input.0.weight|f"], a["u"])  # small / ragged shapes: unfused pair
                ops.swiglu_fwd(a["u"], a["h"])
            ops.gemm_nt(a["h"], sh[pre + "mlp_input.2.weight|f"], a["t2"])
            pend = (a["x1"], a["t2"], mod[:, mo + 5 * D : mo + 6 * D])

        xl = xs[L] if train else xs[L & 1]
        mo = L * 6 * D
        ops.ln_modulate_fwd(pend[0], None, None, mod[:, mo : mo + D], mod[:, mo + D : mo + 2 * D], N, 1e-6, w["xf"], w["meanf"],
                            w["rstdf"], t=pend[1], gate=pend[2], x_out=xl)
        ops.gemm_nt(w["xf"], sh["last_layer.linear.weight|f"], w["otok"], bias=self.P("last_layer.linear.bias"), M=M,
                    N=Fo, K=D)
        ops.unpatchify(w["otok"], w["pred"], d.patch_size)
        return w["pred"]

    def _grouped_wgrad(self, M: int) -> bool:
        """the four weight gradients of a block as ONE launch without atomics (csrc/gemm_w4.hip, dl_gemm_tn_group): every linear of
        the block must be a whole number of 384 x 192 tiles.  DL_WGRAD_GROUP=0 is the A/B switch back to four atomic launches."""
        D, F = self.d.inner_dim, self.d.mlp_ratio * self.d.inner_dim
        return (type(self) is DiTEngine and ops.WgradGroups.shapes_ok(D, F, M) and tuning.on("DL_WGRAD_GROUP"))

    def _qkn_on_load(self, M: int, N: int) -> bool:
        # (dl_gemm_nt_ssq runs the persistent 256 x 384 tiles: it wants >= 64 of them, i.e. 22 samples of 256 tokens at D = 384)
        return (self._row_gemms(M, N) and self.d.inner_dim % 384 == 0 and (M // 256) * (3 * self.d.inner_dim // 384) >= 64
                and tuning.on("DL_QKN_ON_LOAD") and not tuning.on("DL_ROW_GEMM_QK"))

    def _row_gemms(self, M: int, N: int) -> bool:
        """the row-complete GEMM path (csrc/gemm_ln.hip): LayerNorm-modulate forward / backward and QK-norm + RoPE run as epilogues
        of the GEMMs that feed them.  D == 384 with 256 tokens per sample (one 256 x 384 tile = one sample's whole rows)"""
        return type(self) is DiTEngine and ops.v_in_place(N) and ops.row_gemm_ok(M, self.d.inner_dim, N)

    def _forward_row_gemms(self, mod: Tensor, train: bool) -> None:
        """block chain of forward() with every LayerNorm-modulate in the epilogue of the GEMM in front of it: the patch embedding
        carries LN1 of block 0, the projection carries LN2 of its block, the MLP-down GEMM carries LN1 of the NEXT block (or the
        final LayerNorm); the qkv GEMM carries QK-norm + RoPE + the head split"""
        d, w, sh = self.d, self.ws, self.sh
        B, _, _, gh, gw, N, M, _, _ = self.geo
        D, L, Hh = d.inner_dim, d.depth, d.num_heads
        cos, sin = self._rope[(gh, gw)]
        rot = sum(d.rope_axes_dim)
        xs = w["x"]
        lay = lambda i: w["layers"][i if train else 0]  # noqa: E731
        xbuf = lambda i: xs[i] if train else xs[i & 1]  # noqa: E731
        fused_qk = tuning.on("DL_ROW_GEMM_QK")
        a0 = lay(0)
        _must(ops.ln_modulate_gemm_fwd(w["tokP"], sh[self._conv_name + "|f"], None, None, self.P("layers.0.norm_1.weight"),
                                        self.P("layers.0.norm_1.bias"), mod[:, 0:D], mod[:, D : 2 * D], N, 1e-5, None, xbuf(0),
                                        a0["xm1"], a0["mean1"], a0["rstd1"], K=self._ki))
        for i in range(L):
            a, xin, pre, mo = lay(i), xbuf(i), f"layers.{i}.", i * 6 * D
            if "ssq_all" in w:
                ssq = w["ssq_all"][i]
                _must(ops.gemm_nt_ssq(a["xm1"], sh[pre + "attention.qkv.weight|f"], a["qkv"], ssq))
                ops.attn_fwd_qkn(a["qkv"], ssq, self.P(pre + "attention.qk_norm.query_norm.scale"),
                                 self.P(pre + "attention.qk_norm.key_norm.scale"), cos, sin, a["q"] if train else None,
                                 a["k"] if train else None, a["rrms"] if train else None, a["a"], a["lse"], B, Hh, N, 64, rot, 64**-0.5)
            elif fused_qk:
                _must(ops.gemm_nt_qk_norm_rope(a["xm1"], sh[pre + "attention.qkv.weight|f"],
                                                self.P(pre + "attention.qk_norm.query_norm.scale"),
                                                self.P(pre + "attention.qk_norm.key_norm.scale"), cos, sin, a["qkv"], a["q"], a["k"],
                                                a["rrms"], B, N, Hh, 64, rot))
            else:
                ops.gemm_nt(a["xm1"], sh[pre + "attention.qkv.weight|f"], a["qkv"])
                ops.qk_norm_rope_fwd(a["qkv"], self.P(pre + "attention.qk_norm.query_norm.scale"),
                                     self.P(pre + "attention.qk_norm.key_norm.scale"), cos, sin, a["q"], a["k"], None, a["rrms"], B, N,
                                     Hh, 64, rot)
            if "ssq_all" not in w:
                ops.attn_fwd_qkv(a["q"], a["k"], a["qkv"], a["a"], a["lse"], B, Hh, N, 64, 64**-0.5)
            _must(ops.ln_modulate_gemm_fwd(a["a"], sh[pre + "attention.proj_out.weight|f"], xin, mod[:, mo + 2 * D : mo + 3 * D],
                                            self.P(pre + "norm_2.weight"), self.P(pre + "norm_2.bias"), mod[:, mo + 3 * D : mo + 4 * D],
                                            mod[:, mo + 4 * D : mo + 5 * D], N, 1e-5, a["t1"], a["x1"], a["xm2"], a["mean2"], a["rstd2"]))
            if not ops.gemm_nt_swiglu(a["xm2"], sh[pre + "mlp_input.0.weight|g"], a["u"] if train else None, a["h"]):
                raise RuntimeError("the row-complete path needs the fused MLP-up kernel")
            if i + 1 < L:
                nx, mn, nxt = lay(i + 1), (i + 1) * 6 * D, f"layers.{i + 1}."
                lnw, lnb, eps = self.P(nxt + "norm_1.weight"), self.P(nxt + "norm_1.bias"), 1e-5
                xm_n, mu_n, rs_n = nx["xm1"], nx["mean1"], nx["rstd1"]
            else:
                mn, lnw, lnb, eps = L * 6 * D, None, None, 1e-6
                xm_n, mu_n, rs_n = w["xf"], w["meanf"], w["rstdf"]
            _must(ops.ln_modulate_gemm_fwd(a["h"], sh[pre + "mlp_input.2.weight|f"], a["x1"], mod[:, mo + 5 * D : mo + 6 * D], lnw, lnb,
                                            mod[:, mn : mn + D], mod[:, mn + D : mn + 2 * D], N, eps, a["t2"], xbuf(i + 1), xm_n, mu_n,
                                            rs_n))

    # ------------------------------------------------------------------ backward
    def feature(self, k: int) -> Tensor:
        """output of block k of the last train-mode forward: the residual stream after the block, bf16 [B, N, D] (a view of
        the workspace, valid until the next forward) -- what a forward hook on ``layers[k]`` observes in the reference"""
        assert self._train, "block outputs are only kept by the train-mode launch sequence"
        B, _, _, _, _, N, _, _, _ = self.geo
        return self.ws["x"][k + 1].view(B, N, self.d.inner_dim)

    def backward(self, dpred: Tensor, dfeats: dict[int, Tensor] | None = None) -> None:
        """accumulates d(loss)/d(param) into the flat gradient arena (+=) for the last train-mode forward.
        dfeats: {block index k: gradient of an auxiliary loss w.r.t. feature(k)} (RePA), added to the residual-stream gradient."""
        assert self._train and self.grads is not None
        dfeats = {k: g.reshape(-1, self.d.inner_dim).to(torch.bfloat16).contiguous() for k, g in (dfeats or {}).items()}
        d, w, sh = self.d, self.ws, self.sh
        B, H, W, gh, gw, N, M, Bp, Fo = self.geo
        D, E, L, Hh = d.inner_dim, d.embedding_dim, d.depth, d.num_heads
        F = d.mlp_ratio * D
        cos, sin = self._rope[(gh, gw)]
        rot = sum(d.rope_axes_dim)
        mod, dmod, xs = w["mod"], w["dmod32"], w["x"]  # dmod: f32 accumulator, cast to bf16 (w["dmod"]) after the loop
        dmod[:B].zero_()
        Fo8 = _rup(Fo, 8)

        # head: last linear + final adaLN
        ops.patchify(dpred, w["dO"], d.patch_size, ops.PATCH_PPC)
        gl = self.G("last_layer.linear.weight")
        det = w.get("det_scr")
        if Fo == Fo8:
            ops.gemm_tn(w["dO"], w["xf"], gl, M=Fo, N=D, scratch=det)
        else:
            w["scr_last"].zero_()
            ops.gemm_tn(w["dO"], w["xf"], w["scr_last"], M=Fo8, N=D, scratch=det)
            ops.reduce_rows_f32(w["scr_last"], gl, 1, Fo * D)
        ops.colsum(w["dO"], self.G("last_layer.linear.bias"), M, Fo, scratch=det)
        fused = self._row_gemms(M, N)
        if not fused:
            ops.gemm_nt(w["dO"], sh["last_layer.linear.weight|t"], w["dxm"], M=M, N=D, K=self._ko)
        mo = L * 6 * D
        dx, dx_alt = w["dxa"], w["dxb"]
        # every LayerNorm-modulate backward also runs the backward of the gated residual that follows it in the chain
        # (x_new = x + gate * t): it has the residual-stream gradient dx in registers, so dt = gate * dx and dgate += dx * t
        # cost one extra row read / write instead of a separate pass over dx
        ml = (L - 1) * 6 * D
        if fused:  # (row-complete GEMMs: the LayerNorm backward is the epilogue of the data-gradient GEMM in front of it)
            _must(ops.ln_modulate_gemm_bwd(w["dO"], sh["last_layer.linear.weight|t"], xs[L], None, None, mod[:, mo : mo + D], N,
                                            w["meanf"], w["rstdf"], dfeats.get(L - 1), dx, dmod[:, mo : mo + D],
                                            dmod[:, mo + D : mo + 2 * D], None, gate_t=w["layers"][L - 1]["t2"],
                                            gate=mod[:, ml + 5 * D : ml + 6 * D], dt=w["wg"][L - 1]["dt2"],
                                            dgate=dmod[:, ml + 5 * D : ml + 6 * D], K=self._ko))
        else:
            ops.ln_modulate_bwd(w["dxm"], xs[L], None, None, mod[:, mo : mo + D], N, w["meanf"], w["rstdf"], dfeats.get(L - 1), dx,
                                dmod[:, mo : mo + D], dmod[:, mo + D : mo + 2 * D], None, gate_t=w["layers"][L - 1]["t2"],
                                gate=mod[:, ml + 5 * D : ml + 6 * D], dt=w["wg"][L - 1]["dt2"],
                                dgate=dmod[:, ml + 5 * D : ml + 6 * D])

        # The four weight-gradient GEMMs of a block are off the dependency chain (nothing downstream reads them), so they
        # run on a SIDE HIP stream: they overlap the HBM-bound kernels of the main chain (gate/SwiGLU/adaLN/QK-norm
        # backward), which leave the matrix pipes idle.  Their inputs (dt2, du, dt1, dqkv) live in per-block buffers so the
        # main chain never overwrites something the side stream still reads.
        main = torch.cuda.current_stream()
        side = self._side_stream()
        side.wait_stream(main)

        # workgroup cap of the side-stream weight gradients (of 256 CUs).  Four atomic launches per block: 96: 24.5, 128: 24.1, 160:
        # 24.4, 192: 24.8, 256: 25.8 ms/step (round 2).  The grouped launch (one per block, no atomics) wants the whole chip: every
        # kernel of the step is then one workgroup per CU and the two streams simply take turns (64: 23.8, 96: 22.4, 128: 22.3,
        # 192: 22.1, 256: 21.7 ms/step; the same launches inline on the main stream: 21.9)
        side_wgs = tuning.integer("DL_SIDE_WGS", 256 if w.get("tn_slab") is not None else 128)

        serial = tuning.on("DL_WGRAD_SERIAL")  # A/B switch: weight gradients inline on the main stream

        tn_slab = w.get("tn_slab")
        pending: list[tuple[Tensor, Tensor, Tensor]] = []  # (dy, x, g) of the current block, launched together once dqkv exists

        def wgrad(x_grad: Tensor, x_in: Tensor, gname: str) -> None:
            if tn_slab is not None:
                pending.append((x_grad, x_in, self.G(gname)))
                return
            if serial:
                ops.gemm_tn(x_grad, x_in, self.G(gname))
                return
            ev = main.record_event()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                ops.gemm_tn(x_grad, x_in, self.G(gname), max_wgs=side_wgs)

        def wgrad_flush() -> None:  # the block's four weight gradients: one atomics-free launch + fold (dl_gemm_tn_group)
            if not pending:
                return
            if serial:
                _must(ops.gemm_tn_group(pending, tn_slab))
            else:
                ev = main.record_event()
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    _must(ops.gemm_tn_group(pending, tn_slab, max_wgs=side_wgs))
            pending.clear()

        def fold_norm(partial: Tensor, gname: str) -> None:  # [B, 2, D] per-sample sums -> [w; b] gradients, off the chain
            if defer_fold:
                return
            ev = main.record_event()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                ops.reduce_rows_f32(partial, self.G(gname), B, 2 * D, clear=True)

        # Data parallel: the stacked adaLN matrix (L*6D x E, a quarter of the parameters) would only be final in _cond_bwd, i.e. its
        # 42 MB would be reduced after the backward has ended.  Every block's 6D rows of the modulation gradient are final when that
        # block's backward is (its MLP gate chunk was written by the block above), so the block's slice of the weight gradient is
        # computed right there on the side stream and handed to the reducer with the block's own range; the last three blocks ask
        # for an immediate flush, which leaves a few megabytes for finish().
        early_mod = self.reducer is not None and type(self) is DiTEngine and tuning.on("DL_DP_EARLY_MOD")
        self._early_mod_done = early_mod
        w_mod = self.layout.entries[self.mod_name][0]
        b_mod = self.layout.entries[self.layout.mod_b0][0]
        E = d.embedding_dim

        def block_done(i: int) -> None:
            if self.reducer is None:
                return
            if early_mod:
                r0, r1 = i * 6 * D, (i + 1) * 6 * D
                ev = main.record_event()
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    ops.cast2d_f32_to_bf16(w["dmod32"][:B, r0:r1], w["dmod"][:B, r0:r1])
                    ops.gemm_tn(w["dmod"][:, r0:r1], w["se"], self.grads[w_mod + r0 * E : w_mod + r1 * E].view(r1 - r0, E))
                    ops.colsum(w["dmod"][:, r0:r1], self.grads[b_mod + r0 : b_mod + r1], B, r1 - r0)
            ev_side = side.record_event()
            self.reducer.ready(*self.layer_ranges[i], extra_events=(ev_side,))
            if early_mod:
                self.reducer.ready(w_mod + i * 6 * D * E, w_mod + (i + 1) * 6 * D * E, flush=i < 3)

        native = self._native_blocks() and not serial and dx is w["dxa"]
        inline_wgrad = tuning.on("DL_WGRAD_INLINE")  # TUNING: the grouped weight gradients on the main stream
        defer_fold = fused and self.reducer is None  # LayerNorm-affine partials of every block: one batched fold after the loop
        for i in reversed(range(L)):
            if native:
                blk = self._block_args(i, True)
                blk.set(dfeat=dfeats.get(i - 1))
                ops.dit_block_bwd(blk, main.cuda_stream, main.cuda_stream if inline_wgrad else side.cuda_stream, side_wgs)
                block_done(i)
                continue
            a = w["layers"][i]
            g = w["wg"][i]
            pre = f"layers.{i}."
            mo = i * 6 * D
            # MLP branch
            wgrad(g["dt2"], a["h"], pre + "mlp_input.2.weight")  # dt2 / dgate: produced by the LayerNorm backward before
            if a["u"] is None:  # no saved pre-activations: the u tile is recomputed next to the dh tile, neither is written
                _must(ops.mlp_dswiglu_recompute(a["xm2"], sh[pre + "mlp_input.0.weight|g"], g["dt2"], sh[pre + "mlp_input.2.weight|t"],
                                                 g["du"]))
            else:
                ops.gemm_nt(g["dt2"], sh[pre + "mlp_input.2.weight|t"], w["dh"])
                ops.swiglu_bwd(w["dh"], a["u"], g["du"])
            wgrad(g["du"], a["xm2"], pre + "mlp_input.0.weight")
            if fused:
                _must(ops.ln_modulate_gemm_bwd(g["du"], sh[pre + "mlp_input.0.weight|t"], a["x1"], self.P(pre + "norm_2.weight"),
                                                self.P(pre + "norm_2.bias"), mod[:, mo + 3 * D : mo + 4 * D], N, a["mean2"], a["rstd2"],
                                                dx, dx_alt, dmod[:, mo + 3 * D : mo + 4 * D], dmod[:, mo + 4 * D : mo + 5 * D],
                                                w["dwb"][2 * i + 1], gate_t=a["t1"], gate=mod[:, mo + 2 * D : mo + 3 * D], dt=g["dt1"],
                                                dgate=dmod[:, mo + 2 * D : mo + 3 * D]))
            else:
                ops.gemm_nt(g["du"], sh[pre + "mlp_input.0.weight|t"], w["dxm"])
                ops.ln_modulate_bwd(w["dxm"], a["x1"], self.P(pre + "norm_2.weight"), self.P(pre + "norm_2.bias"),
                                    mod[:, mo + 3 * D : mo + 4 * D], N, a["mean2"], a["rstd2"], dx, dx_alt,
                                    dmod[:, mo + 3 * D : mo + 4 * D], dmod[:, mo + 4 * D : mo + 5 * D], w["dwb"][2 * i + 1],
                                    gate_t=a["t1"], gate=mod[:, mo + 2 * D : mo + 3 * D], dt=g["dt1"],
                                    dgate=dmod[:, mo + 2 * D : mo + 3 * D])
            fold_norm(w["dwb"][2 * i + 1], pre + "norm_2.weight")
            dx, dx_alt = dx_alt, dx
            # attention branch
            wgrad(g["dt1"], a["a"], pre + "attention.proj_out.weight")
            ops.gemm_nt(g["dt1"], sh[pre + "attention.proj_out.weight|t"], w["da"])
            v_in_place = ops.v_in_place(N)
            if w.get("qk_part") is not None:  # dQ, dK, dV token-major into dqkv; the QK-norm backward transforms q / k in place
                ops.attn_bwd_tok(a["q"], a["k"], a["qkv"], a["a"], w["da"], a["lse"], g["dqkv"], B, Hh, N, 64, 64**-0.5)
                _must(ops.qk_norm_rope_bwd_inplace(a["qkv"], self.P(pre + "attention.qk_norm.query_norm.scale"),
                                                    self.P(pre + "attention.qk_norm.key_norm.scale"), cos, sin, a["rrms"], g["dqkv"],
                                                    self.G(pre + "attention.qk_norm.query_norm.scale"), w["qk_part"], B, N, Hh, 64, rot))
            else:
                if a.get("ao") is not None:
                    Np = a["q"].shape[2]
                    ops.copy_rows3d(w["da"], N * D, D, w["dao"], Np * D, D, B, N, D)  # (the pad rows of dao stay zero)
                    ops.attn_bwd_ex(a["q"], a["k"], a["v"], a["ao"], w["dao"], a["lse"], w["dq"], w["dk"], w["dv"], B, Hh, Np, Np, 64,
                                    64**-0.5, w["kb"])
                elif v_in_place:  # dV goes straight into the v third of dqkv
                    ops.attn_bwd_qkv(a["q"], a["k"], a["qkv"], a["a"], w["da"], a["lse"], w["dq"], w["dk"], g["dqkv"], B, Hh, N, 64,
                                     64**-0.5)
                else:
                    ops.attn_bwd(a["q"], a["k"], a["v"], a["a"], w["da"], a["lse"], w["dq"], w["dk"], w["dv"], B, Hh, N, 64,
                                 64**-0.5)
                ops.qk_norm_rope_bwd(w["dq"], w["dk"], None if v_in_place else w["dv"], a["qkv"],
                                     self.P(pre + "attention.qk_norm.query_norm.scale"),
                                     self.P(pre + "attention.qk_norm.key_norm.scale"), cos, sin, a["rrms"], g["dqkv"],
                                     self.G(pre + "attention.qk_norm.query_norm.scale"), B, N, Hh, 64, rot)
            wgrad(g["dqkv"], a["xm1"], pre + "attention.qkv.weight")
            wgrad_flush()
            if not fused:
                ops.gemm_nt(g["dqkv"], sh[pre + "attention.qkv.weight|t"], w["dxm"])
            if i - 1 in dfeats:  # auxiliary-loss gradient on the output of block i-1 (= this block's input)
                ops.add_bf16(dx, dfeats[i - 1], dx)
            nxt = {}
            if i > 0:  # gated residual of the previous block's MLP branch
                mp = (i - 1) * 6 * D
                nxt = dict(gate_t=w["layers"][i - 1]["t2"], gate=mod[:, mp + 5 * D : mp + 6 * D], dt=w["wg"][i - 1]["dt2"],
                           dgate=dmod[:, mp + 5 * D : mp + 6 * D])
            if fused:
                _must(ops.ln_modulate_gemm_bwd(g["dqkv"], sh[pre + "attention.qkv.weight|t"], xs[i], self.P(pre + "norm_1.weight"),
                                                self.P(pre + "norm_1.bias"), mod[:, mo : mo + D], N, a["mean1"], a["rstd1"], dx, dx_alt,
                                                dmod[:, mo : mo + D], dmod[:, mo + D : mo + 2 * D], w["dwb"][2 * i], **nxt))
            else:
                ops.ln_modulate_bwd(w["dxm"], xs[i], self.P(pre + "norm_1.weight"), self.P(pre + "norm_1.bias"),
                                    mod[:, mo : mo + D], N, a["mean1"], a["rstd1"], dx, dx_alt, dmod[:, mo : mo + D],
                                    dmod[:, mo + D : mo + 2 * D], w["dwb"][2 * i], **nxt)
            fold_norm(w["dwb"][2 * i], pre + "norm_1.weight")
            dx, dx_alt = dx_alt, dx
            block_done(i)  # this block's gradient range is final once BOTH streams are past this point
        # Without a gradient reducer nothing reads the weight gradients before the optimizer, so the join with the side stream comes
        # LAST: the LayerNorm-affine fold and the conditioning backward (~0.3 ms of small main-stream kernels) run under the grouped
        # weight-gradient launch of the first block instead of behind it.  (Data parallel: _cond_bwd ends in reducer.finish(), which
        # must see every range final -- the join stays in front.)
        join_last = self.reducer is None and tuning.on("DL_JOIN_LAST")
        if not join_last:
            main.wait_stream(side)
        if defer_fold:  # dwb [2L, B, 2, D]: rows 2i -> norm_1 of block i, 2i + 1 -> norm_2; [w; b] of a norm are adjacent in the arena
            ent = self.layout.entries
            stride = ent[self.prefixes[1] + "norm_1.weight"][0] - ent[self.prefixes[0] + "norm_1.weight"][0] if L > 1 else 0
            for j, nm in enumerate(("norm_1.weight", "norm_2.weight")):
                ops.reduce_rows_batched_f32(w["dwb"][j], 2 * B * 2 * D, self.G(self.prefixes[0] + nm), stride, L, B, 2 * D)

        self._cond_bwd(dx)
        if join_last:
            main.wait_stream(side)

    def _cond_bwd(self, dx: Tensor, extra_demb: Tensor | None = None) -> None:
        """backward of _stem_fwd: dx = gradient of the patch-embedded tokens; the modulation gradient was accumulated in
        ws["dmod32"] by the block kernels.  extra_demb: f32 [>= B, E] further gradient of the time embedding (DDT's decoder)"""
        d, w, sh = self.d, self.ws, self.sh
        B, _, _, _, _, _, M, _, _ = self.geo
        D, E = d.inner_dim, d.embedding_dim
        dmod = w["dmod32"]
        det = w.get("det_scr")  # scratch of the bit-reproducible small GEMMs / column sums (None: the atomic forms)
        # stem: conv_proj weight gradient (no gradient flows to the input latents)
        Fi = d.input_channels * d.patch_size**2
        gc = self.G(self._conv_name).view(D, Fi)
        if Fi % 8 == 0:
            ops.gemm_tn(dx, w["tokP"], gc, M=D, N=Fi, scratch=det)
        else:
            w["scr_conv"].zero_()
            ops.gemm_tn(dx, w["tokP"], w["scr_conv"], M=D, N=_rup(Fi, 8), scratch=det)
            gc.add_(w["scr_conv"][:, :Fi])  # ragged edge (e.g. RGB p=2 -> 12 features): torch slice-add, not on the hot path

        # conditioning path: every adaLN linear at once, then the time MLP and the label table
        R = self.layout.mod_rows
        g_modw = self.grads[self.layout.entries[self.mod_name][0] :][: R * E].view(R, E)
        g_modb = self.grads[self.layout.entries[self.layout.mod_b0][0] :][:R]
        L6 = len(self.prefixes) * 6 * D if getattr(self, "_early_mod_done", False) else 0
        if L6:  # the blocks' rows were cast, multiplied and handed to the reducer as the blocks finished (backward: block_done)
            ops.cast2d_f32_to_bf16(dmod[:B, L6:], w["dmod"][:B, L6:])
            dmod = w["dmod"]
            ops.gemm_tn(dmod[:, L6:], w["se"], g_modw[L6:], scratch=det)
            ops.colsum(dmod[:, L6:], g_modb[L6:], B, R - L6, scratch=det)
        else:
            ops.cast_f32_to_bf16(dmod[:B], w["dmod"][:B])
            dmod = w["dmod"]
            ops.gemm_tn(dmod, w["se"], g_modw, scratch=det)
            ops.colsum(dmod, g_modb, B, R, scratch=det)
        ops.gemm_nt(dmod, sh["@mod|t"], w["dse"], M=B, N=E, K=R, scratch=det)
        table = d.n_classes is not None
        ops.cond_combine_bwd(w["dse"][:B], w["emb"][:B], self._yeff if table else None, w["demb"][:B], w["demb16"][:B],
                             self.G("label_embed.embedding.weight") if table else None)
        if extra_demb is not None:
            w["demb"][:B].add_(extra_demb[:B])
            ops.cast_f32_to_bf16(w["demb"][:B], w["demb16"][:B])
        ops.colsum(w["demb"], self.G("time_embed.2.bias"), B, E, scratch=det)
        ops.gemm_tn(w["demb16"], w["h1"], self.G("time_embed.2.weight"), scratch=det)
        ops.gemm_nt(w["demb16"], sh["time_embed.2.weight|t"], w["dh1"], M=B, N=E, K=E)
        ops.silu_bwd(w["dh1"][:B], w["pre1"][:B], w["dpre1"][:B])
        ops.gemm_tn(w["dpre1"], w["temb"], self.G("time_embed.0.weight"), scratch=det)
        ops.colsum(w["dpre1"], self.G("time_embed.0.bias"), B, E, scratch=det)
        if self.reducer is not None:
            if L6:
                w_mod = self.layout.entries[self.mod_name][0]
                self.reducer.ready(0, w_mod)
                self.reducer.ready(w_mod + L6 * E, self.layer_ranges[0][0])
            else:
                self.reducer.ready(0, self.layer_ranges[0][0])
            self.reducer.finish()
