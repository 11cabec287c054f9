// HBM-bound row kernels of the DiT block: adaLN (LayerNorm + modulate) fwd/bwd, gated-residual bwd,
// QK-RMSNorm + N-D RoPE + head split fwd/bwd, SwiGLU fwd/bwd.
// One 64-lane wave owns one token row; every lane moves 16-byte (8 x bf16) chunks; row statistics are
// wavefront shuffles (no LDS); per-sample column reductions (adaLN gradients) are register accumulators
// reduced across the workgroup's waves through LDS once per workgroup.
#include <stdlib.h>

#include "common.h"

#define MAXJ 2  // chunks of 8 per lane -> D <= 1024 (NJ=4 spills; wider rows need an LDS-staged variant)

template <int NJ>
__device__ __forceinline__ void load_row(const bf16_t* __restrict__ p, int D8, int lane, float (&v)[NJ][8]) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    if (c < D8) {
      unpack8(*(const u32x4_t*)(p + c * 8), v[j]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
    }
  }
}
template <int NJ>
__device__ __forceinline__ void load_row_f32(const float* __restrict__ p, int D8, int lane, float (&v)[NJ][8], float fill) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    if (p && c < D8) {
      *(f32x4_t*)&v[j][0] = *(const f32x4_t*)(p + c * 8);
      *(f32x4_t*)&v[j][4] = *(const f32x4_t*)(p + c * 8 + 4);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[j][e] = fill;
    }
  }
}
template <int NJ>
__device__ __forceinline__ void store_row(bf16_t* __restrict__ p, int D8, int lane, const float (&v)[NJ][8]) {
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    if (c < D8) *(u32x4_t*)(p + c * 8) = pack8(v[j]);
  }
}

// ======================================================================== LayerNorm + modulate, forward
template <int NJ>
__global__ __launch_bounds__(256) void ln_mod_fwd_k(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ b, const bf16_t* __restrict__ scale,
                                                    const bf16_t* __restrict__ shift, int64_t ld_mod,
                                                    int64_t rows_per_mod, float eps, bf16_t* __restrict__ out,
                                                    float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                    const bf16_t* __restrict__ t, const bf16_t* __restrict__ gate,
                                                    int64_t ld_gate, bf16_t* __restrict__ x_out, int64_t M, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D8 = D >> 3;
  const float invD = 1.0f / (float)D;
  float wv[NJ][8], bv[NJ][8];
  load_row_f32<NJ>(w, D8, lane, wv, 1.0f);
  load_row_f32<NJ>(b, D8, lane, bv, 0.0f);
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
    float xv[NJ][8], sc[NJ][8], sh[NJ][8];
    load_row<NJ>(x + row * D, D8, lane, xv);
    const int64_t g = row / rows_per_mod;
    if (t) {  // fused gated residual of the previous sub-layer: x <- x + gate * t (mmdit.py:302,308), stored as bf16
      float tv[NJ][8], gv[NJ][8];
      load_row<NJ>(t + row * D, D8, lane, tv);
      load_row<NJ>(gate + g * ld_gate, D8, lane, gv);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[j][e] = bf2f(f2bf(xv[j][e] + gv[j][e] * tv[j][e]));  // statistics of what is stored
      store_row<NJ>(x_out + row * D, D8, lane, xv);
    }
    load_row<NJ>(scale + g * ld_mod, D8, lane, sc);
    load_row<NJ>(shift + g * ld_mod, D8, lane, sh);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) s += xv[j][e];
    const float mu = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const bool on = (lane + 64 * j) < D8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = on ? (xv[j][e] - mu) : 0.f;
        q += d * d;
      }
    }
    const float rs = rsqrtf(wave_sum(q) * invD + eps);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float y = (xv[j][e] - mu) * rs * wv[j][e] + bv[j][e];
        xv[j][e] = y * (1.0f + sc[j][e]) + sh[j][e];
      }
    store_row<NJ>(out + row * D, D8, lane, xv);
    if (lane == 0) {
      mean_o[row] = mu;
      rstd_o[row] = rs;
    }
  }
}

extern "C" int dl_ln_modulate_fwd(const void* x, const float* w, const float* b, const void* scale, const void* shift,
                                  int64_t ld_mod, int64_t rows_per_mod, float eps, void* out, float* mean,
                                  float* rstd, const void* t, const void* gate, int64_t ld_gate, void* x_out, int64_t M,
                                  int64_t D, dl_stream_t stream) {
  DL_CHECK_ARG(x && scale && shift && out && mean && rstd && M > 0, "dl_ln_modulate_fwd: null operand");
  DL_CHECK_ARG(!t || (gate && x_out && ld_gate % 8 == 0 && (((uintptr_t)t | (uintptr_t)gate | (uintptr_t)x_out) & 15) == 0),
               "dl_ln_modulate_fwd: the fused gated residual needs t, gate and x_out (16-byte aligned)");
  DL_CHECK_ARG((w == nullptr) == (b == nullptr), "dl_ln_modulate_fwd: w and b must both be given or both NULL");
  DL_CHECK_ARG(D % 8 == 0 && D <= 512 * MAXJ && ld_mod % 8 == 0 && rows_per_mod > 0, "dl_ln_modulate_fwd: D=%lld",
               (long long)D);
  DL_CHECK_ARG((((uintptr_t)x | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)out) & 15) == 0,
               "dl_ln_modulate_fwd: 16-byte alignment");
  const int nj = cdiv(D, 512);
  int grid = cdiv(M, 4);
  if (grid > 4096) grid = 4096;
#define LAUNCH(NJ)                                                                                              \
  hipLaunchKernelGGL(ln_mod_fwd_k<NJ>, grid, 256, 0, (hipStream_t)stream, (const bf16_t*)x, w, b,                \
                     (const bf16_t*)scale, (const bf16_t*)shift, ld_mod, rows_per_mod, eps, (bf16_t*)out, mean, \
                     rstd, (const bf16_t*)t, (const bf16_t*)gate, ld_gate, (bf16_t*)x_out, M, (int)D)
  if (nj == 1) LAUNCH(1);
  else LAUNCH(2);
#undef LAUNCH
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ======================================================================== LayerNorm + modulate, backward
// Two launches: (1) `split` 256-thread workgroups per modulation group (= one sample's tokens) produce dx rows and
// per-workgroup partial column sums [4][D] (256 threads / ~140 VGPRs / 6 KiB LDS: small enough to co-reside with a
// GEMM workgroup that owns most of the CU -- the side-stream wgrad overlap); (2) a tiny kernel folds the partials.
#define LNB_WAVES 8
#define LNB1_WAVES 4
// Column sums: only three are independent.  With d = dout, xh = (x - mu) rstd, y = xh w + b, dy = d (1 + scale):
//   dscale = sum d y = w S2 + b S1,  dshift = S1,  dw = (1 + scale) S2,  db = (1 + scale) S1,  dgate = S3
//   S1 = sum_rows d,  S2 = sum_rows d xh,  S3 = sum_rows dx_new t
// so the row loop carries 3 accumulators and ONE constant vector g = (1 + scale) w instead of 5 + 3: 179 -> ~110 VGPRs.  That
// matters beside the side-stream weight-gradient GEMM (152 VGPRs x 2 waves per SIMD): two of these waves fit into the 208
// registers it leaves per SIMD instead of one.  PF: software prefetch of the next row (worth it only at one wave per SIMD).
template <int NJ, bool PF>
__device__ __forceinline__ void ln_mod_bwd_body(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ x,
                                                    const float* __restrict__ w, const float* __restrict__ b,
                                                    const bf16_t* __restrict__ scale, int64_t ld_mod,
                                                    int64_t rows_per_mod, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, const bf16_t* __restrict__ dres,
                                                    bf16_t* __restrict__ dx, float* __restrict__ dscale,
                                                    float* __restrict__ dshift, int64_t ld_dmod, float* __restrict__ dwb,
                                                    const bf16_t* __restrict__ gt, const bf16_t* __restrict__ ggate,
                                                    int64_t ld_gate, bf16_t* __restrict__ gdt, float* __restrict__ dgate,
                                                    int split, int64_t M, int D) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = (float*)smem;  // [3][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D8 = D >> 3;
  const float invD = 1.0f / (float)D;
  const int64_t g = blockIdx.x / split;
  const int sp = blockIdx.x - (int)g * split;
  const int64_t rows_per_wg = rows_per_mod / split;
  for (int i = threadIdx.x; i < 3 * D; i += 256) red[i] = 0.f;
  float gw[NJ][8], gv[NJ][8];
  {
    float wv[NJ][8], sc[NJ][8];
    load_row_f32<NJ>(w, D8, lane, wv, 1.0f);
    load_row<NJ>(scale + g * ld_mod, D8, lane, sc);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) gw[j][e] = (1.0f + sc[j][e]) * wv[j][e];
  }
  if (gt) load_row<NJ>(ggate + g * ld_gate, D8, lane, gv);
  float a_s1[NJ][8], a_s2[NJ][8], a_s3[NJ][8];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) a_s1[j][e] = a_s2[j][e] = a_s3[j][e] = 0.f;

  const int64_t row_begin = g * rows_per_mod + sp * rows_per_wg;
  const int64_t row_end = row_begin + rows_per_wg < M ? row_begin + rows_per_wg : M;
  u32x4_t pd[NJ], px[NJ], pr[NJ], pt[NJ];
  float pmu = 0.f, prs = 0.f;
  auto fetch = [&](int64_t row) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      if (c < D8) {
        pd[j] = *(const u32x4_t*)(dout + row * D + c * 8);
        px[j] = *(const u32x4_t*)(x + row * D + c * 8);
        if (dres) pr[j] = *(const u32x4_t*)(dres + row * D + c * 8);
        if (gt) pt[j] = *(const u32x4_t*)(gt + row * D + c * 8);
      }
    }
    pmu = mean[row];
    prs = rstd[row];
  };
  int64_t row = row_begin + wave;
  if (PF && row < row_end) fetch(row);
  for (; row < row_end; row += LNB1_WAVES) {
    if (!PF) fetch(row);
    float dv[NJ][8], xv[NJ][8], rv[NJ][8];
    u32x4_t tq[NJ];  // this row's t (gate fusion), saved before the prefetch of the next row overwrites pt
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const bool on = (lane + 64 * j) < D8;
      tq[j] = pt[j];
      if (on) {
        unpack8(pd[j], dv[j]);
        unpack8(px[j], xv[j]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (!on) dv[j][e] = xv[j][e] = 0.f;
        rv[j][e] = 0.f;
      }
      if (on && dres) unpack8(pr[j], rv[j]);
    }
    const float mu = pmu, rs = prs;
    if (PF && row + LNB1_WAVES < row_end) fetch(row + LNB1_WAVES);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const bool on = (lane + 64 * j) < D8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = on ? (xv[j][e] - mu) * rs : 0.f;
        const float d = dv[j][e];
        a_s1[j][e] += d;
        a_s2[j][e] += d * xh;
        const float dxh = d * gw[j][e];
        s1 += dxh;
        s2 += dxh * xh;
        xv[j][e] = xh;
        dv[j][e] = dxh;
      }
    }
    const float c1 = wave_sum(s1) * invD, c2 = wave_sum(s2) * invD;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) rv[j][e] += rs * (dv[j][e] - c1 - xv[j][e] * c2);
    store_row<NJ>(dx + row * D, D8, lane, rv);
    if (gt) {
      // fused backward of the gated residual that FOLLOWS in the chain (x_new = x + gate * t, mmdit.py:302,308): this row
      // of dx is its upstream gradient -> dt = gate * dx (to the projection / MLP-down dgrad), dgate += dx * t.
      // dx is taken as stored (bf16), exactly what a separate pass would re-read.
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const bool on = (lane + 64 * j) < D8;
        float tv[8];
        if (on) unpack8(tq[j], tv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dxr = bf2f(f2bf(rv[j][e]));
          a_s3[j][e] += on ? dxr * tv[e] : 0.f;
          rv[j][e] = dxr * gv[j][e];
        }
      }
      store_row<NJ>(gdt + row * D, D8, lane, rv);
    }
  }

  // cross-wave reduction of the three column sums: LDS float atomics into ONE [3][D] slab
  __syncthreads();  // zero-fill above happened before the row loop
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    if (c < D8) {
      float* base = red + c * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        atomicAdd(base + e, a_s1[j][e]);
        atomicAdd(base + D + e, a_s2[j][e]);
        if (gt) atomicAdd(base + 2 * D + e, a_s3[j][e]);
      }
    }
  }
  __syncthreads();
  // the `split` workgroups of one sample meet in global f32 accumulators (<= split-way contention per address): the
  // modulation gradients land in the f32 image of dmod, the affine gradients in the per-sample partial [groups, 2, D]
  for (int col = threadIdx.x; col < D; col += 256) {
    const float S1 = red[col], S2 = red[D + col];
    const float wc = w ? w[col] : 1.0f, bc = b ? b[col] : 0.0f, sc1 = 1.0f + bf2f(scale[g * ld_mod + col]);
    unsafeAtomicAdd(dscale + g * ld_dmod + col, wc * S2 + bc * S1);
    unsafeAtomicAdd(dshift + g * ld_dmod + col, S1);
    if (gt) unsafeAtomicAdd(dgate + g * ld_dmod + col, red[2 * D + col]);
    if (dwb) {
      unsafeAtomicAdd(dwb + (size_t)g * 2 * D + col, sc1 * S2);
      unsafeAtomicAdd(dwb + (size_t)g * 2 * D + D + col, sc1 * S1);
    }
  }
}

#define LNB_ARGS                                                                                                              \
  const bf16_t *__restrict__ dout, const bf16_t *__restrict__ x, const float *__restrict__ w, const float *__restrict__ b,            \
      const bf16_t *__restrict__ scale, int64_t ld_mod, int64_t rows_per_mod, const float *__restrict__ mean,                         \
      const float *__restrict__ rstd, const bf16_t *__restrict__ dres, bf16_t *__restrict__ dx, float *__restrict__ dscale,           \
      float *__restrict__ dshift, int64_t ld_dmod, float *__restrict__ dwb, const bf16_t *__restrict__ gt,                            \
      const bf16_t *__restrict__ ggate, int64_t ld_gate, bf16_t *__restrict__ gdt, float *__restrict__ dgate, int split, int64_t M, int D
#define LNB_PASS dout, x, w, b, scale, ld_mod, rows_per_mod, mean, rstd, dres, dx, dscale, dshift, ld_dmod, dwb, gt, ggate, ld_gate, gdt, dgate, split, M, D
// D <= 512: capped at 104 VGPRs (two waves per SIMD beside the 152-register weight-gradient GEMM, four alone)
template <bool PF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void ln_mod_bwd_k(LNB_ARGS) {
  ln_mod_bwd_body<1, PF>(LNB_PASS);
}
template <bool PF>
__global__ __launch_bounds__(256) void ln_mod_bwd_wide_k(LNB_ARGS) {
  ln_mod_bwd_body<2, PF>(LNB_PASS);
}

extern "C" int dl_ln_modulate_bwd(const void* dout, const void* x, const float* w, const float* b, const void* scale,
                                  int64_t ld_mod, int64_t rows_per_mod, const float* mean, const float* rstd,
                                  const void* dres, void* dx, float* dscale, float* dshift, int64_t ld_dmod,
                                  float* dwb_partial, const void* gate_t, const void* gate, int64_t ld_gate, void* dt,
                                  float* dgate, int64_t M, int64_t D, dl_stream_t stream) {
  DL_CHECK_ARG(dout && x && scale && mean && rstd && dx && dscale && dshift && M > 0, "dl_ln_modulate_bwd: null operand");
  DL_CHECK_ARG((w == nullptr) == (b == nullptr), "dl_ln_modulate_bwd: w and b must both be given or both NULL");
  DL_CHECK_ARG(D % 8 == 0 && D <= 512 * MAXJ && ld_mod % 8 == 0 && rows_per_mod > 0 && M % rows_per_mod == 0,
               "dl_ln_modulate_bwd: D=%lld M=%lld rows_per_mod=%lld", (long long)D, (long long)M, (long long)rows_per_mod);
  DL_CHECK_ARG((((uintptr_t)dout | (uintptr_t)x | (uintptr_t)scale | (uintptr_t)dx | (uintptr_t)dres) & 15) == 0,
               "dl_ln_modulate_bwd: 16-byte alignment");
  const int nj = cdiv(D, 512);
  const int groups = (int)(M / rows_per_mod);
  // 64 rows per workgroup; 32 when that would leave the launch under one workgroup per CU (a few thousand rows)
  const int rows_wg = (rows_per_mod % 64 == 0 && rows_per_mod >= 128 && M / 64 < 256) ? 32 : 64;
  const int split = (rows_per_mod % rows_wg == 0) ? (int)(rows_per_mod / rows_wg) : 1;
  DL_CHECK_ARG(!gate_t || (gate && dt && dgate && ld_gate % 8 == 0 && (((uintptr_t)gate_t | (uintptr_t)gate | (uintptr_t)dt) & 15) == 0),
               "dl_ln_modulate_bwd: the fused gate backward needs gate_t, gate, dt and dgate (16-byte aligned)");
  const size_t lds = (size_t)3 * D * sizeof(float);
  const bool pf = false;  // (the two-row prefetch ring of ln_mod_bwd_k<true> measured +0.3 ms per step: round 1)
#define LAUNCH(NJ, PF)                                                                                                        \
  hipLaunchKernelGGL((KERN_##NJ<PF>), groups * split, 256, lds, (hipStream_t)stream, (const bf16_t*)dout, (const bf16_t*)x, \
                     w, b, (const bf16_t*)scale, ld_mod, rows_per_mod, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx,          \
                     dscale, dshift, ld_dmod, dwb_partial, (const bf16_t*)gate_t, (const bf16_t*)gate, ld_gate,              \
                     (bf16_t*)dt, dgate, split, M, (int)D)
#define KERN_1 ln_mod_bwd_k
#define KERN_2 ln_mod_bwd_wide_k
  if (nj == 1 && pf) LAUNCH(1, true);
  else if (nj == 1) LAUNCH(1, false);
  else if (pf) LAUNCH(2, true);
  else LAUNCH(2, false);
#undef LAUNCH
#undef KERN_1
#undef KERN_2
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ======================================================================== LayerNorm + modulate, backward, PER-TOKEN modulation
// DDT's decoder conditions every token on its own vector (ddt.py:423-424 -> Modulation on [B, S, D]): scale / shift / gate have
// one row per token, so their gradients are per-row outputs (written as bf16 straight into the modulation-gradient matrix, no
// accumulation) and only the affine gradients need a column reduction: registers -> LDS -> one of n_part f32 partial slabs.
template <int NJ>
__global__ __launch_bounds__(256) void ln_mod_bwd_tok_k(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ x,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        const bf16_t* __restrict__ scale, int64_t ld_mod,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
                                                        bf16_t* __restrict__ dscale, bf16_t* __restrict__ dshift, int64_t ld_dmod,
                                                        float* __restrict__ dwb, int n_part, const bf16_t* __restrict__ gt,
                                                        const bf16_t* __restrict__ ggate, int64_t ld_gate,
                                                        bf16_t* __restrict__ gdt, bf16_t* __restrict__ dgate, int rows_per_wg,
                                                        int64_t M, int D) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = (float*)smem;  // [2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D8 = D >> 3;
  const float invD = 1.0f / (float)D;
  for (int i = threadIdx.x; i < 2 * D; i += 256) red[i] = 0.f;
  float wv[NJ][8], bv[NJ][8], a_dw[NJ][8], a_db[NJ][8];
  load_row_f32<NJ>(w, D8, lane, wv, 1.0f);
  load_row_f32<NJ>(b, D8, lane, bv, 0.0f);
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) a_dw[j][e] = a_db[j][e] = 0.f;
  const int64_t row_begin = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t row_end = row_begin + rows_per_wg < M ? row_begin + rows_per_wg : M;
  for (int64_t row = row_begin + wave; row < row_end; row += 4) {
    float dv[NJ][8], xv[NJ][8], rv[NJ][8], sc[NJ][8], ov[NJ][8];
    load_row<NJ>(dout + row * D, D8, lane, dv);
    load_row<NJ>(x + row * D, D8, lane, xv);
    load_row<NJ>(scale + row * ld_mod, D8, lane, sc);
    if (dres) load_row<NJ>(dres + row * D, D8, lane, rv);
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const bool on = (lane + 64 * j) < D8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = on ? (xv[j][e] - mu) * rs : 0.f;
        const float d = on ? dv[j][e] : 0.f;
        ov[j][e] = d * (xh * wv[j][e] + bv[j][e]);  // dscale of this token
        const float dy = d * (1.0f + sc[j][e]);
        a_dw[j][e] += dy * xh;
        a_db[j][e] += dy;
        const float dxh = dy * wv[j][e];
        s1 += dxh;
        s2 += dxh * xh;
        xv[j][e] = xh;
        dv[j][e] = dxh;
        if (!dres) rv[j][e] = 0.f;
      }
    }
    store_row<NJ>(dscale + row * ld_dmod, D8, lane, ov);
    {  // dshift of this token = dout (re-read: the registers now hold dxh)
      float sv[NJ][8];
      load_row<NJ>(dout + row * D, D8, lane, sv);
      store_row<NJ>(dshift + row * ld_dmod, D8, lane, sv);
    }
    const float c1 = wave_sum(s1) * invD, c2 = wave_sum(s2) * invD;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) rv[j][e] += rs * (dv[j][e] - c1 - xv[j][e] * c2);
    store_row<NJ>(dx + row * D, D8, lane, rv);
    if (gt) {  // fused backward of the gated residual that follows in the chain, per-token gate
      float tv[NJ][8], gv[NJ][8];
      load_row<NJ>(gt + row * D, D8, lane, tv);
      load_row<NJ>(ggate + row * ld_gate, D8, lane, gv);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dxr = bf2f(f2bf(rv[j][e]));
          tv[j][e] = dxr * tv[j][e];   // dgate of this token
          rv[j][e] = dxr * gv[j][e];   // dt
        }
      store_row<NJ>(dgate + row * ld_dmod, D8, lane, tv);
      store_row<NJ>(gdt + row * D, D8, lane, rv);
    }
  }
  if (!dwb) return;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    if (c < D8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        atomicAdd(red + c * 8 + e, a_dw[j][e]);
        atomicAdd(red + D + c * 8 + e, a_db[j][e]);
      }
    }
  }
  __syncthreads();
  float* dst = dwb + (size_t)(blockIdx.x % n_part) * 2 * D;
  for (int i = threadIdx.x; i < 2 * D; i += 256) unsafeAtomicAdd(dst + i, red[i]);
}

extern "C" int dl_ln_modulate_bwd_tok(const void* dout, const void* x, const float* w, const float* b, const void* scale,
                                      int64_t ld_mod, const float* mean, const float* rstd, const void* dres, void* dx,
                                      void* dscale, void* dshift, int64_t ld_dmod, float* dwb_partial, int64_t n_part,
                                      const void* gate_t, const void* gate, int64_t ld_gate, void* dt, void* dgate, int64_t M,
                                      int64_t D, dl_stream_t stream) {
  DL_CHECK_ARG(dout && x && scale && mean && rstd && dx && dscale && dshift && M > 0, "dl_ln_modulate_bwd_tok: null operand");
  DL_CHECK_ARG((w == nullptr) == (b == nullptr) && (w == nullptr) == (dwb_partial == nullptr) && (!dwb_partial || n_part > 0),
               "dl_ln_modulate_bwd_tok: w, b and dwb_partial must all be given or all NULL");
  DL_CHECK_ARG(D % 8 == 0 && D <= 512 * MAXJ && ld_mod % 8 == 0 && ld_dmod % 8 == 0, "dl_ln_modulate_bwd_tok: D=%lld", (long long)D);
  DL_CHECK_ARG((((uintptr_t)dout | (uintptr_t)x | (uintptr_t)scale | (uintptr_t)dx | (uintptr_t)dres | (uintptr_t)dscale |
                 (uintptr_t)dshift) & 15) == 0, "dl_ln_modulate_bwd_tok: 16-byte alignment");
  DL_CHECK_ARG(!gate_t || (gate && dt && dgate && ld_gate % 8 == 0 &&
                           (((uintptr_t)gate_t | (uintptr_t)gate | (uintptr_t)dt | (uintptr_t)dgate) & 15) == 0),
               "dl_ln_modulate_bwd_tok: the fused gate backward needs gate_t, gate, dt and dgate (16-byte aligned)");
  const int rows_per_wg = 64;
  const int grid = cdiv(M, rows_per_wg);
  const size_t lds = (size_t)2 * D * sizeof(float);
  const int nj = cdiv(D, 512);
#define LAUNCH(NJ)                                                                                                              \
  hipLaunchKernelGGL(ln_mod_bwd_tok_k<NJ>, grid, 256, lds, (hipStream_t)stream, (const bf16_t*)dout, (const bf16_t*)x, w, b,        \
                     (const bf16_t*)scale, ld_mod, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, (bf16_t*)dscale, (bf16_t*)dshift,  \
                     ld_dmod, dwb_partial, (int)n_part, (const bf16_t*)gate_t, (const bf16_t*)gate, ld_gate, (bf16_t*)dt,            \
                     (bf16_t*)dgate, rows_per_wg, M, (int)D)
  if (nj == 1) LAUNCH(1);
  else LAUNCH(2);
#undef LAUNCH
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ======================================================================== gated residual, backward
template <int NJ>
__global__ __launch_bounds__(512) void gate_bwd_k(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ t,
                                                  const bf16_t* __restrict__ gate, int64_t ld_mod,
                                                  int64_t rows_per_mod, bf16_t* __restrict__ dt,
                                                  float* __restrict__ dgate, int64_t ld_dmod, int64_t M, int D) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = (float*)smem;  // [LNB_WAVES][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D8 = D >> 3;
  const int64_t g = blockIdx.x;
  float gv[NJ][8], acc[NJ][8];
  load_row<NJ>(gate + g * ld_mod, D8, lane, gv);
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
  const int64_t row_end = (g + 1) * rows_per_mod < M ? (g + 1) * rows_per_mod : M;
  for (int64_t row = g * rows_per_mod + wave; row < row_end; row += LNB_WAVES) {
    float dv[NJ][8], tv[NJ][8];
    load_row<NJ>(dout + row * D, D8, lane, dv);
    load_row<NJ>(t + row * D, D8, lane, tv);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        acc[j][e] += dv[j][e] * tv[j][e];
        dv[j][e] *= gv[j][e];
      }
    store_row<NJ>(dt + row * D, D8, lane, dv);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    if (c < D8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(size_t)wave * D + c * 8 + e] = acc[j][e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < D; i += 512) {
    float s = 0.f;
#pragma unroll
    for (int wv_ = 0; wv_ < LNB_WAVES; ++wv_) s += red[(size_t)wv_ * D + i];
    dgate[g * ld_dmod + i] = s;
  }
}

extern "C" int dl_gate_bwd(const void* dout, const void* t, const void* gate, int64_t ld_mod, int64_t rows_per_mod,
                           void* dt, float* dgate, int64_t ld_dmod, int64_t M, int64_t D, dl_stream_t stream) {
  DL_CHECK_ARG(dout && t && gate && dt && dgate && M > 0, "dl_gate_bwd: null operand");
  DL_CHECK_ARG(D % 8 == 0 && D <= 512 * MAXJ && ld_mod % 8 == 0 && rows_per_mod > 0 && M % rows_per_mod == 0,
               "dl_gate_bwd: bad dims");
  DL_CHECK_ARG((((uintptr_t)dout | (uintptr_t)t | (uintptr_t)gate | (uintptr_t)dt) & 15) == 0, "dl_gate_bwd: alignment");
  const int nj = cdiv(D, 512);
  const int groups = (int)(M / rows_per_mod);
  const size_t lds = (size_t)LNB_WAVES * D * sizeof(float);
#define LAUNCH(NJ)                                                                                               \
  hipLaunchKernelGGL(gate_bwd_k<NJ>, groups, 512, lds, (hipStream_t)stream, (const bf16_t*)dout, (const bf16_t*)t, \
                     (const bf16_t*)gate, ld_mod, rows_per_mod, (bf16_t*)dt, dgate, ld_dmod, M, (int)D)
  if (nj == 1) LAUNCH(1);
  else LAUNCH(2);
#undef LAUNCH
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ======================================================================== QK RMSNorm + RoPE + head split
// forward: one wave per token; lane c owns elements [8c, 8c+8) of the q, k and v thirds of the qkv row.
template <int NJ>
__global__ __launch_bounds__(256) void qk_norm_rope_fwd_k(const bf16_t* __restrict__ qkv, const float* __restrict__ sq,
                                                          const float* __restrict__ sk, const float* __restrict__ cs,
                                                          const float* __restrict__ sn, bf16_t* __restrict__ qo,
                                                          bf16_t* __restrict__ ko, bf16_t* __restrict__ vo,
                                                          float* __restrict__ rrms, int64_t M, int N, int H, int dh,
                                                          int rot, float eps, const int* __restrict__ pos, int n_dst,
                                                          int n_off) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = H * dh, D8 = D >> 3;
  const float invD = 1.0f / (float)D;
  float wq[NJ][8], wk[NJ][8];
  load_row_f32<NJ>(sq, D8, lane, wq, 1.0f);
  load_row_f32<NJ>(sk, D8, lane, wk, 1.0f);
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
    const bf16_t* p = qkv + row * 3 * D;
    float q[NJ][8], k[NJ][8], v[NJ][8];
    load_row<NJ>(p, D8, lane, q);
    load_row<NJ>(p + D, D8, lane, k);
    if (vo) load_row<NJ>(p + 2 * D, D8, lane, v);  // vo == NULL: the attention reads V in place (dl_attn_fwd_sv)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1 += q[j][e] * q[j][e];
        s2 += k[j][e] * k[j][e];
      }
    const float rq = rsqrtf(wave_sum(s1) * invD + eps), rk = rsqrtf(wave_sum(s2) * invD + eps);
    const int64_t b = row / N;
    const int n = (int)(row - b * N);
    const int nt = pos ? pos[row] : n;  // table row: the token's position on the grid (SPRINT keeps a subset of the tokens)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      if (c >= D8) continue;
      const int col = c * 8, h = col / dh, d0 = col - h * dh;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        q[j][e] = q[j][e] * rq * wq[j][e];
        k[j][e] = k[j][e] * rk * wk[j][e];
      }
      if (d0 < rot) {  // rot is a multiple of 8: a chunk is either fully rotary or fully pass-through
        const f32x4_t cc = *(const f32x4_t*)(cs + (int64_t)nt * (rot >> 1) + (d0 >> 1));
        const f32x4_t ss = *(const f32x4_t*)(sn + (int64_t)nt * (rot >> 1) + (d0 >> 1));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float qa = q[j][2 * i], qb = q[j][2 * i + 1], ka = k[j][2 * i], kb = k[j][2 * i + 1];
          q[j][2 * i] = qa * cc[i] - qb * ss[i];
          q[j][2 * i + 1] = qa * ss[i] + qb * cc[i];
          k[j][2 * i] = ka * cc[i] - kb * ss[i];
          k[j][2 * i + 1] = ka * ss[i] + kb * cc[i];
        }
      }
      const int64_t o = (((int64_t)b * H + h) * n_dst + n_off + n) * dh + d0;
      *(u32x4_t*)(qo + o) = pack8(q[j]);
      *(u32x4_t*)(ko + o) = pack8(k[j]);
      if (vo) *(u32x4_t*)(vo + o) = pack8(v[j]);
    }
    if (lane == 0) {
      rrms[row * 2] = rq;
      rrms[row * 2 + 1] = rk;
    }
  }
}

extern "C" int dl_qk_norm_rope_fwd_ex(const void* qkv, const float* scale_q, const float* scale_k, const float* cos,
                                      const float* sin, void* q, void* k, void* v, float* rrms, int64_t B, int64_t N,
                                      int64_t H, int64_t dh, int64_t rot, float eps, const int32_t* pos, int64_t n_dst,
                                      int64_t n_off, dl_stream_t stream);
extern "C" int dl_qk_norm_rope_fwd(const void* qkv, const float* scale_q, const float* scale_k, const float* cos,
                                   const float* sin, void* q, void* k, void* v, float* rrms, int64_t B, int64_t N,
                                   int64_t H, int64_t dh, int64_t rot, float eps, dl_stream_t stream) {
  return dl_qk_norm_rope_fwd_ex(qkv, scale_q, scale_k, cos, sin, q, k, v, rrms, B, N, H, dh, rot, eps, nullptr, N, 0, stream);
}
extern "C" int dl_qk_norm_rope_fwd_ex(const void* qkv, const float* scale_q, const float* scale_k, const float* cos,
                                      const float* sin, void* q, void* k, void* v, float* rrms, int64_t B, int64_t N,
                                      int64_t H, int64_t dh, int64_t rot, float eps, const int32_t* pos, int64_t n_dst,
                                      int64_t n_off, dl_stream_t stream) {
  DL_CHECK_ARG(qkv && scale_q && scale_k && cos && sin && q && k && rrms && B > 0 && N > 0,
               "dl_qk_norm_rope_fwd: null operand");
  DL_CHECK_ARG(n_off >= 0 && n_off + N <= n_dst, "dl_qk_norm_rope_fwd: row window [%lld, %lld) outside n_dst=%lld",
               (long long)n_off, (long long)(n_off + N), (long long)n_dst);
  const int64_t D = H * dh;
  DL_CHECK_ARG(dh % 8 == 0 && rot % 8 == 0 && rot <= dh && D <= 512 * MAXJ, "dl_qk_norm_rope_fwd: dh=%lld rot=%lld",
               (long long)dh, (long long)rot);
  const int64_t M = B * N;
  int grid = cdiv(M, 4);
  if (grid > 4096) grid = 4096;
  const int nj = cdiv(D, 512);
#define LAUNCH(NJ)                                                                                                  \
  hipLaunchKernelGGL(qk_norm_rope_fwd_k<NJ>, grid, 256, 0, (hipStream_t)stream, (const bf16_t*)qkv, scale_q, scale_k, \
                     cos, sin, (bf16_t*)q, (bf16_t*)k, (bf16_t*)v, rrms, M, (int)N, (int)H, (int)dh, (int)rot, eps, pos,      \
                     (int)n_dst, (int)n_off)
  if (nj == 1) LAUNCH(1);
  else LAUNCH(2);
#undef LAUNCH
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// backward: dq,dk,dv [B,H,N,dh] -> dqkv [B*N, 3D]; scale gradients accumulated per wave, reduced across the
// workgroup in LDS and added atomically to dscale[2][D].
template <int NJ>
__device__ __forceinline__ void qk_norm_rope_bwd_body(const bf16_t* __restrict__ dq, const bf16_t* __restrict__ dk,
                                                          const bf16_t* __restrict__ dv, const bf16_t* __restrict__ qkv,
                                                          const float* __restrict__ sq, const float* __restrict__ sk,
                                                          const float* __restrict__ cs, const float* __restrict__ sn,
                                                          const float* __restrict__ rrms, bf16_t* __restrict__ dqkv,
                                                          float* __restrict__ dscale, int64_t M, int N, int H, int dh,
                                                          int rot, const int* __restrict__ pos, int n_dst, int n_off) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = (float*)smem;  // [4 waves][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = H * dh, D8 = D >> 3;
  const float invD = 1.0f / (float)D;
  float wq[NJ][8], wk[NJ][8], aq[NJ][8], ak[NJ][8];
  load_row_f32<NJ>(sq, D8, lane, wq, 1.0f);
  load_row_f32<NJ>(sk, D8, lane, wk, 1.0f);
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) aq[j][e] = ak[j][e] = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
    const int64_t b = row / N;
    const int n = (int)(row - b * N);
    const int nt = pos ? pos[row] : n;
    const bf16_t* p = qkv + row * 3 * D;
    float xq[NJ][8], xk[NJ][8], gq[NJ][8], gk[NJ][8], gv[NJ][8];
    load_row<NJ>(p, D8, lane, xq);
    load_row<NJ>(p + D, D8, lane, xk);
    const float rq = rrms[row * 2], rk = rrms[row * 2 + 1];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      if (c >= D8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) gq[j][e] = gk[j][e] = gv[j][e] = 0.f;
        continue;
      }
      const int col = c * 8, h = col / dh, d0 = col - h * dh;
      const int64_t o = (((int64_t)b * H + h) * n_dst + n_off + n) * dh + d0;
      unpack8(*(const u32x4_t*)(dq + o), gq[j]);
      unpack8(*(const u32x4_t*)(dk + o), gk[j]);
      if (dv) unpack8(*(const u32x4_t*)(dv + o), gv[j]);
      if (d0 < rot) {  // transpose of the rotation
        const f32x4_t cc = *(const f32x4_t*)(cs + (int64_t)nt * (rot >> 1) + (d0 >> 1));
        const f32x4_t ss = *(const f32x4_t*)(sn + (int64_t)nt * (rot >> 1) + (d0 >> 1));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float qa = gq[j][2 * i], qb = gq[j][2 * i + 1], ka = gk[j][2 * i], kb = gk[j][2 * i + 1];
          gq[j][2 * i] = qa * cc[i] + qb * ss[i];
          gq[j][2 * i + 1] = -qa * ss[i] + qb * cc[i];
          gk[j][2 * i] = ka * cc[i] + kb * ss[i];
          gk[j][2 * i + 1] = -ka * ss[i] + kb * cc[i];
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        aq[j][e] += gq[j][e] * xq[j][e] * rq;  // dscale
        ak[j][e] += gk[j][e] * xk[j][e] * rk;
        gq[j][e] *= wq[j][e];                  // s * dy
        gk[j][e] *= wk[j][e];
        s1 += gq[j][e] * xq[j][e];
        s2 += gk[j][e] * xk[j][e];
      }
    }
    const float mq = wave_sum(s1) * invD * rq * rq * rq, mk = wave_sum(s2) * invD * rk * rk * rk;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        gq[j][e] = rq * gq[j][e] - xq[j][e] * mq;
        gk[j][e] = rk * gk[j][e] - xk[j][e] * mk;
      }
    bf16_t* o = dqkv + row * 3 * D;
    store_row<NJ>(o, D8, lane, gq);
    store_row<NJ>(o + D, D8, lane, gk);
    if (dv) store_row<NJ>(o + 2 * D, D8, lane, gv);  // dv == NULL: dl_attn_bwd_sv already wrote the v third of dqkv
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = lane + 64 * j;
    if (c < D8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(size_t)wave * 2 * D + c * 8 + e] = aq[j][e];
        red[(size_t)wave * 2 * D + D + c * 8 + e] = ak[j][e];
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += 256) {
    const float s = red[i] + red[2 * D + i] + red[4 * D + i] + red[6 * D + i];
    unsafeAtomicAdd(&dscale[i], s);
  }
}

#define QKB_ARGS                                                                                                                 \
  const bf16_t *__restrict__ dq, const bf16_t *__restrict__ dk, const bf16_t *__restrict__ dv, const bf16_t *__restrict__ qkv,        \
      const float *__restrict__ sq, const float *__restrict__ sk, const float *__restrict__ cs, const float *__restrict__ sn,         \
      const float *__restrict__ rrms, bf16_t *__restrict__ dqkv, float *__restrict__ dscale, int64_t M, int N, int H, int dh, int rot, \
      const int *__restrict__ pos, int n_dst, int n_off
#define QKB_PASS dq, dk, dv, qkv, sq, sk, cs, sn, rrms, dqkv, dscale, M, N, H, dh, rot, pos, n_dst, n_off
// D <= 512: four waves per SIMD (128 VGPRs, 5 of them spilled outside the row loop): 114 -> 96 us at B=256
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void qk_norm_rope_bwd_k(QKB_ARGS) {
  qk_norm_rope_bwd_body<1>(QKB_PASS);
}
__global__ __launch_bounds__(256) void qk_norm_rope_bwd_wide_k(QKB_ARGS) { qk_norm_rope_bwd_body<2>(QKB_PASS); }

extern "C" int dl_qk_norm_rope_bwd_ex(const void* dq, const void* dk, const void* dv, const void* qkv,
                                   const float* scale_q, const float* scale_k, const float* cos, const float* sin,
                                   const float* rrms, void* dqkv, float* dscale, int64_t B, int64_t N, int64_t H,
                                   int64_t dh, int64_t rot, const int32_t* pos, int64_t n_dst, int64_t n_off, dl_stream_t stream);
extern "C" int dl_qk_norm_rope_bwd(const void* dq, const void* dk, const void* dv, const void* qkv,
                                   const float* scale_q, const float* scale_k, const float* cos, const float* sin,
                                   const float* rrms, void* dqkv, float* dscale, int64_t B, int64_t N, int64_t H,
                                   int64_t dh, int64_t rot, dl_stream_t stream) {
  return dl_qk_norm_rope_bwd_ex(dq, dk, dv, qkv, scale_q, scale_k, cos, sin, rrms, dqkv, dscale, B, N, H, dh, rot, nullptr, N, 0, stream);
}
extern "C" int dl_qk_norm_rope_bwd_ex(const void* dq, const void* dk, const void* dv, const void* qkv,
                                   const float* scale_q, const float* scale_k, const float* cos, const float* sin,
                                   const float* rrms, void* dqkv, float* dscale, int64_t B, int64_t N, int64_t H,
                                   int64_t dh, int64_t rot, const int32_t* pos, int64_t n_dst, int64_t n_off, dl_stream_t stream) {
  DL_CHECK_ARG(dq && dk && qkv && scale_q && scale_k && cos && sin && rrms && dqkv && dscale && B > 0 && N > 0,
               "dl_qk_norm_rope_bwd: null operand");
  const int64_t D = H * dh;
  DL_CHECK_ARG(dh % 8 == 0 && rot % 8 == 0 && rot <= dh && D <= 512 * MAXJ && n_off >= 0 && n_off + N <= n_dst,
               "dl_qk_norm_rope_bwd: bad dims");
  const int64_t M = B * N;
  // >= 16 rows per wave so the atomics are amortised; a few thousand rows would leave half the chip idle that way: 4 rows per wave
  int grid = cdiv(M, M >= 32768 ? 4 * 16 : 4 * 4);
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  const int nj = cdiv(D, 512);
  const size_t lds = (size_t)4 * 2 * D * sizeof(float);
#define QKB_KERN_1 qk_norm_rope_bwd_k
#define QKB_KERN_2 qk_norm_rope_bwd_wide_k
#define LAUNCH(NJ)                                                                                                   \
  hipLaunchKernelGGL(QKB_KERN_##NJ, grid, 256, lds, (hipStream_t)stream, (const bf16_t*)dq, (const bf16_t*)dk, \
                     (const bf16_t*)dv, (const bf16_t*)qkv, scale_q, scale_k, cos, sin, rrms, (bf16_t*)dqkv, dscale, M, \
                     (int)N, (int)H, (int)dh, (int)rot, pos, (int)n_dst, (int)n_off)
  if (nj == 1) LAUNCH(1);
  else LAUNCH(2);
#undef LAUNCH
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ------------------------------------------------------------------------ QK-norm + RoPE backward, token-major and in place
// The q / k thirds of the dqkv rows arrive holding d(loss)/d(q, k after norm + RoPE) in token-major order (dl_attn_bwd_tok stores
// them there) and leave holding the gradient of the pre-norm q / k.  One wave per row, lane c owns columns [8c, 8c + 8) of the
// D-wide q and k rows; every global operand of a row (2 x 2 contiguous D-wide segments, two f32 scalars, the RoPE table entries)
// is fetched ONE ROW AHEAD of its use -- the head-major kernel above reads H 128-byte segments per row and tensor and waits for
// each row's loads before it starts (111 us at B = 256, 3.0 TB/s).  Scale-gradient partials: registers -> LDS -> one [2, D] slot per
// workgroup (plain stores); folded in a fixed order by fold_rows_k (no atomics).
template <int OCC>
__device__ __forceinline__ void qk_norm_rope_bwd_inplace_body(
    const bf16_t* __restrict__ qkv, const float* __restrict__ sq, const float* __restrict__ sk, const float* __restrict__ cs,
    const float* __restrict__ sn, const float* __restrict__ rrms, bf16_t* dqkv, float* __restrict__ partials, int64_t M, int N,
    int H, int dh, int rot, const int* __restrict__ pos) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red = (float*)smem;  // [4 waves][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = H * dh, D8 = D >> 3;
  const bool on = lane < D8;
  const float invD = 1.0f / (float)D;
  const int col = lane * 8, hh = on ? col / dh : 0, d0 = col - hh * dh;
  const bool rope = on && d0 < rot;
  float wq[8], wk[8], aq[8], ak[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    wq[e] = on ? sq[col + e] : 0.f;
    wk[e] = on ? sk[col + e] : 0.f;
    aq[e] = ak[e] = 0.f;
  }
  const int64_t stride = (int64_t)gridDim.x * 4;
  u32x4_t nxq = {0, 0, 0, 0}, nxk = nxq, ngq = nxq, ngk = nxq;
  f32x4_t ncc = {0.f, 0.f, 0.f, 0.f}, nss = ncc;
  float nrq = 0.f, nrk = 0.f;
  auto fetch = [&](int64_t r) {
    const bf16_t* p = qkv + r * 3 * D + col;
    const bf16_t* g = dqkv + r * 3 * D + col;
    if (on) {
      nxq = *(const u32x4_t*)p;
      nxk = *(const u32x4_t*)(p + D);
      ngq = *(const u32x4_t*)g;
      ngk = *(const u32x4_t*)(g + D);
    }
    nrq = rrms[r * 2];
    nrk = rrms[r * 2 + 1];
    if (rope) {
      const int nt = pos ? pos[r] : (int)(r % N);
      ncc = *(const f32x4_t*)(cs + (int64_t)nt * (rot >> 1) + (d0 >> 1));
      nss = *(const f32x4_t*)(sn + (int64_t)nt * (rot >> 1) + (d0 >> 1));
    }
  };
  int64_t row = (int64_t)blockIdx.x * 4 + wave;
  if (row < M) fetch(row);
  for (; row < M; row += stride) {
    float xq[8], xk[8], gq[8], gk[8];
    unpack8(nxq, xq);
    unpack8(nxk, xk);
    unpack8(ngq, gq);
    unpack8(ngk, gk);
    const f32x4_t cc = ncc, ss = nss;
    const float rq = nrq, rk = nrk;
    if (row + stride < M) fetch(row + stride);
    if (rope) {  // transpose of the rotation
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float qa = gq[2 * i], qb = gq[2 * i + 1], ka = gk[2 * i], kb = gk[2 * i + 1];
        gq[2 * i] = qa * cc[i] + qb * ss[i];
        gq[2 * i + 1] = -qa * ss[i] + qb * cc[i];
        gk[2 * i] = ka * cc[i] + kb * ss[i];
        gk[2 * i + 1] = -ka * ss[i] + kb * cc[i];
      }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      aq[e] += gq[e] * xq[e] * rq;  // dscale
      ak[e] += gk[e] * xk[e] * rk;
      gq[e] *= wq[e];               // s * dy
      gk[e] *= wk[e];
      s1 += gq[e] * xq[e];
      s2 += gk[e] * xk[e];
    }
    const float mq = wave_sum(s1) * invD * rq * rq * rq, mk = wave_sum(s2) * invD * rk * rk * rk;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      gq[e] = rq * gq[e] - xq[e] * mq;
      gk[e] = rk * gk[e] - xk[e] * mk;
    }
    if (on) {
      bf16_t* o = dqkv + row * 3 * D + col;
      *(u32x4_t*)o = pack8(gq);
      *(u32x4_t*)(o + D) = pack8(gk);
    }
  }
  if (on) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(size_t)wave * 2 * D + col + e] = aq[e];
      red[(size_t)wave * 2 * D + D + col + e] = ak[e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += 256)
    partials[(size_t)blockIdx.x * 2 * D + i] = (red[i] + red[2 * D + i]) + (red[4 * D + i] + red[6 * D + i]);
}
#define QKI_ARGS                                                                                                                  \
  const bf16_t *__restrict__ qkv, const float *__restrict__ sq, const float *__restrict__ sk, const float *__restrict__ cs,         \
      const float *__restrict__ sn, const float *__restrict__ rrms, bf16_t *dqkv, float *__restrict__ partials, int64_t M, int N, \
      int H, int dh, int rot, const int *__restrict__ pos
__global__ __launch_bounds__(256) void qk_norm_rope_bwd_inplace_k(QKI_ARGS) {
  qk_norm_rope_bwd_inplace_body<3>(qkv, sq, sk, cs, sn, rrms, dqkv, partials, M, N, H, dh, rot, pos);
}
// (capped at 128 VGPRs for four waves per SIMD it spills 84 bytes per lane inside the row loop: 165 us instead of 110)
// out[j] += sum_g partial[g * n + j] with ONE writer per element and a fixed order: 16 interleaved row lanes with four independent
// chains each (48 loads per thread at 768 partial rows, 12 deep), then a fixed tree.  (Four row lanes with one chain each -- 192
// dependent loads per thread on 12 workgroups -- took 49 us per block on the main chain: 0.6 ms of every DiT-S/2 step.)
#define FOLD_LANES 16
__global__ __launch_bounds__(64 * FOLD_LANES) void fold_rows_k(const float* __restrict__ partial, float* __restrict__ out, int G, int n) {
  __shared__ float red[FOLD_LANES][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < n) {
    const float* p = partial + c;
    int g = rl;
    for (; g + 3 * FOLD_LANES < G; g += 4 * FOLD_LANES) {
      a0 += p[(size_t)g * n];
      a1 += p[(size_t)(g + FOLD_LANES) * n];
      a2 += p[(size_t)(g + 2 * FOLD_LANES) * n];
      a3 += p[(size_t)(g + 3 * FOLD_LANES) * n];
    }
    for (; g < G; g += FOLD_LANES) a0 += p[(size_t)g * n];
  }
  red[rl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  float t = 0.f;  // fixed tree: 16 -> 4 -> 1
  if (rl < 4) t = (red[rl][cl] + red[rl + 4][cl]) + (red[rl + 8][cl] + red[rl + 12][cl]);
  __syncthreads();
  if (rl < 4) red[rl][cl] = t;
  __syncthreads();
  if (rl == 0 && c < n) out[c] += (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
int dl_fold_rows_launch(const float* partial, float* out, int G, int n, hipStream_t stream) {
  hipLaunchKernelGGL(fold_rows_k, cdiv(n, 64), 64 * FOLD_LANES, 0, stream, partial, out, G, n);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_qk_norm_rope_bwd_inplace(const void* qkv, const float* scale_q, const float* scale_k, const float* cos,
                                           const float* sin, const float* rrms, void* dqkv, float* dscale, float* dscale_partials,
                                           int64_t B, int64_t N, int64_t H, int64_t dh, int64_t rot, const int32_t* pos,
                                           dl_stream_t stream) {
  DL_CHECK_ARG(qkv && scale_q && scale_k && cos && sin && rrms && dqkv && dscale && dscale_partials && B > 0 && N > 0,
               "dl_qk_norm_rope_bwd_inplace: null operand");
  const int64_t D = H * dh;
  if (D > 512 || D % 8) {
    dl_set_error("dl_qk_norm_rope_bwd_inplace: inner width %lld (one wave owns a row of at most 512 columns)", (long long)D);
    return DL_ERR_UNSUPPORTED;
  }
  DL_CHECK_ARG(dh % 8 == 0 && rot % 8 == 0 && rot <= dh, "dl_qk_norm_rope_bwd_inplace: dh=%lld rot=%lld", (long long)dh, (long long)rot);
  DL_CHECK_ARG((((uintptr_t)qkv | (uintptr_t)dqkv | (uintptr_t)cos | (uintptr_t)sin) & 15) == 0,
               "dl_qk_norm_rope_bwd_inplace: 16-byte alignment");
  const int64_t M = B * N;
  // three workgroups per CU are resident (150 VGPRs): at most 768 workgroups, so that no second, partly filled round follows
  int grid = cdiv(M, M >= 32768 ? 4 * 16 : 4 * 4);
  if (grid > 768) grid = 768;
  const size_t lds = (size_t)4 * 2 * D * sizeof(float);
  hipLaunchKernelGGL(qk_norm_rope_bwd_inplace_k, grid, 256, lds, (hipStream_t)stream, (const bf16_t*)qkv, scale_q, scale_k, cos, sin,
                     rrms, (bf16_t*)dqkv, dscale_partials, M, (int)N, (int)H, (int)dh, (int)rot, pos);
  hipLaunchKernelGGL(fold_rows_k, cdiv(2 * D, 64), 64 * FOLD_LANES, 0, (hipStream_t)stream, dscale_partials, dscale, grid, (int)(2 * D));
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ======================================================================== SwiGLU
__global__ void swiglu_fwd_k(const bf16_t* __restrict__ u, bf16_t* __restrict__ h, int64_t M, int F8) {
  const int64_t total = M * F8, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t row = i / F8;
    const int c = (int)(i - row * F8);
    const bf16_t* p = u + row * (int64_t)F8 * 16 + c * 8;
    float a[8], b[8];
    unpack8(*(const u32x4_t*)p, a);
    unpack8(*(const u32x4_t*)(p + (int64_t)F8 * 8), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = silu_f(a[e]) * b[e];
    *(u32x4_t*)(h + row * (int64_t)F8 * 8 + c * 8) = pack8(a);
  }
}
__global__ void swiglu_bwd_k(const bf16_t* __restrict__ dh, const bf16_t* __restrict__ u, bf16_t* __restrict__ du,
                             int64_t M, int F8) {
  const int64_t total = M * F8, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t row = i / F8;
    const int c = (int)(i - row * F8);
    const int64_t uo = row * (int64_t)F8 * 16 + c * 8;
    float a[8], b[8], g[8], da[8], db[8];
    unpack8(*(const u32x4_t*)(u + uo), a);
    unpack8(*(const u32x4_t*)(u + uo + (int64_t)F8 * 8), b);
    unpack8(*(const u32x4_t*)(dh + row * (int64_t)F8 * 8 + c * 8), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      da[e] = g[e] * b[e] * dsilu_f(a[e]);
      db[e] = g[e] * silu_f(a[e]);
    }
    *(u32x4_t*)(du + uo) = pack8(da);
    *(u32x4_t*)(du + uo + (int64_t)F8 * 8) = pack8(db);
  }
}
extern "C" int dl_swiglu_fwd(const void* u, void* h, int64_t M, int64_t F, dl_stream_t stream) {
  DL_CHECK_ARG(u && h && M > 0 && F > 0 && F % 8 == 0, "dl_swiglu_fwd: bad args");
  int64_t g = (M * (F / 8) + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(swiglu_fwd_k, (int)g, 256, 0, (hipStream_t)stream, (const bf16_t*)u, (bf16_t*)h, M, (int)(F / 8));
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_swiglu_bwd(const void* dh, const void* u, void* du, int64_t M, int64_t F, dl_stream_t stream) {
  DL_CHECK_ARG(dh && u && du && M > 0 && F > 0 && F % 8 == 0, "dl_swiglu_bwd: bad args");
  int64_t g = (M * (F / 8) + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(swiglu_bwd_k, (int)g, 256, 0, (hipStream_t)stream, (const bf16_t*)dh, (const bf16_t*)u,
                     (bf16_t*)du, M, (int)(F / 8));
  DL_LAUNCH_CHECK();
  return DL_OK;
}
