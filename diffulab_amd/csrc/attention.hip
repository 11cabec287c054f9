// Multi-head self-attention for DiT token grids (gfx950): softmax(q k^T * scale) v, head_dim 64, no mask.
//
// Up to 256 tokens: one workgroup per (batch, head), the whole K and V of a head (N x 64 bf16 = 32 KiB each
// at N = 256) are DMA'd once into LDS (global_load_lds_dwordx4, bank swizzle applied on the source address)
// and stay resident; every wave owns a 32-row block.  All matmuls are MFMA 32x32x16 bf16 computed in the
// TRANSPOSED orientation (S^T = K Q^T, O^T = V^T P^T ...): the 32x32 accumulator then holds, per lane, one
// query column and 16 keys, so (a) softmax row statistics are in-register + one cross-half shuffle and
// (b) the exponentiated tile is ALREADY in the B-operand register layout of the next MFMA (k-slot j of lane
// half hi <-> key (j&3) + 8(j>>2) + 4hi), i.e. P never round-trips through LDS.  The matching A operands
// (V^T, K^T, Q^T, dO^T) come from the row-major LDS tiles through ds_read_b64_tr_b16 transposing reads.
//
// Backward (FlashAttention-2 style recompute from lse) runs two phases:
//   A: wave owns 32 queries, sweeps key blocks   -> dQ     (K, V resident in LDS; its q / dO fragments come from HBM)
//   B: wave owns 32 keys,    sweeps query blocks -> dK, dV (Q, dO resident in the SAME LDS; its k / v fragments from HBM)
// S and dP are recomputed in both phases (7 instead of 5 matmuls) which removes every cross-wave reduction.  Only two of
// the four tiles are resident at a time (64 KiB at N = 256) and a workgroup is N/64 waves, each owning two 32-row blocks:
// two workgroups share a CU, so one head's tile loads overlap the other's MFMA phases.
// Longer sequences (multiples of 256 up to 2048 tokens): `*_tiled_k` below run the same wave-level algorithms with one
// workgroup per (batch, head, 256-row chunk) that streams the other operand through LDS in 256-row chunks.
#include "common.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

#define DH 64
#define ROWB 128  // bytes per LDS tile row (64 bf16)
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// 16-byte-slot swizzle of a [rows][64 bf16] tile: conflict-free for ds_read_b128 by 16 consecutive rows
// (bijection of (row>>1)&7) AND for ds_read_b64_tr_b16 blocks of 4 consecutive rows (bit 2 separates rows r, r+2).
__device__ __forceinline__ int swz8(int row) {
  const int v = (row >> 1) & 7;
  return ((v & 1) << 2) | (v >> 1);
}
__device__ __forceinline__ int tile_off(int row, int slot) { return row * ROWB + ((slot ^ swz8(row)) << 4); }

// DMA rows [0, nrows) of a row-major [nrows][64] bf16 matrix (row pitch `pitch` elements) into an LDS tile.
__device__ __forceinline__ void tile_dma(const bf16_t* __restrict__ g, int64_t pitch, char* tile, int nrows, int wave,
                                         int nwaves, int lane) {
  const int nchunk = nrows >> 3;  // 8 rows = 1 KiB per wave-instruction
  for (int c = wave; c < nchunk; c += nwaves) {
    const int r = c * 8 + (lane >> 3);
    const int q = (lane & 7) ^ swz8(r);
    __builtin_amdgcn_global_load_lds((glb_void_t*)(g + (int64_t)r * pitch + q * 8), (lds_void_t*)(tile + c * 1024), 16, 0, 0);
  }
}

// A/B fragment with k = head-dim: row `row` of the tile, k-slots 16*ks + 8*hi .. +7
__device__ __forceinline__ bf16x8_t frag_rows(const char* tile, int row, int ks, int hi) {
  return *(const bf16x8_t*)(tile + tile_off(row, ks * 2 + hi));
}
// A fragment with k = tile rows (transposing read): output row i = head-dim column colbase + (lane&31),
// k-slot j of half hi <-> tile row rbase + (j&3) + 8*(j>>2) + 4*hi
__device__ __forceinline__ bf16x8_t frag_cols(const char* tile, int rbase, int colbase, int lane) {
  const int li = lane & 15, g = lane >> 4;
  const int col = colbase + (g & 1) * 16 + (li & 3) * 4;
  const int r0 = rbase + (g >> 1) * 4 + (li >> 2);
  union {
    s16x4_t h[2];
    bf16x8_t v;
  } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)(tile + r0 * ROWB + (((col >> 3) ^ swz8(r0)) << 4) + (col & 7) * 2));
  const int r1 = r0 + 8;
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)(tile + r1 * ROWB + (((col >> 3) ^ swz8(r1)) << 4) + (col & 7) * 2));
  return u.v;
}
__device__ __forceinline__ bf16x8_t pack_frag(const float* p) {
  union {
    u32x4_t u;
    bf16x8_t v;
  } r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.u[i] = pack2bf(p[2 * i], p[2 * i + 1]);
  return r.v;
}
// raw v_exp_f32: exp2f() wraps it in a denormal-range rescale (cmp / cndmask / add / ldexp, 6 VALU instructions per element, half of
// the backward kernel's VALU stream); a probability below 2^-126 flushing to zero is exactly what softmax wants
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// store a transposed 64-column accumulator pair (lane & 31 = row, register r <-> column 8*(r>>2) + 4*hi + (r&3) of each 32-wide
// half) as bf16: one v_permlane32_swap per register pair gives every lane 8 consecutive columns = one 16-byte store, 32 contiguous
// bytes per row and instruction (the GEMM epilogue's scheme) instead of 8-byte stores
__device__ __forceinline__ void store_rows64(bf16_t* rowp, const f32x16_t (&a)[2], float mul, int hi) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[dt][8 * gp + e] * mul), __float_as_uint(a[dt][8 * gp + 4 + e] * mul),
                                                   false, false);
        v[e] = __uint_as_float(sw[0]);
        v[4 + e] = __uint_as_float(sw[1]);
      }
      *(u32x4_t*)(rowp + dt * 32 + 16 * gp + 8 * hi) = pack8(v);
    }
}

// the same packing without the stores: pk[dt * 2 + gp] = the 16 bytes store_rows64 writes at column 32 dt + 16 gp + 8 hi
__device__ __forceinline__ void pack_rows64(u32x4_t (&pk)[4], const f32x16_t (&a)[2], float mul) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[dt][8 * gp + e] * mul), __float_as_uint(a[dt][8 * gp + 4 + e] * mul),
                                                   false, false);
        v[e] = __uint_as_float(sw[0]);
        v[4 + e] = __uint_as_float(sw[1]);
      }
      pk[dt * 2 + gp] = pack8(v);
    }
}

// where head (b, h) of V (or dV) lives: element offset b * bs + h * hs, rows `pitch` elements apart.  Head-major [B, H, N, 64] is
// {H*N*64, N*64, 64}; the v third of the token-major qkv rows [B*N, 3D] is {N*3D, 64, 3D} on a pointer advanced by 2D, which lets
// the N <= 256 kernels read V / write dV in place (no head-split copy of V in the QK-norm kernels).
struct HeadLayout {
  int64_t bs, hs;
  int pitch;
};

extern "C" int dl_attn_fwd_ex(const void* q, const void* k, const void* v, void* out, float* lse, int64_t B, int64_t H,
                              int64_t Nq, int64_t Nk, int64_t dh, float scale, const float* key_bias, dl_stream_t stream);
extern "C" int dl_attn_fwd_sv(const void* q, const void* k, const void* v, int64_t v_batch_stride, int64_t v_head_stride,
                              int64_t v_pitch, void* out, float* lse, int64_t B, int64_t H, int64_t N, int64_t dh, float scale,
                              dl_stream_t stream);
extern "C" int dl_attn_bwd_sv(const void* q, const void* k, const void* v, int64_t v_batch_stride, int64_t v_head_stride,
                              int64_t v_pitch, const void* out, const void* dout, const float* lse, void* dq, void* dk, void* dv,
                              int64_t dv_batch_stride, int64_t dv_head_stride, int64_t dv_pitch, int64_t B, int64_t H, int64_t N,
                              int64_t dh, float scale, dl_stream_t stream);
extern "C" int dl_attn_bwd_ex(const void* q, const void* k, const void* v, const void* out, const void* dout,
                              const float* lse, void* dq, void* dk, void* dv, int64_t B, int64_t H, int64_t Nq, int64_t Nk,
                              int64_t dh, float scale, const float* key_bias, dl_stream_t stream);

// ====================================================================================== forward
__global__ __launch_bounds__(512) void attn_fwd_k(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                  const bf16_t* __restrict__ v, bf16_t* __restrict__ out,
                                                  float* __restrict__ lse, int H, int N, float scale, HeadLayout vl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kt = smem;
  char* vt = smem + N * ROWB;
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const bf16_t* qg = q + (int64_t)bh * N * DH;
  const bf16_t* kg = k + (int64_t)bh * N * DH;
  const bf16_t* vg = v + b * vl.bs + h * vl.hs;
  tile_dma(kg, DH, kt, N, wave, nwaves, lane);
  tile_dma(vg, vl.pitch, vt, N, wave, nwaves, lane);

  const int q0 = wave * 32;
  bf16x8_t qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8_t*)(qg + (int64_t)(q0 + (lane & 31)) * DH + ks * 16 + hi * 8);

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const float c = scale * LOG2E;
  f32x16_t o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) o[0][r] = o[1][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int kb = 0; kb < N; kb += 64) {
    f32x16_t s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s[t] = MFMA(frag_rows(kt, kb + t * 32 + (lane & 31), ks, hi), qf[ks], s[t]);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[t][r] *= c;
        mx = fmaxf(mx, s[t][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = fast_exp2(m_run - m_new);
    m_run = m_new;
    float ps = 0.f;
    float p[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[t][r] = fast_exp2(s[t][r] - m_new);
        ps += p[t][r];
      }
    l_run = l_run * alpha + ps;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o[0][r] *= alpha;
      o[1][r] *= alpha;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kg2 = 0; kg2 < 2; ++kg2) {
        const bf16x8_t pf = pack_frag(&p[t][kg2 * 8]);
        const int rbase = kb + t * 32 + kg2 * 16;
        o[0] = MFMA(frag_cols(vt, rbase, 0, lane), pf, o[0]);
        o[1] = MFMA(frag_cols(vt, rbase, 32, lane), pf, o[1]);
      }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qrow = q0 + (lane & 31);
  store_rows64(out + ((int64_t)b * N + qrow) * (H * DH) + h * DH, o, inv, hi);
  if (hi == 0) lse[(int64_t)bh * N + qrow] = (m_run + log2f(l_tot)) * LN2;
}

// ------------------------------------------------------------------------------ forward, long sequences (N = 512 .. 2048)
// Same wave-level algorithm, but a workgroup owns a 256-row query chunk of one head and streams K / V through LDS in
// 256-key chunks (online softmax across chunks); grid = B * H * N/256.
#define ACH 256
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4))) void attn_fwd_tiled_k(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                        const bf16_t* __restrict__ v, bf16_t* __restrict__ out,
                                                        float* __restrict__ lse, int H, int Nq, int Nk, float scale,
                                                        const float* __restrict__ key_bias) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kt = smem;
  char* vt = smem + ACH * ROWB;
  float* bs = (float*)(vt + ACH * ROWB);  // additive key bias of the resident chunk, pre-multiplied by log2(e)
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
  const int nchq = Nq / ACH, nch = Nk / ACH;
  const int bh = blockIdx.x / nchq, qc = blockIdx.x - bh * nchq, b = bh / H, h = bh - b * H;
  const bf16_t* qg = q + (int64_t)bh * Nq * DH;
  const bf16_t* kg = k + (int64_t)bh * Nk * DH;
  const bf16_t* vg = v + (int64_t)bh * Nk * DH;
  const int q0 = qc * ACH + wave * 32;
  bf16x8_t qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const bf16x8_t*)(qg + (int64_t)(q0 + (lane & 31)) * DH + ks * 16 + hi * 8);
  const float c = scale * LOG2E;
  f32x16_t o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) o[0][r] = o[1][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  for (int kc = 0; kc < nch; ++kc) {
    __syncthreads();  // the previous chunk is consumed
    tile_dma(kg + (int64_t)kc * ACH * DH, DH, kt, ACH, wave, nwaves, lane);
    tile_dma(vg + (int64_t)kc * ACH * DH, DH, vt, ACH, wave, nwaves, lane);
    for (int i = threadIdx.x; i < ACH; i += blockDim.x)
      bs[i] = key_bias ? key_bias[(int64_t)b * Nk + kc * ACH + i] * LOG2E : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kb = 0; kb < ACH; kb += 64) {
      f32x16_t s[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s[t] = MFMA(frag_rows(kt, kb + t * 32 + (lane & 31), ks, hi), qf[ks], s[t]);
      }
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4_t b4 = *(const f32x4_t*)(bs + kb + t * 32 + g4 * 8 + hi * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = g4 * 4 + e;
            s[t][r] = s[t][r] * c + b4[e];
            mx = fmaxf(mx, s[t][r]);
          }
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float m_new = fmaxf(m_run, mx);
      if (m_new == -INFINITY) m_new = 0.f;  // every key so far is masked: keep exp2(-inf - m) = 0 well defined
      const float alpha = fast_exp2(m_run - m_new);
      m_run = m_new;
      float ps = 0.f;
      float p[2][16];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          p[t][r] = fast_exp2(s[t][r] - m_new);
          ps += p[t][r];
        }
      l_run = l_run * alpha + ps;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[0][r] *= alpha;
        o[1][r] *= alpha;
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kg2 = 0; kg2 < 2; ++kg2) {
          const bf16x8_t pf = pack_frag(&p[t][kg2 * 8]);
          const int rbase = kb + t * 32 + kg2 * 16;
          o[0] = MFMA(frag_cols(vt, rbase, 0, lane), pf, o[0]);
          o[1] = MFMA(frag_cols(vt, rbase, 32, lane), pf, o[1]);
        }
    }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qrow = q0 + (lane & 31);
  store_rows64(out + ((int64_t)b * Nq + qrow) * (H * DH) + h * DH, o, inv, hi);
  if (hi == 0) lse[(int64_t)bh * Nq + qrow] = (m_run + log2f(l_tot)) * LN2;
}

/* general form: Nq queries against Nk keys (both multiples of 256, <= 2048) with an optional additive key bias f32 [B, Nk]
 * (0 = keep, -inf = masked / padded key): cross-attention and key-padding masks */
extern "C" int dl_attn_fwd_ex(const void* q, const void* k, const void* v, void* out, float* lse, int64_t B, int64_t H,
                              int64_t Nq, int64_t Nk, int64_t dh, float scale, const float* key_bias, dl_stream_t stream) {
  DL_CHECK_ARG(q && k && v && out && lse && B > 0 && H > 0, "dl_attn_fwd_ex: null operand");
  DL_CHECK_ARG(dh == DH, "dl_attn_fwd_ex: head_dim %lld unsupported (64 only)", (long long)dh);
  DL_CHECK_ARG(Nq % ACH == 0 && Nk % ACH == 0 && Nq > 0 && Nk > 0 && Nq <= 2048 && Nk <= 2048,
               "dl_attn_fwd_ex: Nq=%lld Nk=%lld must be multiples of 256 up to 2048 (pad and mask)", (long long)Nq, (long long)Nk);
  const int ldt = 2 * ACH * ROWB + ACH * (int)sizeof(float);
  (void)hipFuncSetAttribute((const void*)attn_fwd_tiled_k, hipFuncAttributeMaxDynamicSharedMemorySize, ldt);
  hipLaunchKernelGGL(attn_fwd_tiled_k, (int)(B * H * (Nq / ACH)), 512, ldt, (hipStream_t)stream, (const bf16_t*)q,
                     (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, lse, (int)H, (int)Nq, (int)Nk, scale, key_bias);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int64_t B, int64_t H,
                           int64_t N, int64_t dh, float scale, dl_stream_t stream) {
  DL_CHECK_ARG(q && k && v && out && lse && B > 0 && H > 0, "dl_attn_fwd: null operand");
  DL_CHECK_ARG(dh == DH, "dl_attn_fwd: head_dim %lld unsupported (64 only)", (long long)dh);
  DL_CHECK_ARG(N % 64 == 0 && N >= 64 && (N <= 256 || (N % ACH == 0 && N <= 2048)),
               "dl_attn_fwd: N=%lld must be a multiple of 64 up to 256, or a multiple of 256 up to 2048", (long long)N);
  if (N > 256)  // (the resident-tile kernel runs N/32 waves per workgroup: 8 at most)
    return dl_attn_fwd_ex(q, k, v, out, lse, B, H, N, N, dh, scale, nullptr, stream);
  return dl_attn_fwd_sv(q, k, v, H * N * DH, N * DH, DH, out, lse, B, H, N, dh, scale, stream);
}
extern "C" int dl_attn_fwd_sv(const void* q, const void* k, const void* v, int64_t v_batch_stride, int64_t v_head_stride,
                              int64_t v_pitch, void* out, float* lse, int64_t B, int64_t H, int64_t N, int64_t dh, float scale,
                              dl_stream_t stream) {
  DL_CHECK_ARG(q && k && v && out && lse && B > 0 && H > 0, "dl_attn_fwd_sv: null operand");
  DL_CHECK_ARG(dh == DH, "dl_attn_fwd_sv: head_dim %lld unsupported (64 only)", (long long)dh);
  DL_CHECK_ARG(N % 64 == 0 && N >= 64 && N <= 256, "dl_attn_fwd_sv: N=%lld must be a multiple of 64 up to 256", (long long)N);
  DL_CHECK_ARG(v_pitch >= DH && v_pitch % 8 == 0 && v_head_stride % 8 == 0 && v_batch_stride % 8 == 0 && ((uintptr_t)v & 15) == 0,
               "dl_attn_fwd_sv: V rows must be 16-byte aligned");
  const int lds = (int)(2 * N * ROWB);
  (void)hipFuncSetAttribute((const void*)attn_fwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(attn_fwd_k, (int)(B * H), (int)(N / 32) * 64, lds, (hipStream_t)stream, (const bf16_t*)q,
                     (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, lse, (int)H, (int)N, scale,
                     HeadLayout{v_batch_stride, v_head_stride, (int)v_pitch});
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ------------------------------------------------------------------------------ forward with QK-RMSNorm + RoPE on load
// attn_fwd_k that reads the PRE-NORM q and k straight out of the token-major qkv rows (like V) and applies the QK-norm and the
// rotary embedding itself (mmdit.py:81-91: q = rms_norm(q) * scale over the full D-wide row, nn.py:427-431, then the rotation of
// interleaved pairs, nn.py:345-353): the separate qk_norm_rope_fwd pass over qkv (200 MB per launch at the headline shape) is gone.
// The row statistics come from the qkv GEMM's epilogue (dl_gemm_nt_ssq: sums of squares of the q / k rows), so a (batch, head)
// workgroup never has to see the other heads: r = rsqrt(ssq / D + eps).  K: the raw tile is DMA'd like V, then every thread
// transforms its four 16-byte chunks in place in LDS; Q: every lane transforms the four fragments of its query row in registers.
// The arithmetic is qk_norm_rope_fwd_k's ((x r) s, then a c - b s / a s + b c in f32, one rounding to bf16).  The normalised q, k
// are also written head-major (the backward kernels read them) and head 0's workgroups store r as rrms [M, 2] for the QK-norm
// backward.
struct QkNorm {
  const float* ssq;   // f32 [B*N, 2]: sums of squares of the q / k rows
  const float* sq;    // f32 [D] query_norm.scale
  const float* sk;    // f32 [D] key_norm.scale
  const float* cs;    // f32 [N, rot/2]
  const float* sn;
  bf16_t* qo;         // [B, H, N, 64] normalised + rotated q (kept for the backward)
  bf16_t* ko;
  float* rrms;        // f32 [B*N, 2]
  float inv_d, eps;
  int rot;
  int pitch;          // elements between token rows of qkv (3 D)
};
__device__ __forceinline__ u32x4_t qk_xform8(u32x4_t raw, float r, const float* __restrict__ s8, const float* __restrict__ c4,
                                             const float* __restrict__ n4, bool rotary) {
  float x[8];
  unpack8(raw, x);
  const f32x4_t s0 = *(const f32x4_t*)s8, s1 = *(const f32x4_t*)(s8 + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    x[e] = x[e] * r * s0[e];
    x[4 + e] = x[4 + e] * r * s1[e];
  }
  if (rotary) {
    const f32x4_t cc = *(const f32x4_t*)c4, ss = *(const f32x4_t*)n4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = x[2 * i], b = x[2 * i + 1];
      x[2 * i] = a * cc[i] - b * ss[i];
      x[2 * i + 1] = a * ss[i] + b * cc[i];
    }
  }
  return pack8(x);
}

__global__ __launch_bounds__(512) void attn_fwd_qkn_k(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ lse,
                                                      int H, int N, float scale, QkNorm qn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kt = smem;
  char* vt = smem + N * ROWB;
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int D = H * DH, half = qn.rot >> 1;
  const bf16_t* base = qkv + (int64_t)b * N * qn.pitch + h * DH;  // q of this head; k at + D, v at + 2 D
  tile_dma(base + D, qn.pitch, kt, N, wave, nwaves, lane);
  tile_dma(base + 2 * D, qn.pitch, vt, N, wave, nwaves, lane);

  // ---- everything the two transforms read from global memory is requested NOW, under the flight of the tile DMA: the raw q
  //      fragments of this wave's query rows, the row statistics, the rotary table rows and the scales of (up to) four K chunks per
  //      thread (chunk id = thread + i * blockDim: row id >> 3, 16-byte slot id & 7 -- the slot is the same for all of a thread's
  //      chunks because blockDim is a multiple of 8)
  const int q0 = wave * 32, qrow = q0 + (lane & 31);
  const int64_t tok = (int64_t)b * N + qrow;
  const float ssq_q = qn.ssq[tok * 2];
  u32x4_t qraw[4];
  f32x4_t qc[4], qs[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int d0 = ks * 16 + hi * 8;
    qraw[ks] = *(const u32x4_t*)(base + (int64_t)qrow * qn.pitch + d0);
    // the tables are [N, rot / 2]: only the rotated channels have an entry (rot < 64: the next row / past the end; rot == 0: NULL)
    if (d0 < qn.rot) {
      qc[ks] = *(const f32x4_t*)(qn.cs + (int64_t)qrow * half + (d0 >> 1));
      qs[ks] = *(const f32x4_t*)(qn.sn + (int64_t)qrow * half + (d0 >> 1));
    } else {
      qc[ks] = qs[ks] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
  }
  const int kslot = threadIdx.x & 7, kd0 = kslot * 8, krow0 = threadIdx.x >> 3, kstep = blockDim.x >> 3;
  float kssq[4];
  f32x4_t kc[4], kn[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = krow0 + i * kstep;
    if (row < N) {
      kssq[i] = qn.ssq[((int64_t)b * N + row) * 2 + 1];
      if (kd0 < qn.rot) {
        kc[i] = *(const f32x4_t*)(qn.cs + (int64_t)row * half + (kd0 >> 1));
        kn[i] = *(const f32x4_t*)(qn.sn + (int64_t)row * half + (kd0 >> 1));
      } else {
        kc[i] = kn[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
  // ---- this wave's query rows: normalised + rotated in registers
  const float rq = rsqrtf(ssq_q * qn.inv_d + qn.eps);
  bf16x8_t qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int d0 = ks * 16 + hi * 8;
    const u32x4_t t = qk_xform8(qraw[ks], rq, qn.sq + h * DH + d0, (const float*)&qc[ks], (const float*)&qs[ks], d0 < qn.rot);
    qf[ks] = __builtin_bit_cast(bf16x8_t, t);
    if (qn.qo) *(u32x4_t*)(qn.qo + ((int64_t)bh * N + qrow) * DH + d0) = t;  // (NULL in inference: only the backward reads them)
  }
  if (qn.rrms && h == 0 && hi == 0) qn.rrms[tok * 2] = rq;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- K tile: in place in LDS (source slot `slot` of row `row` sits at tile_off(row, slot))
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = krow0 + i * kstep;
    if (row < N) {
      const float rk = rsqrtf(kssq[i] * qn.inv_d + qn.eps);
      char* p = kt + tile_off(row, kslot);
      const u32x4_t t = qk_xform8(*(const u32x4_t*)p, rk, qn.sk + h * DH + kd0, (const float*)&kc[i], (const float*)&kn[i], kd0 < qn.rot);
      *(u32x4_t*)p = t;
      if (qn.ko) *(u32x4_t*)(qn.ko + ((int64_t)bh * N + row) * DH + kd0) = t;
      if (qn.rrms && h == 0 && kslot == 0) qn.rrms[((int64_t)b * N + row) * 2 + 1] = rk;
    }
  }
  __syncthreads();

  const float c = scale * LOG2E;
  f32x16_t o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) o[0][r] = o[1][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  for (int kb = 0; kb < N; kb += 64) {
    f32x16_t s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s[t] = MFMA(frag_rows(kt, kb + t * 32 + (lane & 31), ks, hi), qf[ks], s[t]);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[t][r] *= c;
        mx = fmaxf(mx, s[t][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = fast_exp2(m_run - m_new);
    m_run = m_new;
    float ps = 0.f;
    float p[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[t][r] = fast_exp2(s[t][r] - m_new);
        ps += p[t][r];
      }
    l_run = l_run * alpha + ps;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o[0][r] *= alpha;
      o[1][r] *= alpha;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kg2 = 0; kg2 < 2; ++kg2) {
        const bf16x8_t pf = pack_frag(&p[t][kg2 * 8]);
        const int rbase = kb + t * 32 + kg2 * 16;
        o[0] = MFMA(frag_cols(vt, rbase, 0, lane), pf, o[0]);
        o[1] = MFMA(frag_cols(vt, rbase, 32, lane), pf, o[1]);
      }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  store_rows64(out + ((int64_t)b * N + qrow) * (H * DH) + h * DH, o, inv, hi);
  if (hi == 0) lse[(int64_t)bh * N + qrow] = (m_run + log2f(l_tot)) * LN2;
}

// ------------------------------------------------------------------------------ the same forward, persistent + pipelined (round 6)
// attn_fwd_qkn_k is one dependency chain per (sample, head) workgroup -- tile DMA -> transform -> 4 key blocks -> stores -- with two
// workgroups per CU to cover each other: 95 us at the headline shape against 49 us for its 311 MB at the HBM rate.  Here ONE 8-wave
// workgroup per CU walks a contiguous run of (sample, head) items (at B = 256: the six heads of one sample, i.e. whole 2304-byte
// qkv rows in a burst) with the K / V tiles double-buffered in LDS (2 x 64 KiB): the DMA of item i + 1 and the raw q fragments /
// row statistics of item i + 1 are in flight under the MFMA / softmax work of item i.  The arithmetic (operation order included) is
// attn_fwd_qkn_k's: results are bit-identical (tests/test_row_gemm_gpu.py).
//   * the V fragments come out of ds_read_b64_tr_b16 as INLINE ASM with an explicit lgkmcnt wait: behind an in-flight direct-to-LDS
//     DMA the compiler puts s_waitcnt vmcnt(0) in front of a transposing read it can see (gemm.hip, lds_tr16), which would wait
//     for the prefetch before computing;
//   * an item's O rows are packed and stored ONE ITEM LATE, in front of the next prefetch: the vmcnt(0) at the top of an item then
//     only ever waits for loads requested a whole item ago (loads and stores share one in-order counter, and across the loop's
//     back edge the compiler waits with vmcnt(0) whatever is counted by hand);
//   * the QK-norm scales of all heads sit in LDS (their global loads would be on every item's critical chain);
//   * rotary table rows and the chunk geometry depend on the token row only: loaded once per workgroup.
static int g_attn_pipe = 1;  // 1: the persistent forms where they apply; 0: the chain forms everywhere
extern "C" __attribute__((visibility("default"))) void dl_lab_set_attn_pipe(int mode) { g_attn_pipe = mode; }  // LAB A/B switch (not in the header)
static int attn_fwd_pipe_on() { return g_attn_pipe & 1; }
__device__ __forceinline__ s16x4_t attn_lds_tr16(const char* p) {
  s16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((unsigned)(uintptr_t)(lds_void_t*)p));
  return v;
}
#define PIPE_N 256
__global__ __launch_bounds__(512) void attn_fwd_qkn_pipe_k(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ lse,
                                                           int H, int items, float scale, QkNorm qn) {
  constexpr int N = PIPE_N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, hi = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int per = (items + (int)gridDim.x - 1) / (int)gridDim.x;
  const int it0 = blockIdx.x * per, it1 = it0 + per < items ? it0 + per : items;
  if (it0 >= it1) return;
  const int D = H * DH, half = qn.rot >> 1;
  const int qrow = wave * 32 + l31;
  // ---- loop-invariant: rotary rows of this lane's token row, LDS offsets of the fragment reads.  A thread transforms the SAME
  //      (row, channel chunk) set of K that it holds of Q -- row qrow, chunks d0 = 16 ks + 8 hi -- so one set of table rows serves both
  f32x4_t qc[4], qs[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int d0 = ks * 16 + hi * 8;
    if (d0 < qn.rot) {
      qc[ks] = *(const f32x4_t*)(qn.cs + (int64_t)qrow * half + (d0 >> 1));
      qs[ks] = *(const f32x4_t*)(qn.sn + (int64_t)qrow * half + (d0 >> 1));
    } else {
      qc[ks] = qs[ks] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
  }
  int ro[4];  // frag_rows(tile, base + l31, ks, hi) = tile + base * ROWB + ro[ks]   (base % 32 == 0: the swizzle term is the lane's)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) ro[ks] = tile_off(l31, ks * 2 + hi);
  int co[2][2];  // frag_cols(tile, rbase, 32 cb, lane) halves = tile + rbase * ROWB + co[cb][0 / 1]   (rbase % 16 == 0)
  {
    const int li = lane & 15, g = lane >> 4;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int col = cb * 32 + (g & 1) * 16 + (li & 3) * 4, r = (g >> 1) * 4 + (li >> 2) + 8 * hh;
        co[cb][hh] = r * ROWB + (((col >> 3) ^ swz8(r)) << 4) + (col & 7) * 2;
      }
  }
  const float c = scale * LOG2E;
  // the QK-norm scales of every head in LDS behind the tile buffers (their per-item global loads would sit on the item's chain)
  float* sq_l = (float*)(smem + 4 * N * ROWB);
  float* sk_l = sq_l + D;
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    sq_l[i] = qn.sq[i];
    sk_l[i] = qn.sk[i];
  }

  u32x4_t qraw[4];
  float ssq_q, ssq_k;
  u32x4_t opk[4];       // the previous item's O rows, packed: stored one item late, BEFORE the next prefetch is requested, so that the
  float lse_pend = 0.f; // wait at the top of an item (vmcnt(0): tiles + registers of this item) never waits for a store just issued
  auto store_pending = [&](int item) {
    const int b = item / H, h = item - b * H;
    bf16_t* rowp = out + ((int64_t)b * N + qrow) * (H * DH) + h * DH;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) *(u32x4_t*)(rowp + dt * 32 + 16 * gp + 8 * hi) = opk[dt * 2 + gp];
    if (hi == 0) lse[(int64_t)item * N + qrow] = lse_pend;
  };
  auto issue_tiles = [&](int item, int buf) {
    const int b = item / H, h = item - b * H;
    const bf16_t* base = qkv + (int64_t)b * N * qn.pitch + h * DH;
    char* kt = smem + buf * (2 * N * ROWB);
    tile_dma(base + D, qn.pitch, kt, N, wave, 8, lane);
    tile_dma(base + 2 * D, qn.pitch, kt + N * ROWB, N, wave, 8, lane);
  };
  auto issue_regs = [&](int item) {
    const int b = item / H, h = item - b * H;
    const bf16_t* base = qkv + (int64_t)b * N * qn.pitch + h * DH;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qraw[ks] = *(const u32x4_t*)(base + (int64_t)qrow * qn.pitch + ks * 16 + hi * 8);
    const float2 sq2 = *(const float2*)(qn.ssq + ((int64_t)b * N + qrow) * 2);
    ssq_q = sq2.x;
    ssq_k = sq2.y;
  };
  issue_regs(it0);
  issue_tiles(it0, 0);

  for (int it = it0; it < it1; ++it) {
    const int buf = (it - it0) & 1;
    char* kt = smem + buf * (2 * N * ROWB);
    char* vt = kt + N * ROWB;
    const int b = it / H, h = it - b * H, bh = it;
    const int64_t tok = (int64_t)b * N + qrow;
    // the tiles and registers of THIS item were requested one item ago, before that item's main loop; nothing younger is in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // every wave's share of the tiles has landed; every wave is done with the other buffer (and, first item, sq_l / sk_l)
    // ---- this wave's query rows: normalised + rotated in registers
    const float rq = rsqrtf(ssq_q * qn.inv_d + qn.eps);
    bf16x8_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int d0 = ks * 16 + hi * 8;
      const u32x4_t t = qk_xform8(qraw[ks], rq, sq_l + h * DH + d0, (const float*)&qc[ks], (const float*)&qs[ks], d0 < qn.rot);
      qf[ks] = __builtin_bit_cast(bf16x8_t, t);
      if (qn.qo) *(u32x4_t*)(qn.qo + ((int64_t)bh * N + qrow) * DH + d0) = t;
    }
    if (qn.rrms && h == 0 && hi == 0) qn.rrms[tok * 2] = rq;
    // ---- K tile: in place in LDS (this thread's chunks: row qrow, the fragment slots of its q registers)
    const float rk = rsqrtf(ssq_k * qn.inv_d + qn.eps);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int d0 = ks * 16 + hi * 8;
      char* p = kt + wave * 32 * ROWB + ro[ks];
      const u32x4_t t = qk_xform8(*(const u32x4_t*)p, rk, sk_l + h * DH + d0, (const float*)&qc[ks], (const float*)&qs[ks], d0 < qn.rot);
      *(u32x4_t*)p = t;
      if (qn.ko) *(u32x4_t*)(qn.ko + ((int64_t)bh * N + qrow) * DH + d0) = t;
    }
    if (qn.rrms && h == 0 && hi == 0) qn.rrms[tok * 2 + 1] = rk;
    if (it > it0) store_pending(it - 1);
    // ---- prefetch the next item: raw q fragments + statistics into the registers just consumed, K / V into the other buffer
    if (it + 1 < it1) issue_regs(it + 1);
    __syncthreads();
    if (it + 1 < it1) issue_tiles(it + 1, buf ^ 1);

    f32x16_t o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) o[0][r] = o[1][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
#pragma unroll 1
    for (int kb = 0; kb < N; kb += 64) {
      // V fragments of this key block first (asm: no compiler-side wait), consumed after the softmax arithmetic
      union {
        s16x4_t h[2];
        bf16x8_t v;
      } vf[2][2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kg2 = 0; kg2 < 2; ++kg2)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            const char* pb = vt + (kb + t * 32 + kg2 * 16) * ROWB;
            vf[t][kg2][cb].h[0] = attn_lds_tr16(pb + co[cb][0]);
            vf[t][kg2][cb].h[1] = attn_lds_tr16(pb + co[cb][1]);
          }
      f32x16_t s[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s[t] = MFMA(*(const bf16x8_t*)(kt + (kb + t * 32) * ROWB + ro[ks]), qf[ks], s[t]);
      }
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[t][r] *= c;
          mx = fmaxf(mx, s[t][r]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = fast_exp2(m_run - m_new);
      m_run = m_new;
      float ps = 0.f;
      float p[2][16];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          p[t][r] = fast_exp2(s[t][r] - m_new);
          ps += p[t][r];
        }
      l_run = l_run * alpha + ps;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[0][r] *= alpha;
        o[1][r] *= alpha;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kg2 = 0; kg2 < 2; ++kg2)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) asm volatile("" : "+v"(vf[t][kg2][cb].v));  // (no MFMA on these registers above the wait)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kg2 = 0; kg2 < 2; ++kg2) {
          const bf16x8_t pf = pack_frag(&p[t][kg2 * 8]);
          o[0] = MFMA(vf[t][kg2][0].v, pf, o[0]);
          o[1] = MFMA(vf[t][kg2][1].v, pf, o[1]);
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    pack_rows64(opk, o, inv);
    lse_pend = (m_run + log2f(l_tot)) * LN2;
  }
  store_pending(it1 - 1);
}

/* DiTAttention.forward mmdit.py:81-100 from the pre-norm qkv rows: QK-RMSNorm (row statistics = ssq of dl_gemm_nt_ssq) + RoPE applied
 * on load, softmax(q k^T scale) v; also writes the normalised q, k head-major and rrms for the backward (q_out = k_out = rrms = NULL in
 * inference: 100 MB of stores per launch at the headline shape that only the backward reads).  N % 64 == 0 up to 256. */
extern "C" int dl_attn_fwd_qkn(const void* qkv, const float* ssq, const float* scale_q, const float* scale_k, const float* cos,
                               const float* sin, float eps, int64_t rot, void* q_out, void* k_out, float* rrms, void* out, float* lse,
                               int64_t B, int64_t H, int64_t N, int64_t dh, float scale, dl_stream_t stream) {
  DL_CHECK_ARG(qkv && ssq && scale_q && scale_k && out && lse && B > 0 && H > 0, "dl_attn_fwd_qkn: null operand");
  DL_CHECK_ARG((q_out && k_out && rrms) || (!q_out && !k_out && !rrms), "dl_attn_fwd_qkn: q_out, k_out and rrms are kept together (training) or not at all");
  DL_CHECK_ARG(rot == 0 || (cos && sin), "dl_attn_fwd_qkn: rot > 0 needs the cos / sin tables");
  DL_CHECK_ARG(dh == DH && rot % 8 == 0 && rot <= DH, "dl_attn_fwd_qkn: head_dim 64, rot %% 8 == 0 (dh=%lld rot=%lld)", (long long)dh,
               (long long)rot);
  DL_CHECK_ARG(N % 64 == 0 && N >= 64 && N <= 256, "dl_attn_fwd_qkn: N=%lld must be a multiple of 64 up to 256", (long long)N);
  DL_CHECK_ARG((((uintptr_t)qkv | (uintptr_t)q_out | (uintptr_t)k_out | (uintptr_t)out | (uintptr_t)scale_q | (uintptr_t)scale_k |
                 (uintptr_t)cos | (uintptr_t)sin) & 15) == 0, "dl_attn_fwd_qkn: 16-byte alignment");
  const int lds = (int)(2 * N * ROWB);
  static DevOnce once;
  (void)dev_cus(once, [] { (void)hipFuncSetAttribute((const void*)attn_fwd_qkn_k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * ROWB); });
  const int64_t D = H * DH;
  QkNorm qn{ssq, scale_q, scale_k, cos, sin, (bf16_t*)q_out, (bf16_t*)k_out, rrms, 1.0f / (float)D, eps, (int)rot, (int)(3 * D)};
  const int cus = dev_cus(once, [] {});
  if (N == PIPE_N && D <= 2048 && B * H >= 3 * (int64_t)cus && attn_fwd_pipe_on()) {
    // persistent + pipelined form: one workgroup per CU, >= 3 items each (below that the chain form's two workgroups per CU win)
    static DevOnce once2;
    (void)dev_cus(once2, [] { (void)hipFuncSetAttribute((const void*)attn_fwd_qkn_pipe_k, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * PIPE_N * ROWB + 8 * 2048); });
    const int items = (int)(B * H);
    const int per = (items + cus - 1) / cus, grid = (items + per - 1) / per;
    hipLaunchKernelGGL(attn_fwd_qkn_pipe_k, grid, 512, 4 * PIPE_N * ROWB + 8 * (int)D, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)out, lse, (int)H,
                       items, scale, qn);
    DL_LAUNCH_CHECK();
    return DL_OK;
  }
  hipLaunchKernelGGL(attn_fwd_qkn_k, (int)(B * H), (int)(N / 32) * 64, lds, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)out, lse, (int)H,
                     (int)N, scale, qn);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ====================================================================================== backward
// global-memory version of frag_rows: row-major [N][64] matrix with row pitch `pitch` elements
__device__ __forceinline__ bf16x8_t frag_rows_g(const bf16_t* __restrict__ g, int64_t pitch, int row, int ks, int hi) {
  return *(const bf16x8_t*)(g + (int64_t)row * pitch + (ks * 2 + hi) * 8);
}

__global__ __launch_bounds__(256, 2) void attn_bwd_k(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                     const bf16_t* __restrict__ v, const bf16_t* __restrict__ out,
                                                     const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                     bf16_t* __restrict__ dq, bf16_t* __restrict__ dk,
                                                     bf16_t* __restrict__ dv, int H, int N, float scale, HeadLayout vl,
                                                     HeadLayout dvl, HeadLayout dql) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ta = smem;             // phase A: K   | phase B: Q
  char* tb = ta + N * ROWB;    // phase A: V   | phase B: dO
  float* lse2 = (float*)(tb + N * ROWB);  // lse * log2(e)
  float* delta = lse2 + N;                // rowsum(dO * O)
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int64_t hoff = (int64_t)bh * N * DH;
  const int64_t tok_pitch = (int64_t)H * DH;
  const bf16_t* og = out + (int64_t)b * N * tok_pitch + h * DH;
  const bf16_t* dog = dout + (int64_t)b * N * tok_pitch + h * DH;
  tile_dma(k + hoff, DH, ta, N, wave, nwaves, lane);
  const bf16_t* vg = v + b * vl.bs + h * vl.hs;
  tile_dma(vg, vl.pitch, tb, N, wave, nwaves, lane);
  // (delta = rowsum(dO * O) and lse * log2(e) of a query block are computed by the wave that owns the block in phase A, from the
  // dO fragments it loads anyway plus the matching O fragments, and left in LDS for phase B: no separate pass over O and dO)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const float c = scale * LOG2E;

  // ------------------------------------------------------------------ phase A: dQ for the two query blocks of this wave
  for (int ob = 0; ob < 2; ++ob) {
    const int own = (wave * 2 + ob) * 32;
    bf16x8_t qf[4], dof[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = frag_rows_g(q + hoff, DH, own + (lane & 31), ks, hi);
      dof[ks] = frag_rows_g(dog, tok_pitch, own + (lane & 31), ks, hi);
    }
    float my_delta = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8_t of = frag_rows_g(og, tok_pitch, own + (lane & 31), ks, hi);
#pragma unroll
      for (int e = 0; e < 8; ++e) my_delta += (float)of[e] * (float)dof[ks][e];
    }
    my_delta += __shfl_xor(my_delta, 32, 64);  // the other 32 head-dim columns of the row live in the other lane half
    const float my_lse = lse[(int64_t)bh * N + own + (lane & 31)] * LOG2E;
    if (hi == 0) {
      delta[own + (lane & 31)] = my_delta;
      lse2[own + (lane & 31)] = my_lse;
    }
    f32x16_t dqa[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) dqa[0][r] = dqa[1][r] = 0.f;
    // software pipeline over the key blocks: the S / dP MFMAs of block kb + 32 are issued before the exp / dS arithmetic of block kb
    auto scores = [&](int kb, f32x16_t& st, f32x16_t& dpt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = dpt[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        st = MFMA(frag_rows(ta, kb + (lane & 31), ks, hi), qf[ks], st);
        dpt = MFMA(frag_rows(tb, kb + (lane & 31), ks, hi), dof[ks], dpt);
      }
    };
    auto step = [&](int kb, f32x16_t& st, f32x16_t& dpt, f32x16_t& stn, f32x16_t& dptn) {
      if (kb + 32 < N) scores(kb + 32, stn, dptn);
      float ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(st[r] * c - my_lse);
        ds[r] = p * (dpt[r] - my_delta);
      }
#pragma unroll
      for (int kg2 = 0; kg2 < 2; ++kg2) {
        const bf16x8_t df = pack_frag(&ds[kg2 * 8]);
        dqa[0] = MFMA(frag_cols(ta, kb + kg2 * 16, 0, lane), df, dqa[0]);
        dqa[1] = MFMA(frag_cols(ta, kb + kg2 * 16, 32, lane), df, dqa[1]);
      }
    };
    f32x16_t s0, d0, s1, d1;  // two tile pairs in flight, roles swapped by the 2x unroll (N is a multiple of 64)
    scores(0, s0, d0);
    for (int kb = 0; kb < N; kb += 64) {
      step(kb, s0, d0, s1, d1);
      step(kb + 32, s1, d1, s0, d0);
    }
    store_rows64(dq + b * dql.bs + h * dql.hs + (int64_t)(own + (lane & 31)) * dql.pitch, dqa, scale, hi);
  }

  // ------------------------------------------------------------------ swap the resident tiles: Q and dO replace K and V
  __syncthreads();  // every wave is done reading K / V
  tile_dma(q + hoff, DH, ta, N, wave, nwaves, lane);
  tile_dma(dog, tok_pitch, tb, N, wave, nwaves, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ------------------------------------------------------------------ phase B: dK, dV for the two key blocks of this wave
  for (int ob = 0; ob < 2; ++ob) {
    const int own = (wave * 2 + ob) * 32;
    bf16x8_t kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = frag_rows_g(k + hoff, DH, own + (lane & 31), ks, hi);
      vf[ks] = frag_rows_g(vg, vl.pitch, own + (lane & 31), ks, hi);
    }
    f32x16_t dka[2], dva[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) dka[0][r] = dka[1][r] = dva[0][r] = dva[1][r] = 0.f;
    // (software pipeline as in phase A, on S only: dK / dV / S / dP accumulators leave no room for a second dP tile)
    auto scores_t = [&](int qb, f32x16_t& s) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) s = MFMA(frag_rows(ta, qb + (lane & 31), ks, hi), kf[ks], s);
    };
    auto step_t = [&](int qb, f32x16_t& s, f32x16_t& sn) {
      f32x16_t dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) dp = MFMA(frag_rows(tb, qb + (lane & 31), ks, hi), vf[ks], dp);
      if (qb + 32 < N) scores_t(qb + 32, sn);
      float p[16], ds[16];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const f32x4_t l4 = *(const f32x4_t*)(lse2 + qb + g4 * 8 + hi * 4);
        const f32x4_t d4 = *(const f32x4_t*)(delta + qb + g4 * 8 + hi * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = g4 * 4 + e;
          p[r] = fast_exp2(s[r] * c - l4[e]);
          ds[r] = p[r] * (dp[r] - d4[e]);
        }
      }
#pragma unroll
      for (int kg2 = 0; kg2 < 2; ++kg2) {
        const bf16x8_t pf = pack_frag(&p[kg2 * 8]);
        const bf16x8_t df = pack_frag(&ds[kg2 * 8]);
        const int rbase = qb + kg2 * 16;
        dva[0] = MFMA(frag_cols(tb, rbase, 0, lane), pf, dva[0]);
        dva[1] = MFMA(frag_cols(tb, rbase, 32, lane), pf, dva[1]);
        dka[0] = MFMA(frag_cols(ta, rbase, 0, lane), df, dka[0]);
        dka[1] = MFMA(frag_cols(ta, rbase, 32, lane), df, dka[1]);
      }
    };
    f32x16_t s0, s1;  // (no 2x unroll here: it spills; the 16 register copies per block are cheaper)
    scores_t(0, s0);
    for (int qb = 0; qb < N; qb += 32) {
      step_t(qb, s0, s1);
      s0 = s1;
    }
    store_rows64(dk + b * dql.bs + h * dql.hs + (int64_t)(own + (lane & 31)) * dql.pitch, dka, scale, hi);
    store_rows64(dv + b * dvl.bs + h * dvl.hs + (int64_t)(own + (lane & 31)) * dvl.pitch, dva, 1.0f, hi);
  }
}

// ------------------------------------------------------------------------------ backward, long sequences (N = 512 .. 2048)
// A workgroup owns chunk c (256 rows) of one head: phase A computes dQ of its 256 queries streaming K / V chunks, phase B
// computes dK / dV of its 256 keys streaming Q / dO chunks; lse and delta of all N rows sit in LDS.  grid = B * H * N/256.
__global__ __launch_bounds__(256, 2) void attn_bwd_tiled_k(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                           const bf16_t* __restrict__ v, const bf16_t* __restrict__ out,
                                                           const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                           bf16_t* __restrict__ dq, bf16_t* __restrict__ dk,
                                                           bf16_t* __restrict__ dv, int H, int Nq, int Nk, float scale,
                                                           const float* __restrict__ key_bias) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ta = smem;
  char* tb = ta + ACH * ROWB;
  float* lse2 = (float*)(tb + ACH * ROWB);
  float* delta = lse2 + Nq;
  float* bs = delta + Nq;  // key bias of the resident K chunk (phase A), times log2(e)
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
  const int nchq = Nq / ACH, nchk = Nk / ACH, nmax = nchq > nchk ? nchq : nchk;
  const int bh = blockIdx.x / nmax, cc = blockIdx.x - bh * nmax, b = bh / H, h = bh - b * H;
  const int64_t hoffq = (int64_t)bh * Nq * DH, hoffk = (int64_t)bh * Nk * DH;
  const int64_t tok_pitch = (int64_t)H * DH;
  const bf16_t* og = out + (int64_t)b * Nq * tok_pitch + h * DH;
  const bf16_t* dog = dout + (int64_t)b * Nq * tok_pitch + h * DH;
  for (int row = threadIdx.x >> 1; row < Nq; row += blockDim.x >> 1) {
    const int half = threadIdx.x & 1;
    const bf16_t* po = og + (int64_t)row * tok_pitch + half * 32;
    const bf16_t* pd = dog + (int64_t)row * tok_pitch + half * 32;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a[8], d[8];
      unpack8(*(const u32x4_t*)(po + i * 8), a);
      unpack8(*(const u32x4_t*)(pd + i * 8), d);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += a[e] * d[e];
    }
    acc += __shfl_xor(acc, 1, 64);
    if (half == 0) {
      delta[row] = acc;
      lse2[row] = lse[(int64_t)bh * Nq + row] * LOG2E;
    }
  }
  const float c = scale * LOG2E;

  // ---- phase A: dQ of queries cc*256 + own..
  if (cc < nchq) {
    bf16x8_t qf[2][4], dof[2][4];
    float my_lse[2], my_delta[2];
    f32x16_t dqa[2][2];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
      const int own = cc * ACH + (wave * 2 + ob) * 32;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        qf[ob][ks] = frag_rows_g(q + hoffq, DH, own + (lane & 31), ks, hi);
        dof[ob][ks] = frag_rows_g(dog, tok_pitch, own + (lane & 31), ks, hi);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dqa[ob][0][r] = dqa[ob][1][r] = 0.f;
    }
    for (int kc = 0; kc < nchk; ++kc) {
      __syncthreads();
      tile_dma(k + hoffk + (int64_t)kc * ACH * DH, DH, ta, ACH, wave, nwaves, lane);
      tile_dma(v + hoffk + (int64_t)kc * ACH * DH, DH, tb, ACH, wave, nwaves, lane);
      for (int i = threadIdx.x; i < ACH; i += blockDim.x)
        bs[i] = key_bias ? key_bias[(int64_t)b * Nk + kc * ACH + i] * LOG2E : 0.f;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kc == 0) {  // lse2 / delta are complete after the first barrier pair
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
          const int own = cc * ACH + (wave * 2 + ob) * 32;
          my_lse[ob] = lse2[own + (lane & 31)];
          my_delta[ob] = delta[own + (lane & 31)];
        }
      }
#pragma unroll
      for (int ob = 0; ob < 2; ++ob)
        for (int kb = 0; kb < ACH; kb += 32) {
          f32x16_t st, dpt;
#pragma unroll
          for (int r = 0; r < 16; ++r) st[r] = dpt[r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            st = MFMA(frag_rows(ta, kb + (lane & 31), ks, hi), qf[ob][ks], st);
            dpt = MFMA(frag_rows(tb, kb + (lane & 31), ks, hi), dof[ob][ks], dpt);
          }
          float ds[16];
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4_t b4 = *(const f32x4_t*)(bs + kb + g4 * 8 + hi * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = g4 * 4 + e;
              const float p = fast_exp2(st[r] * c + b4[e] - my_lse[ob]);
              ds[r] = p * (dpt[r] - my_delta[ob]);
            }
          }
#pragma unroll
          for (int kg2 = 0; kg2 < 2; ++kg2) {
            const bf16x8_t df = pack_frag(&ds[kg2 * 8]);
            dqa[ob][0] = MFMA(frag_cols(ta, kb + kg2 * 16, 0, lane), df, dqa[ob][0]);
            dqa[ob][1] = MFMA(frag_cols(ta, kb + kg2 * 16, 32, lane), df, dqa[ob][1]);
          }
        }
    }
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
      const int own = cc * ACH + (wave * 2 + ob) * 32;
      store_rows64(dq + hoffq + (int64_t)(own + (lane & 31)) * DH, dqa[ob], scale, hi);
    }
  }

  // ---- phase B: dK, dV of keys cc*256 + own.., one own block at a time (accumulators: 64 regs each)
  for (int ob = 0; ob < 2 && cc < nchk; ++ob) {
    const int own = cc * ACH + (wave * 2 + ob) * 32;
    const float kbias = key_bias ? key_bias[(int64_t)b * Nk + own + (lane & 31)] * LOG2E : 0.f;
    bf16x8_t kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = frag_rows_g(k + hoffk, DH, own + (lane & 31), ks, hi);
      vf[ks] = frag_rows_g(v + hoffk, DH, own + (lane & 31), ks, hi);
    }
    f32x16_t dka[2], dva[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) dka[0][r] = dka[1][r] = dva[0][r] = dva[1][r] = 0.f;
    for (int qc = 0; qc < nchq; ++qc) {
      __syncthreads();
      tile_dma(q + hoffq + (int64_t)qc * ACH * DH, DH, ta, ACH, wave, nwaves, lane);
      tile_dma(dog + (int64_t)qc * ACH * tok_pitch, tok_pitch, tb, ACH, wave, nwaves, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int qb = 0; qb < ACH; qb += 32) {
        f32x16_t s2, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s2[r] = dp[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s2 = MFMA(frag_rows(ta, qb + (lane & 31), ks, hi), kf[ks], s2);
          dp = MFMA(frag_rows(tb, qb + (lane & 31), ks, hi), vf[ks], dp);
        }
        float p[16], ds[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4_t l4 = *(const f32x4_t*)(lse2 + qc * ACH + qb + g4 * 8 + hi * 4);
          const f32x4_t d4 = *(const f32x4_t*)(delta + qc * ACH + qb + g4 * 8 + hi * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = g4 * 4 + e;
            p[r] = fast_exp2(s2[r] * c + kbias - l4[e]);
            ds[r] = p[r] * (dp[r] - d4[e]);
          }
        }
#pragma unroll
        for (int kg2 = 0; kg2 < 2; ++kg2) {
          const bf16x8_t pf = pack_frag(&p[kg2 * 8]);
          const bf16x8_t df = pack_frag(&ds[kg2 * 8]);
          const int rbase = qb + kg2 * 16;
          dva[0] = MFMA(frag_cols(tb, rbase, 0, lane), pf, dva[0]);
          dva[1] = MFMA(frag_cols(tb, rbase, 32, lane), pf, dva[1]);
          dka[0] = MFMA(frag_cols(ta, rbase, 0, lane), df, dka[0]);
          dka[1] = MFMA(frag_cols(ta, rbase, 32, lane), df, dka[1]);
        }
      }
    }
    store_rows64(dk + hoffk + (int64_t)(own + (lane & 31)) * DH, dka, scale, hi);
    store_rows64(dv + hoffk + (int64_t)(own + (lane & 31)) * DH, dva, 1.0f, hi);
  }
}

extern "C" int dl_attn_bwd_ex(const void* q, const void* k, const void* v, const void* out, const void* dout,
                              const float* lse, void* dq, void* dk, void* dv, int64_t B, int64_t H, int64_t Nq, int64_t Nk,
                              int64_t dh, float scale, const float* key_bias, dl_stream_t stream) {
  DL_CHECK_ARG(q && k && v && out && dout && lse && dq && dk && dv && B > 0 && H > 0, "dl_attn_bwd_ex: null operand");
  DL_CHECK_ARG(dh == DH, "dl_attn_bwd_ex: head_dim %lld unsupported (64 only)", (long long)dh);
  DL_CHECK_ARG(Nq % ACH == 0 && Nk % ACH == 0 && Nq > 0 && Nk > 0 && Nq <= 2048 && Nk <= 2048,
               "dl_attn_bwd_ex: Nq=%lld Nk=%lld must be multiples of 256 up to 2048", (long long)Nq, (long long)Nk);
  const int ldt = (int)(2 * ACH * ROWB + (2 * Nq + ACH) * sizeof(float));
  const int64_t nmax = (Nq > Nk ? Nq : Nk) / ACH;
  (void)hipFuncSetAttribute((const void*)attn_bwd_tiled_k, hipFuncAttributeMaxDynamicSharedMemorySize, ldt);
  hipLaunchKernelGGL(attn_bwd_tiled_k, (int)(B * H * nmax), 256, ldt, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k,
                     (const bf16_t*)v, (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv,
                     (int)H, (int)Nq, (int)Nk, scale, key_bias);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                           const float* lse, void* dq, void* dk, void* dv, int64_t B, int64_t H, int64_t N,
                           int64_t dh, float scale, dl_stream_t stream) {
  DL_CHECK_ARG(q && k && v && out && dout && lse && dq && dk && dv && B > 0 && H > 0, "dl_attn_bwd: null operand");
  DL_CHECK_ARG(dh == DH, "dl_attn_bwd: head_dim %lld unsupported (64 only)", (long long)dh);
  DL_CHECK_ARG(N % 64 == 0 && N >= 64 && (N <= 256 || (N % ACH == 0 && N <= 2048)),
               "dl_attn_bwd: N=%lld must be a multiple of 64 up to 256, or a multiple of 256 up to 2048", (long long)N);
  if (N > 256) return dl_attn_bwd_ex(q, k, v, out, dout, lse, dq, dk, dv, B, H, N, N, dh, scale, nullptr, stream);
  return dl_attn_bwd_sv(q, k, v, H * N * DH, N * DH, DH, out, dout, lse, dq, dk, dv, H * N * DH, N * DH, DH, B, H, N, dh, scale,
                        stream);
}
static int attn_bwd_launch(const void* q, const void* k, const void* v, HeadLayout vl, const void* out, const void* dout,
                           const float* lse, void* dq, void* dk, HeadLayout dql, void* dv, HeadLayout dvl, int64_t B, int64_t H,
                           int64_t N, float scale, dl_stream_t stream) {
  const int lds = (int)(2 * N * ROWB + 2 * N * sizeof(float));
  (void)hipFuncSetAttribute((const void*)attn_bwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(attn_bwd_k, (int)(B * H), (int)(N / 64) * 64, lds, (hipStream_t)stream, (const bf16_t*)q,
                     (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)out, (const bf16_t*)dout, lse, (bf16_t*)dq,
                     (bf16_t*)dk, (bf16_t*)dv, (int)H, (int)N, scale, vl, dvl, dql);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_attn_bwd_sv(const void* q, const void* k, const void* v, int64_t v_batch_stride, int64_t v_head_stride,
                              int64_t v_pitch, const void* out, const void* dout, const float* lse, void* dq, void* dk, void* dv,
                              int64_t dv_batch_stride, int64_t dv_head_stride, int64_t dv_pitch, int64_t B, int64_t H, int64_t N,
                              int64_t dh, float scale, dl_stream_t stream) {
  DL_CHECK_ARG(q && k && v && out && dout && lse && dq && dk && dv && B > 0 && H > 0, "dl_attn_bwd_sv: null operand");
  DL_CHECK_ARG(dh == DH, "dl_attn_bwd_sv: head_dim %lld unsupported (64 only)", (long long)dh);
  DL_CHECK_ARG(N % 64 == 0 && N >= 64 && N <= 256, "dl_attn_bwd_sv: N=%lld must be a multiple of 64 up to 256", (long long)N);
  DL_CHECK_ARG(v_pitch >= DH && v_pitch % 8 == 0 && v_head_stride % 8 == 0 && v_batch_stride % 8 == 0 && ((uintptr_t)v & 15) == 0 &&
                   dv_pitch >= DH && dv_pitch % 8 == 0 && dv_head_stride % 8 == 0 && dv_batch_stride % 8 == 0 &&
                   ((uintptr_t)dv & 15) == 0,
               "dl_attn_bwd_sv: V and dV rows must be 16-byte aligned");
  return attn_bwd_launch(q, k, v, HeadLayout{v_batch_stride, v_head_stride, (int)v_pitch}, out, dout, lse, dq, dk,
                         HeadLayout{H * N * DH, N * DH, DH}, dv, HeadLayout{dv_batch_stride, dv_head_stride, (int)dv_pitch}, B, H, N,
                         scale, stream);
}
extern "C" int dl_attn_bwd_tok(const void* q, const void* k, const void* qkv, const void* out, const void* dout, const float* lse,
                               void* dqkv, int64_t B, int64_t H, int64_t N, int64_t dh, float scale, dl_stream_t stream) {
  DL_CHECK_ARG(q && k && qkv && out && dout && lse && dqkv && B > 0 && H > 0, "dl_attn_bwd_tok: null operand");
  DL_CHECK_ARG(dh == DH, "dl_attn_bwd_tok: head_dim %lld unsupported (64 only)", (long long)dh);
  DL_CHECK_ARG(N % 64 == 0 && N >= 64 && N <= 256, "dl_attn_bwd_tok: N=%lld must be a multiple of 64 up to 256", (long long)N);
  DL_CHECK_ARG((((uintptr_t)qkv | (uintptr_t)dqkv) & 15) == 0, "dl_attn_bwd_tok: qkv / dqkv rows must be 16-byte aligned");
  const int64_t D = H * DH;
  const HeadLayout tok{N * 3 * D, DH, (int)(3 * D)};  // head (b, h) of a third starts at b * N * 3D + h * 64, rows 3D apart
  return attn_bwd_launch(q, k, (const bf16_t*)qkv + 2 * D, tok, out, dout, lse, dqkv, (bf16_t*)dqkv + D, tok, (bf16_t*)dqkv + 2 * D, tok,
                         B, H, N, scale, stream);
}
// ====================================================================================== small attention (UNet AttentionBlock)
// n <= 64 tokens, head_dim a multiple of 64 up to 512 (unet.py:296-322: 8x8 and 4x4 feature maps, 256..512-wide heads): one
// workgroup (4 waves) per (batch, head), every matmul on MFMA 32x32x16 in the transposed orientation of the kernels above.
//   forward : S^T = K Q^T from global fragments (contraction over head_dim) -> LDS f32 -> row softmax (probabilities saved for the
//             backward, as before) -> O^T = V^T P^T with V staged in LDS as head_dim/64 swizzled [64][64] tiles
//   backward: dP^T = V dO^T from global fragments; dS = P o (dP - rowsum(P o dP)) * scale in LDS; then dV^T = dO^T P, dQ^T = K^T dS^T,
//             dK^T = Q^T dS with dO / K / Q staged one after the other in the same LDS tiles
// Rows >= n of a staged operand re-read row n-1 (finite) and meet probabilities that are exactly 0.
#define ASM_PITCH 68  // floats per row of the LDS score matrices (16-byte aligned rows, bank skew)
__device__ __forceinline__ void tile_dma_clamp(const bf16_t* __restrict__ g, int64_t pitch, char* tile, int nvalid, int wave,
                                               int nwaves, int lane) {
  for (int c = wave; c < 8; c += nwaves) {  // 64 rows
    int r = c * 8 + (lane >> 3);
    const int q = (lane & 7) ^ swz8(r);
    r = r < nvalid ? r : nvalid - 1;
    __builtin_amdgcn_global_load_lds((glb_void_t*)(g + (int64_t)r * pitch + q * 8), (lds_void_t*)(tile + c * 1024), 16, 0, 0);
  }
}
// B-operand fragment from a ROW of an LDS score matrix: lane & 31 = row, k-slot j of half hi <-> column cbase + (j&3) + 8(j>>2) + 4hi
__device__ __forceinline__ bf16x8_t score_frag_row(const float* S, int row, int cbase, int hi) {
  const float* p = S + row * ASM_PITCH + cbase + hi * 4;
  const f32x4_t a = *(const f32x4_t*)p, b = *(const f32x4_t*)(p + 8);
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return pack_frag(v);
}
// ... from a COLUMN: lane & 31 = column, k-slot j of half hi <-> row rbase + (j&3) + 8(j>>2) + 4hi
__device__ __forceinline__ bf16x8_t score_frag_col(const float* S, int col, int rbase, int hi) {
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = S[(rbase + (j & 3) + 8 * (j >> 2) + 4 * hi) * ASM_PITCH + col];
  return pack_frag(v);
}
// transposed 32 x 32 accumulator (lane & 31 = token row, register r <-> column 8(r>>2) + 4hi + (r&3)) -> bf16, 8 bytes per store
__device__ __forceinline__ void store_tile32(bf16_t* rowp, const f32x16_t& a, int hi) {
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    uint2 w;
    w.x = pack2bf(a[g4 * 4 + 0], a[g4 * 4 + 1]);
    w.y = pack2bf(a[g4 * 4 + 2], a[g4 * 4 + 3]);
    *(uint2*)(rowp + g4 * 8 + hi * 4) = w;
  }
}

__global__ __launch_bounds__(256) void attn_small_mfma_fwd_k(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                             const bf16_t* __restrict__ v, int64_t ldq, int64_t ldkv,
                                                             bf16_t* __restrict__ out, int64_t ldo, float* __restrict__ probs,
                                                             int n, int H, int dh, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* P = (float*)smem;
  char* vt = smem + 64 * ASM_PITCH * 4;
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const bf16_t* qb = q + (int64_t)b * n * ldq + h * dh;
  const bf16_t* kb = k + (int64_t)b * n * ldkv + h * dh;
  const bf16_t* vb = v + (int64_t)b * n * ldkv + h * dh;
  const int ncb = dh >> 6;
  for (int cb = 0; cb < ncb; ++cb) tile_dma_clamp(vb + cb * 64, ldkv, vt + cb * 8192, n, wave, 4, lane);
  {  // S^T tile (ti, tj) of this wave: lane & 31 = query, register r <-> key tj*32 + 8(r>>2) + 4hi + (r&3)
    const int ti = wave >> 1, tj = wave & 1;
    const int qi = ti * 32 + (lane & 31), kj = tj * 32 + (lane & 31);
    const bf16_t* qrow = qb + (int64_t)(qi < n ? qi : n - 1) * ldq + hi * 8;
    const bf16_t* krow = kb + (int64_t)(kj < n ? kj : n - 1) * ldkv + hi * 8;
    f32x16_t st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
    for (int ks = 0; ks < (dh >> 4); ++ks)
      st = MFMA(*(const bf16x8_t*)(krow + ks * 16), *(const bf16x8_t*)(qrow + ks * 16), st);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4_t w = {st[g4 * 4] * scale, st[g4 * 4 + 1] * scale, st[g4 * 4 + 2] * scale, st[g4 * 4 + 3] * scale};
      *(f32x4_t*)(P + qi * ASM_PITCH + tj * 32 + g4 * 8 + hi * 4) = w;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < 64) {  // row softmax; rows / columns beyond n become exact zeros
    const int i = threadIdx.x;
    float* pr = P + i * ASM_PITCH;
    if (i < n) {
      float m = -INFINITY;
      for (int j = 0; j < n; ++j) m = fmaxf(m, pr[j]);
      float l = 0.f;
      for (int j = 0; j < n; ++j) {
        const float e = __expf(pr[j] - m);
        pr[j] = e;
        l += e;
      }
      const float inv = 1.0f / l;
      float* gp = probs + (((int64_t)b * H + h) * n + i) * n;
      for (int j = 0; j < n; ++j) {
        pr[j] *= inv;
        gp[j] = pr[j];
      }
      for (int j = n; j < 64; ++j) pr[j] = 0.f;
    } else {
      for (int j = 0; j < 64; ++j) pr[j] = 0.f;
    }
  }
  __syncthreads();
  const int ntile = 2 * (dh >> 5);  // O^T tiles: (token half, 32-column block)
  for (int t = wave; t < ntile; t += 4) {
    const int th = t & 1, c32 = t >> 1;
    const int i = th * 32 + (lane & 31);
    f32x16_t o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg)
      o = MFMA(frag_cols(vt + (c32 >> 1) * 8192, kg * 16, (c32 & 1) * 32, lane), score_frag_row(P, i, kg * 16, hi), o);
    if (i < n) store_tile32(out + ((int64_t)b * n + i) * ldo + h * dh + c32 * 32, o, hi);
  }
}

__global__ __launch_bounds__(256) void attn_small_mfma_bwd_k(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                             const bf16_t* __restrict__ v, int64_t ldq, int64_t ldkv,
                                                             const bf16_t* __restrict__ dout, int64_t ldo,
                                                             const float* __restrict__ probs, bf16_t* __restrict__ dq,
                                                             bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int n, int H, int dh,
                                                             float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* P = (float*)smem;
  float* dS = P + 64 * ASM_PITCH;
  char* tt = smem + 2 * 64 * ASM_PITCH * 4;
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const bf16_t* qb = q + (int64_t)b * n * ldq + h * dh;
  const bf16_t* kb = k + (int64_t)b * n * ldkv + h * dh;
  const bf16_t* vb = v + (int64_t)b * n * ldkv + h * dh;
  const bf16_t* dob = dout + (int64_t)b * n * ldo + h * dh;
  const int ncb = dh >> 6, ntile = 2 * (dh >> 5);
  for (int cb = 0; cb < ncb; ++cb) tile_dma_clamp(dob + cb * 64, ldo, tt + cb * 8192, n, wave, 4, lane);
  const float* gp = probs + ((int64_t)b * H + h) * n * n;
  for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
    const int i = idx >> 6, j = idx & 63;
    P[i * ASM_PITCH + j] = (i < n && j < n) ? gp[i * n + j] : 0.f;
  }
  {  // dP^T tile of this wave = V dO^T: lane & 31 = query, registers <-> keys
    const int ti = wave >> 1, tj = wave & 1;
    const int qi = ti * 32 + (lane & 31), kj = tj * 32 + (lane & 31);
    const bf16_t* drow = dob + (int64_t)(qi < n ? qi : n - 1) * ldo + hi * 8;
    const bf16_t* vrow = vb + (int64_t)(kj < n ? kj : n - 1) * ldkv + hi * 8;
    f32x16_t dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) dp[r] = 0.f;
    for (int ks = 0; ks < (dh >> 4); ++ks)
      dp = MFMA(*(const bf16x8_t*)(vrow + ks * 16), *(const bf16x8_t*)(drow + ks * 16), dp);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4_t w = {dp[g4 * 4], dp[g4 * 4 + 1], dp[g4 * 4 + 2], dp[g4 * 4 + 3]};
      *(f32x4_t*)(dS + qi * ASM_PITCH + tj * 32 + g4 * 8 + hi * 4) = w;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < 64) {  // dS = P o (dP - rowsum(P o dP)) * scale; P is zero outside [n, n], hence so is dS
    const int i = threadIdx.x;
    float dot = 0.f;
    for (int j = 0; j < 64; ++j) dot += P[i * ASM_PITCH + j] * dS[i * ASM_PITCH + j];
    for (int j = 0; j < 64; ++j) dS[i * ASM_PITCH + j] = P[i * ASM_PITCH + j] * (dS[i * ASM_PITCH + j] - dot) * scale;
  }
  __syncthreads();
  // dV^T = dO^T P: contraction over queries, lane & 31 = key
  for (int t = wave; t < ntile; t += 4) {
    const int th = t & 1, c32 = t >> 1;
    const int j = th * 32 + (lane & 31);
    f32x16_t a;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg)
      a = MFMA(frag_cols(tt + (c32 >> 1) * 8192, kg * 16, (c32 & 1) * 32, lane), score_frag_col(P, j, kg * 16, hi), a);
    if (j < n) store_tile32(dv + ((int64_t)b * n + j) * ldkv + h * dh + c32 * 32, a, hi);
  }
  __syncthreads();
  for (int cb = 0; cb < ncb; ++cb) tile_dma_clamp(kb + cb * 64, ldkv, tt + cb * 8192, n, wave, 4, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // dQ^T = K^T dS^T: contraction over keys, lane & 31 = query
  for (int t = wave; t < ntile; t += 4) {
    const int th = t & 1, c32 = t >> 1;
    const int i = th * 32 + (lane & 31);
    f32x16_t a;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg)
      a = MFMA(frag_cols(tt + (c32 >> 1) * 8192, kg * 16, (c32 & 1) * 32, lane), score_frag_row(dS, i, kg * 16, hi), a);
    if (i < n) store_tile32(dq + ((int64_t)b * n + i) * ldq + h * dh + c32 * 32, a, hi);
  }
  __syncthreads();
  for (int cb = 0; cb < ncb; ++cb) tile_dma_clamp(qb + cb * 64, ldq, tt + cb * 8192, n, wave, 4, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // dK^T = Q^T dS: contraction over queries, lane & 31 = key
  for (int t = wave; t < ntile; t += 4) {
    const int th = t & 1, c32 = t >> 1;
    const int j = th * 32 + (lane & 31);
    f32x16_t a;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg)
      a = MFMA(frag_cols(tt + (c32 >> 1) * 8192, kg * 16, (c32 & 1) * 32, lane), score_frag_col(dS, j, kg * 16, hi), a);
    if (j < n) store_tile32(dk + ((int64_t)b * n + j) * ldkv + h * dh + c32 * 32, a, hi);
  }
}

// launchers for dl_attn_small_{fwd,bwd} (csrc/unet.hip); false: the shape is not theirs (head_dim % 64, > 512)
bool launch_attn_small_mfma_fwd(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldkv, void* out, int64_t ldo,
                                float* probs, int64_t B, int64_t n, int64_t H, int64_t dh, float scale, hipStream_t stream) {
  if (dh % 64 || dh > 512 || n > 64 || ldq % 8 || ldkv % 8 || ldo % 4) return false;
  const int lds = 64 * ASM_PITCH * 4 + (int)dh * 128;
  static DevOnce once;
  (void)dev_cus(once, [] {
    (void)hipFuncSetAttribute((const void*)attn_small_mfma_fwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * ASM_PITCH * 4 + 512 * 128);
  });
  hipLaunchKernelGGL(attn_small_mfma_fwd_k, (int)(B * H), 256, lds, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq,
                     ldkv, (bf16_t*)out, ldo, probs, (int)n, (int)H, (int)dh, scale);
  return true;
}
bool launch_attn_small_mfma_bwd(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldkv, const void* dout,
                                int64_t ldo, const float* probs, void* dq, void* dk, void* dv, int64_t B, int64_t n, int64_t H,
                                int64_t dh, float scale, hipStream_t stream) {
  if (dh % 64 || dh > 512 || n > 64 || ldq % 8 || ldkv % 8 || ldo % 8) return false;
  const int lds = 2 * 64 * ASM_PITCH * 4 + (int)dh * 128;
  static DevOnce once;
  (void)dev_cus(once, [] {
    (void)hipFuncSetAttribute((const void*)attn_small_mfma_bwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * ASM_PITCH * 4 + 512 * 128);
  });
  hipLaunchKernelGGL(attn_small_mfma_bwd_k, (int)(B * H), 256, lds, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq,
                     ldkv, (const bf16_t*)dout, ldo, probs, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, (int)n, (int)H, (int)dh, scale);
  return true;
}
