// fp8 (OCP e4m3) attention forward for the long joint text-image sequences (BASELINE config 5: 1152 tokens, head_dim 64):
// the CDNA4 block-scaled matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4 contracts 64 fp8 values per lane pair in one issue,
// i.e. a whole head dimension (S^T = K Q^T: ONE MFMA per 32-key x 32-query tile) or 64 keys (O^T = V^T P^T), at twice the bf16
// rate.  All block scales are 2^0; the per-(batch, head) tensor scales of q, k, v are applied in f32 around the MFMAs.
//
// Reference op: F.scaled_dot_product_attention on the joint sequence, mmdit.py:172-190 (MMDiTAttention.forward).
//
// Data flow
//   dl_probe_attn_fp8_quantize : q, k, v bf16 [B,H,N,64]  ->  q8, k8 fp8 [B,H,N,64] (64-byte rows), v8t fp8 [B,H,64,N] (V transposed,
//                          keys of every 64-key block stored in the ORDER THE P REGISTERS HOLD THEM, see below), scales f32 [B,H,3]
//                          = amax / 448 of each tensor per head (two launches: amax, then quantize)
//   dl_probe_attn_fwd_fp8      : one workgroup per (b, h, 256-query chunk), 8 waves x 32 queries; K / V^T chunks of 256 keys are DMA'd
//                          into LDS (16 KiB each); transposed orientation as in attention.hip: the 32x32 accumulator of S^T holds
//                          per lane one query column and 16 keys, so the softmax statistics are in-register and the exponentiated
//                          tile, converted to fp8, IS the B operand of the next MFMA: k-slot (16 t + r) of lane half hi <-> key
//                          32 t + (r & 3) + 8 (r >> 2) + 4 hi of the 64-key block.  V^T is stored with that key order, so the
//                          A operand is 32 contiguous bytes per lane.
// P is quantised as p * 2^8 (e4m3 keeps 3 mantissa bits down to 2^-6; p <= 1 would waste the top of the range); the row sum l is
// accumulated from the UNQUANTISED p in f32 and the 2^-8 is folded into the final normalisation.
// The backward stays bf16 (dl_attn_bwd_ex recomputes S from the bf16 q, k with the lse written here).
#include "../common.h"
#include "../../../include/diffulab_probe.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
typedef int v8i_t __attribute__((ext_vector_type(8)));

#define DH 64
#define FCH 256  // keys / queries per chunk
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#define P_SHIFT 8.0f
#define F8_MAX 448.0f

__device__ __forceinline__ f32x16_t mfma_f8(const v8i_t& a, const v8i_t& b, const f32x16_t& c) {
  // cbsz = 0 / blgp = 0: both operands OCP e4m3; scale exponents 127 = 2^0 in every byte
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}
// four floats -> four e4m3 bytes (a in byte 0): v_cvt_pk_fp8_f32 writes two bytes into the low / high half of its destination
__device__ __forceinline__ uint32_t cvt4(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (uint32_t)w;
}
__device__ __forceinline__ float fast_exp2f8(float x) {
  float r;
  asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// ------------------------------------------------------------------------------------------------ quantisation
// amax of q, k, v per (b, h): one workgroup per head and tensor
__global__ __launch_bounds__(256) void f8_amax_k(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                 float* __restrict__ scales, int Nq, int Nk) {
  __shared__ float red[4];
  const int bh = blockIdx.x / 3, which = blockIdx.x - bh * 3;
  const int n = (which == 0 ? Nq : Nk) * DH;
  const bf16_t* p = (which == 0 ? q : which == 1 ? k : v) + (int64_t)bh * n;
  float m = 0.f;
  for (int i = threadIdx.x * 8; i < n; i += 256 * 8) {
    float f[8];
    unpack8(*(const u32x4_t*)(p + i), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(f[e]));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    scales[bh * 3 + which] = m > 0.f ? m / F8_MAX : 1.0f;  // dequantisation scale: x ~= x8 * scale
  }
}
// q8 / k8: same [N,64] rows, 64 bytes each.  One thread converts 8 elements.
__global__ void f8_quant_rows_k(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, uint8_t* __restrict__ q8,
                                uint8_t* __restrict__ k8, const float* __restrict__ scales, int Nq, int Nk, int64_t nbh) {
  const int64_t per_q = (int64_t)Nq * DH / 8, per_k = (int64_t)Nk * DH / 8;
  const int64_t total = nbh * (per_q + per_k), stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const bool is_k = i >= nbh * per_q;
    const int64_t j = is_k ? i - nbh * per_q : i, per = is_k ? per_k : per_q;
    const int64_t bh = j / per;
    const float inv = 1.0f / scales[bh * 3 + (is_k ? 1 : 0)];
    float f[8];
    unpack8(*(const u32x4_t*)((is_k ? k : q) + j * 8), f);
    u32x2_t o;
    o[0] = cvt4(f[0] * inv, f[1] * inv, f[2] * inv, f[3] * inv);
    o[1] = cvt4(f[4] * inv, f[5] * inv, f[6] * inv, f[7] * inv);
    *(u32x2_t*)((is_k ? k8 : q8) + j * 8) = o;
  }
}
// v8t[bh][c][blk*64 + p] = V[bh][blk*64 + key(p)][c] / scale_v, key(p) = 32 t + (r & 3) + 8 (r >> 2) + 4 hi for p = 32 hi + 16 t + r.
// One workgroup per (bh, 64-key block): the block is staged through LDS so both sides are coalesced.
__global__ __launch_bounds__(256) void f8_quant_vt_k(const bf16_t* __restrict__ v, uint8_t* __restrict__ v8t,
                                                     const float* __restrict__ scales, int Nk) {
  __shared__ float tile[64][65];
  const int nblk = Nk / 64, bh = blockIdx.x / nblk, blk = blockIdx.x - bh * nblk;
  const float inv = 1.0f / scales[bh * 3 + 2];
  const bf16_t* src = v + ((int64_t)bh * Nk + blk * 64) * DH;
  for (int i = threadIdx.x; i < 64 * 8; i += 256) {  // 64 keys x 8 chunks of 8 columns
    const int key = i >> 3, c8 = (i & 7) * 8;
    float f[8];
    unpack8(*(const u32x4_t*)(src + key * DH + c8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[key][c8 + e] = f[e] * inv;
  }
  __syncthreads();
  // thread -> (column c, 16 positions): 64 columns x 4 groups of 16 positions
  const int c = threadIdx.x >> 2, pg = threadIdx.x & 3;  // positions 16 pg .. 16 pg + 15  <->  hi = pg >> 1, t = pg & 1
  const int hi = pg >> 1, t = pg & 1;
  u32x4_t o;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    float f[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = w * 4 + e;
      f[e] = tile[32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi][c];
    }
    o[w] = cvt4(f[0], f[1], f[2], f[3]);
  }
  *(u32x4_t*)(v8t + ((int64_t)bh * DH + c) * Nk + blk * 64 + pg * 16) = o;
}

extern "C" int dl_probe_attn_fp8_quantize(const void* q, const void* k, const void* v, void* q8, void* k8, void* v8t, float* scales,
                                    int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t dh, dl_stream_t stream) {
  DL_CHECK_ARG(q && k && v && q8 && k8 && v8t && scales && B > 0 && H > 0, "dl_probe_attn_fp8_quantize: null operand");
  DL_CHECK_ARG(dh == DH && Nq % 64 == 0 && Nk % 64 == 0 && Nq > 0 && Nk > 0, "dl_probe_attn_fp8_quantize: dh=64, N %% 64 == 0");
  DL_CHECK_ARG((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)q8 | (uintptr_t)k8 | (uintptr_t)v8t) & 15) == 0,
               "dl_probe_attn_fp8_quantize: 16-byte alignment");
  const int64_t nbh = B * H;
  hipLaunchKernelGGL(f8_amax_k, (int)(nbh * 3), 256, 0, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                     scales, (int)Nq, (int)Nk);
  int64_t g = (nbh * (Nq + Nk) * DH / 8 + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(f8_quant_rows_k, (int)g, 256, 0, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k, (uint8_t*)q8,
                     (uint8_t*)k8, scales, (int)Nq, (int)Nk, nbh);
  hipLaunchKernelGGL(f8_quant_vt_k, (int)(nbh * (Nk / 64)), 256, 0, (hipStream_t)stream, (const bf16_t*)v, (uint8_t*)v8t, scales,
                     (int)Nk);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ------------------------------------------------------------------------------------------------ forward
// LDS images (both DMA'd with 16-byte lanes, swizzle on the source address, rule 21):
//   K chunk  [256 keys][64 B]   : 16-byte slot' = slot ^ ((key >> 2) & 3)   (16 consecutive keys -> 16 distinct bank quads)
//   V^T chunk [64 dh][256 B]    : 16-byte slot' = slot ^ (row & 15)
__global__ __launch_bounds__(512) void attn_fwd_fp8_k(const uint8_t* __restrict__ q8, const uint8_t* __restrict__ k8,
                                                      const uint8_t* __restrict__ v8t, const float* __restrict__ scales,
                                                      bf16_t* __restrict__ out, float* __restrict__ lse, int H, int Nq, int Nk,
                                                      float scale, const float* __restrict__ key_bias) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kt = smem;                      // 16 KiB
  char* vt = smem + FCH * DH;           // 16 KiB
  float* bs = (float*)(vt + DH * FCH);  // key bias of the chunk (x log2 e)
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nchq = Nq / FCH, nch = Nk / FCH;
  const int bh = blockIdx.x / nchq, qc = blockIdx.x - bh * nchq, b = bh / H, h = bh - b * H;
  const float sq = scales[bh * 3], sk = scales[bh * 3 + 1], sv = scales[bh * 3 + 2];
  const int q0 = qc * FCH + wave * 32;
  const v8i_t qf = *(const v8i_t*)(q8 + ((int64_t)bh * Nq + q0 + (lane & 31)) * DH + hi * 32);
  const float c = scale * LOG2E * sq * sk;  // raw fp8 dot product -> scaled scores in the exp2 domain
  f32x16_t o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) o[0][r] = o[1][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const uint8_t* kg = k8 + (int64_t)bh * Nk * DH;
  const uint8_t* vg = v8t + (int64_t)bh * DH * Nk;
  for (int kc = 0; kc < nch; ++kc) {
    __syncthreads();  // the previous chunk is consumed
    // K chunk: 16 wave-instructions of 1 KiB = 16 keys each; lane -> key 16 i + (lane >> 2), slot lane & 3
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ch = wave * 2 + i, key = ch * 16 + (lane >> 2), slot = (lane & 3) ^ ((key >> 2) & 3);
      __builtin_amdgcn_global_load_lds((glb_void_t*)(kg + ((int64_t)kc * FCH + key) * DH + slot * 16), (lds_void_t*)(kt + ch * 1024), 16, 0, 0);
    }
    // V^T chunk: row c = dh column, 256 bytes of this chunk's keys; 4 rows per wave-instruction
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ch = wave * 2 + i, row = ch * 4 + (lane >> 4), slot = (lane & 15) ^ (row & 15);
      __builtin_amdgcn_global_load_lds((glb_void_t*)(vg + (int64_t)row * Nk + kc * FCH + slot * 16), (lds_void_t*)(vt + ch * 1024), 16, 0, 0);
    }
    for (int i = threadIdx.x; i < FCH; i += blockDim.x) bs[i] = key_bias ? key_bias[(int64_t)b * Nk + kc * FCH + i] * LOG2E : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kb = 0; kb < FCH; kb += 64) {
      f32x16_t s[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int key = kb + t * 32 + (lane & 31);
        const char* row = kt + key * DH;
        const int sw = (key >> 2) & 3;
        const u32x4_t k0 = *(const u32x4_t*)(row + (((2 * hi) ^ sw) << 4));
        const u32x4_t k1 = *(const u32x4_t*)(row + (((2 * hi + 1) ^ sw) << 4));
        const v8i_t kf = {(int)k0[0], (int)k0[1], (int)k0[2], (int)k0[3], (int)k1[0], (int)k1[1], (int)k1[2], (int)k1[3]};
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
        s[t] = mfma_f8(kf, qf, s[t]);
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
      if (m_new == -INFINITY) m_new = 0.f;  // every key so far is masked
      const float alpha = fast_exp2f8(m_run - m_new);
      m_run = m_new;
      float ps = 0.f;
      v8i_t pf;  // k-slot 16 t + r of this lane half
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          float p4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            p4[e] = fast_exp2f8(s[t][w * 4 + e] - m_new);
            ps += p4[e];
          }
          pf[t * 4 + w] = (int)cvt4(p4[0] * 256.0f, p4[1] * 256.0f, p4[2] * 256.0f, p4[3] * 256.0f);
        }
      l_run = l_run * alpha + ps;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[0][r] *= alpha;
        o[1][r] *= alpha;
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {  // dh columns 0-31 / 32-63
        const int row = half * 32 + (lane & 31);
        const char* rp = vt + row * FCH;
        const int s0 = (kb >> 4) + 2 * hi;  // 16-byte slot of key position 32 hi inside this 64-key block
        const u32x4_t v0 = *(const u32x4_t*)(rp + ((s0 ^ (row & 15)) << 4));
        const u32x4_t v1 = *(const u32x4_t*)(rp + (((s0 + 1) ^ (row & 15)) << 4));
        const v8i_t vf = {(int)v0[0], (int)v0[1], (int)v0[2], (int)v0[3], (int)v1[0], (int)v1[1], (int)v1[2], (int)v1[3]};
        o[half] = mfma_f8(vf, pf, o[half]);
      }
    }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = sv / (l_tot * 256.0f);
  const int qrow = q0 + (lane & 31);
  // O^T accumulator: lane = query column, registers = dh rows (r & 3) + 8 (r >> 2) + 4 hi (+ 32 for o[1])
  bf16_t* op = out + ((int64_t)b * Nq + qrow) * (H * DH) + h * DH;
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      u32x2_t w2;
      w2[0] = pack2bf(o[half][g4 * 4 + 0] * inv, o[half][g4 * 4 + 1] * inv);
      w2[1] = pack2bf(o[half][g4 * 4 + 2] * inv, o[half][g4 * 4 + 3] * inv);
      *(u32x2_t*)(op + half * 32 + g4 * 8 + hi * 4) = w2;
    }
  if (hi == 0) lse[(int64_t)bh * Nq + qrow] = (m_run + log2f(l_tot)) * LN2;
}

extern "C" int dl_probe_attn_fwd_fp8(const void* q8, const void* k8, const void* v8t, const float* scales, void* out, float* lse,
                               int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t dh, float scale, const float* key_bias,
                               dl_stream_t stream) {
  DL_CHECK_ARG(q8 && k8 && v8t && scales && out && lse && B > 0 && H > 0, "dl_probe_attn_fwd_fp8: null operand");
  DL_CHECK_ARG(dh == DH, "dl_probe_attn_fwd_fp8: head_dim %lld unsupported (64 only)", (long long)dh);
  DL_CHECK_ARG(Nq % FCH == 0 && Nk % FCH == 0 && Nq > 0 && Nk > 0 && Nq <= 4096 && Nk <= 4096,
               "dl_probe_attn_fwd_fp8: Nq=%lld Nk=%lld must be multiples of 256 up to 4096 (pad and mask)", (long long)Nq, (long long)Nk);
  const int lds = 2 * FCH * DH + FCH * (int)sizeof(float);
  hipLaunchKernelGGL(attn_fwd_fp8_k, (int)(B * H * (Nq / FCH)), 512, lds, (hipStream_t)stream, (const uint8_t*)q8,
                     (const uint8_t*)k8, (const uint8_t*)v8t, scales, (bf16_t*)out, lse, (int)H, (int)Nq, (int)Nk, scale, key_bias);
  DL_LAUNCH_CHECK();
  return DL_OK;
}


// ================================================================================================ round 6: QK^T ONLY in fp8
// "QK^T in MX-scaled e4m3, P.V stays bf16" -- the narrower variant: S^T = K Q^T on v_mfma_scale_f32_32x32x64_f8f6f4
// (one issue per 32 x 32 tile instead of four bf16 ones), the probabilities stay f32 -> bf16 and meet a bf16 V tile exactly as in
// attention.hip::attn_fwd_tiled_k (transposing LDS reads, 32x32x16 bf16 MFMAs).  K chunk: fp8 [256][64 B] as above; V chunk: bf16
// [256][128 B] with attention.hip's 16-byte-slot swizzle.
#define ROWB 128
__device__ __forceinline__ int q_swz8(int row) {
  const int v = (row >> 1) & 7;
  return ((v & 1) << 2) | (v >> 1);
}
__device__ __forceinline__ bf16x8_t q_frag_cols(const char* tile, int rbase, int colbase, int lane) {
  const int li = lane & 15, g = lane >> 4;
  const int col = colbase + (g & 1) * 16 + (li & 3) * 4;
  const int r0 = rbase + (g >> 1) * 4 + (li >> 2);
  union {
    s16x4_t h[2];
    bf16x8_t v;
  } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)(tile + r0 * ROWB + (((col >> 3) ^ q_swz8(r0)) << 4) + (col & 7) * 2));
  const int r1 = r0 + 8;
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4_t*)(tile + r1 * ROWB + (((col >> 3) ^ q_swz8(r1)) << 4) + (col & 7) * 2));
  return u.v;
}
__global__ __launch_bounds__(512) void attn_fwd_fp8qk_k(const uint8_t* __restrict__ q8, const uint8_t* __restrict__ k8,
                                                        const bf16_t* __restrict__ v, const float* __restrict__ scales,
                                                        bf16_t* __restrict__ out, float* __restrict__ lse, int H, int Nq, int Nk,
                                                        float scale, const float* __restrict__ key_bias) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kt = smem;                        // 16 KiB fp8 K chunk
  char* vt = smem + FCH * DH;             // 32 KiB bf16 V chunk
  float* bs = (float*)(vt + FCH * ROWB);  // key bias of the chunk (x log2 e)
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nchq = Nq / FCH, nch = Nk / FCH;
  const int bh = blockIdx.x / nchq, qc = blockIdx.x - bh * nchq, b = bh / H, h = bh - b * H;
  const float sq = scales[bh * 3], sk = scales[bh * 3 + 1];
  const int q0 = qc * FCH + wave * 32;
  const v8i_t qf = *(const v8i_t*)(q8 + ((int64_t)bh * Nq + q0 + (lane & 31)) * DH + hi * 32);
  const float c = scale * LOG2E * sq * sk;
  f32x16_t o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) o[0][r] = o[1][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const uint8_t* kg = k8 + (int64_t)bh * Nk * DH;
  const bf16_t* vg = v + (int64_t)bh * Nk * DH;
  for (int kc = 0; kc < nch; ++kc) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ch = wave * 2 + i, key = ch * 16 + (lane >> 2), slot = (lane & 3) ^ ((key >> 2) & 3);
      __builtin_amdgcn_global_load_lds((glb_void_t*)(kg + ((int64_t)kc * FCH + key) * DH + slot * 16), (lds_void_t*)(kt + ch * 1024), 16, 0, 0);
    }
    for (int cch = wave; cch < FCH / 8; cch += 8) {  // V: 8 rows = 1 KiB per wave-instruction
      const int r = cch * 8 + (lane >> 3), qs = (lane & 7) ^ q_swz8(r);
      __builtin_amdgcn_global_load_lds((glb_void_t*)(vg + ((int64_t)kc * FCH + r) * DH + qs * 8), (lds_void_t*)(vt + cch * 1024), 16, 0, 0);
    }
    for (int i = threadIdx.x; i < FCH; i += blockDim.x) bs[i] = key_bias ? key_bias[(int64_t)b * Nk + kc * FCH + i] * LOG2E : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kb = 0; kb < FCH; kb += 64) {
      f32x16_t s[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int key = kb + t * 32 + (lane & 31);
        const char* row = kt + key * DH;
        const int sw = (key >> 2) & 3;
        const u32x4_t k0 = *(const u32x4_t*)(row + (((2 * hi) ^ sw) << 4));
        const u32x4_t k1 = *(const u32x4_t*)(row + (((2 * hi + 1) ^ sw) << 4));
        const v8i_t kf = {(int)k0[0], (int)k0[1], (int)k0[2], (int)k0[3], (int)k1[0], (int)k1[1], (int)k1[2], (int)k1[3]};
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
        s[t] = mfma_f8(kf, qf, s[t]);
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
      if (m_new == -INFINITY) m_new = 0.f;
      const float alpha = fast_exp2f8(m_run - m_new);
      m_run = m_new;
      float ps = 0.f;
      float p[2][16];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          p[t][r] = fast_exp2f8(s[t][r] - m_new);
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
          union {
            u32x4_t u;
            bf16x8_t v;
          } pf;
#pragma unroll
          for (int i = 0; i < 4; ++i) pf.u[i] = pack2bf(p[t][kg2 * 8 + 2 * i], p[t][kg2 * 8 + 2 * i + 1]);
          const int rbase = kb + t * 32 + kg2 * 16;
          o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q_frag_cols(vt, rbase, 0, lane), pf.v, o[0], 0, 0, 0);
          o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q_frag_cols(vt, rbase, 32, lane), pf.v, o[1], 0, 0, 0);
        }
    }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qrow = q0 + (lane & 31);
  bf16_t* op = out + ((int64_t)b * Nq + qrow) * (H * DH) + h * DH;
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      u32x2_t w2;
      w2[0] = pack2bf(o[half][g4 * 4 + 0] * inv, o[half][g4 * 4 + 1] * inv);
      w2[1] = pack2bf(o[half][g4 * 4 + 2] * inv, o[half][g4 * 4 + 3] * inv);
      *(u32x2_t*)(op + half * 32 + g4 * 8 + hi * 4) = w2;
    }
  if (hi == 0) lse[(int64_t)bh * Nq + qrow] = (m_run + log2f(l_tot)) * LN2;
}
extern "C" int dl_probe_attn_fwd_fp8qk(const void* q8, const void* k8, const void* v, const float* scales, void* out, float* lse,
                                       int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t dh, float scale, const float* key_bias,
                                       dl_stream_t stream) {
  DL_CHECK_ARG(q8 && k8 && v && scales && out && lse && B > 0 && H > 0, "dl_probe_attn_fwd_fp8qk: null operand");
  DL_CHECK_ARG(dh == DH && Nq % FCH == 0 && Nk % FCH == 0 && Nq > 0 && Nk > 0 && Nq <= 4096 && Nk <= 4096,
               "dl_probe_attn_fwd_fp8qk: dh = 64, Nq / Nk multiples of 256 up to 4096");
  const int lds = FCH * DH + FCH * ROWB + FCH * (int)sizeof(float);
  hipLaunchKernelGGL(attn_fwd_fp8qk_k, (int)(B * H * (Nq / FCH)), 512, lds, (hipStream_t)stream, (const uint8_t*)q8, (const uint8_t*)k8,
                     (const bf16_t*)v, scales, (bf16_t*)out, lse, (int)H, (int)Nq, (int)Nk, scale, key_bias);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
