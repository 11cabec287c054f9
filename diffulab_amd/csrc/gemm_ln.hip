// Row-complete GEMMs: C[M, 384-wide tiles] = A[M,K] . W[*,K]^T on persistent 256 x 384 tiles whose EPILOGUE is a row kernel.
//
// At D = 384 a 256 x 384 output tile holds WHOLE rows, and with 256 tokens per sample it holds one whole sample.  So the
// HBM-bound row kernels that used to run between the GEMMs of a DiT block (round-2 profile: 6.9 of 22.7 ms per step) can run on
// the tile while it is still on chip -- the GEMM result never makes the round trip through HBM and the launches disappear:
//   RK_LNF  projection / MLP-down / patch-embedding GEMM  ->  x += gate * t ; LayerNorm ; modulate       (mmdit.py:296-308, nn.py:539)
//           writes t (kept for the gate gradient), the new residual stream x, the modulated rows xm and the row statistics
//   RK_LNB  the two N = D data-gradient GEMMs (K = 3D and 2F) and the head's ->  LayerNorm-modulate backward + the backward of the
//           gated residual that follows in the chain; the per-sample column sums (dscale, dshift, dgate, LayerNorm-affine
//           partials) are COMPLETE inside the tile: plain stores, no atomics, a fixed summation order (deterministic)
//   RK_QK   qkv GEMM (N = 3D: tile 0 = q, 1 = k, 2 = v)  ->  RMSNorm over the full 384-wide q / k row, RoPE, head-major store
//           (mmdit.py:81-91, nn.py:345-352,430); the pre-norm qkv rows are still written (the backward and the in-place V need them)
//
// Main loop: the 256 x 384 walk of gemm_nt_big_k (gemm.hip): 8 waves as 4 (M) x 2 (N), each 64 x 192 = 2 x 6 MFMA 32x32x16 tiles,
// 64-deep k-steps DMA'd straight into a two-slot LDS ring (80 KiB per slot), source-side XOR swizzle, operand-swapped MFMAs.
//
// Epilogue = "the row kernel, fed from LDS": the accumulators are rounded to bf16 (exactly what the unfused GEMM would have
// stored), dumped 64 rows at a time into the LDS slot the last k-step has just released (row pitch 784 B: consecutive rows
// shift by four banks, the 16-byte writes of eight lanes cover 32 banks), and then every wave owns a ROW: lane c holds columns
// [8c, 8c+8) -- the thread layout of norm.hip's kernels, the same DPP wave sums in the same order, so every per-row output is
// BIT-IDENTICAL to the unfused launch sequence (tests/test_kernels_gpu.py checks equality), and per-sample column sums are
// plain in-lane accumulators over the tile's rows.  Global operands of the row (x, dres, t / residual) are prefetched one row
// ahead; their rows are contiguous 768-byte segments per wave instruction.
#include <stdlib.h>

#include "common.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// Direct-to-LDS 16-byte load in the SGPR-base + 32-bit lane-offset form.  Inline assembly on purpose: through the builtin the
// compiler materialises a 64-bit per-lane address for every chunk (20 VGPRs that it then spills next to the 192 accumulators,
// and a scratch reload inside the k-loop waits on vmcnt(0), i.e. drains the operand ring).  M0 (the LDS destination base) is
// compiler-reserved: it is saved and restored inside the statement.  The statement is not part of the compiler's vmcnt
// bookkeeping; every wait for these loads in this file is an explicit rk_wait_vmcnt.
static __device__ __forceinline__ void glds16_s(const void* sbase, uint32_t voff, const void* lds_dst_wave_base) {
  uint32_t keep;
  const uint32_t dst = (uint32_t)(uintptr_t)(lds_void_t*)lds_dst_wave_base;
  asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(dst)
               : "memory");
}

#define RK_BM 256
#define RK_BN 384
#define RK_BK 64
#define RK_THREADS 512
#define RK_STAGE ((RK_BM + RK_BN) * 128)  // bytes per ring slot (81920)
#define RK_CH 10                          // 1 KiB DMA chunks per wave per stage
#define RK_JN 6                           // 32-wide MFMA column tiles per wave
#define RK_PITCH 784                      // bytes per row of the epilogue's LDS piece
#define RK_D8 48                          // 16-byte chunks per 384-wide row = active lanes of a row wave

enum { RK_LNF = 0, RK_LNB = 1, RK_QK = 2 };

struct RowEpi {
  // ---- RK_LNF
  const bf16_t* resid;   // [M, D] residual stream before the gated add, or NULL (x = t: patch embedding)
  const bf16_t* gate;    // row view [groups, ld_gate] or NULL (then x = resid + t)
  int64_t ld_gate;
  const float* ln_w;     // f32 [D] or NULL (no affine)
  const float* ln_b;
  const bf16_t* scale;   // row views [groups, ld_mod]
  const bf16_t* shift;
  int64_t ld_mod;
  float eps;
  bf16_t* t_out;         // [M, D] GEMM result as bf16 (NULL: not kept)
  bf16_t* x_out;         // [M, D] new residual stream (NULL: not kept -- only legal when resid == NULL and nobody reads it)
  bf16_t* xm_out;        // [M, D] modulated rows
  float* mean;           // f32 [M]
  float* rstd;
  // ---- RK_LNB (ln_w, ln_b, scale, ld_mod, gate, ld_gate as above; gate = the gate of the residual that follows in the chain)
  const bf16_t* x;       // [M, D] LayerNorm input of the forward
  const float* mean_i;   // f32 [M]
  const float* rstd_i;
  const bf16_t* dres;    // [M, D] or NULL
  bf16_t* dx;            // [M, D]
  float* dscale;         // f32 row views [groups, ld_dmod], written (=)
  float* dshift;
  int64_t ld_dmod;
  float* dwb;            // f32 [groups, 2, D] written (=), or NULL
  const bf16_t* gt;      // [M, D] t of the gated residual that follows, or NULL
  bf16_t* gdt;           // [M, D]
  float* dgate;          // f32 row view, written (=)
  // ---- RK_QK
  bf16_t* qkv;           // [M, 3D] pre-norm rows
  bf16_t* qo;            // [B, H, n_dst, dh]
  bf16_t* ko;
  float* rrms;           // f32 [M, 2]
  const float* sq;       // f32 [D]
  const float* sk;
  const float* cs;       // f32 [tokens, rot/2]
  const float* sn;
  int n_tok, heads, dh, rot, n_dst, n_off;
};

template <int N>
static __device__ __forceinline__ void rk_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
static __device__ __forceinline__ void rk_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

static __device__ __forceinline__ u32x4_t rk_ld16(const bf16_t* p) { return *(const u32x4_t*)p; }

// ---------------------------------------------------------------------------------------------------------------------
// epilogue: P[j][i][q] = bf16 pairs of acc[j][i][2q], acc[j][i][2q+1]  (tile row wm*64 + i*32 + (lane & 31),
// tile column wn*192 + j*32 + 8*(r >> 2) + 4*hi + (r & 3), r = accumulator register)
// ---------------------------------------------------------------------------------------------------------------------
template <int MODE>
static __device__ __forceinline__ void rk_row_epilogue(uint32_t (&P)[RK_JN][2][8], char* lds, int tile_m, int tile_n, int lane,
                                                       int wave, const RowEpi& ep) {
  constexpr int D = RK_BN;
  const int hi = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const bool on = lane < RK_D8;
  const int64_t m0 = (int64_t)tile_m * RK_BM;
  const int64_t g = tile_m;  // modulation group = sample (256 rows per group)
  const float invD = 1.0f / (float)D;
  const int c8 = lane * 8;   // first column of this lane's chunk (row waves)

  // ---- per-tile constants of the row loop (lane = column chunk)
  float k0[8], k1[8], k2[8], k3[8], k4[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) k0[e] = k1[e] = k2[e] = k3[e] = k4[e] = 0.f;
  if constexpr (MODE == RK_LNF) {
    // k0 = w, k1 = b, k2 = scale, k3 = shift, k4 = gate
    if (on) {
      if (ep.ln_w) {
        *(f32x4_t*)&k0[0] = *(const f32x4_t*)(ep.ln_w + c8);
        *(f32x4_t*)&k0[4] = *(const f32x4_t*)(ep.ln_w + c8 + 4);
        *(f32x4_t*)&k1[0] = *(const f32x4_t*)(ep.ln_b + c8);
        *(f32x4_t*)&k1[4] = *(const f32x4_t*)(ep.ln_b + c8 + 4);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) k0[e] = 1.0f;
      }
      unpack8(rk_ld16(ep.scale + g * ep.ld_mod + c8), k2);
      unpack8(rk_ld16(ep.shift + g * ep.ld_mod + c8), k3);
      if (ep.gate) unpack8(rk_ld16(ep.gate + g * ep.ld_gate + c8), k4);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) k0[e] = 1.0f;
    }
  } else if constexpr (MODE == RK_LNB) {
    // k0 = (1 + scale) * w, k4 = gate of the residual that follows; k1..k3 = the three column accumulators S1, S2, S3
    if (on) {
      float wv[8], sc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) wv[e] = 1.0f;
      if (ep.ln_w) {
        *(f32x4_t*)&wv[0] = *(const f32x4_t*)(ep.ln_w + c8);
        *(f32x4_t*)&wv[4] = *(const f32x4_t*)(ep.ln_w + c8 + 4);
      }
      unpack8(rk_ld16(ep.scale + g * ep.ld_mod + c8), sc);
#pragma unroll
      for (int e = 0; e < 8; ++e) k0[e] = (1.0f + sc[e]) * wv[e];
      if (ep.gt) unpack8(rk_ld16(ep.gate + g * ep.ld_gate + c8), k4);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) k0[e] = 1.0f;
    }
  } else {
    // k0 = RMSNorm scale of this tile's third (q or k)
    if (on && tile_n < 2) {
      const float* s = tile_n == 0 ? ep.sq : ep.sk;
      *(f32x4_t*)&k0[0] = *(const f32x4_t*)(s + c8);
      *(f32x4_t*)&k0[4] = *(const f32x4_t*)(s + c8 + 4);
    }
  }

  // ---- global operands of a row, fetched one row ahead of its use
  u32x4_t pa = {0, 0, 0, 0}, pb = {0, 0, 0, 0}, pc = {0, 0, 0, 0};
  float pf0 = 0.f, pf1 = 0.f;
  auto row_of = [&](int s) -> int64_t {  // s = (i * 2 + pair) * 8 + rr: the s-th row this wave processes
    const int i = s >> 4, pair = (s >> 3) & 1, lr = (s & 7) * 8 + wave;
    return m0 + (2 * pair + (lr >> 5)) * 64 + i * 32 + (lr & 31);
  };
  auto fetch = [&](int s) {
    const int64_t row = row_of(s);
    if constexpr (MODE == RK_LNF) {
      if (on && ep.resid) pa = rk_ld16(ep.resid + row * D + c8);
    } else if constexpr (MODE == RK_LNB) {
      if (on) {
        pa = rk_ld16(ep.x + row * D + c8);
        if (ep.dres) pb = rk_ld16(ep.dres + row * D + c8);
        if (ep.gt) pc = rk_ld16(ep.gt + row * D + c8);
      }
      pf0 = ep.mean_i[row];
      pf1 = ep.rstd_i[row];
    } else {
      if (on && tile_n < 2) {
        const int dh = ep.dh, h = c8 / dh, d0 = c8 - h * dh;
        if (d0 < ep.rot) {
          const int n = (int)(row % ep.n_tok);
          const f32x4_t cc = *(const f32x4_t*)(ep.cs + (int64_t)n * (ep.rot >> 1) + (d0 >> 1));
          const f32x4_t ss = *(const f32x4_t*)(ep.sn + (int64_t)n * (ep.rot >> 1) + (d0 >> 1));
          pa = __builtin_bit_cast(u32x4_t, cc);
          pb = __builtin_bit_cast(u32x4_t, ss);
        }
      }
    }
  };

  fetch(0);
  rk_lds_barrier();  // every wave has finished the fragment reads of the last k-step: its ring slot is free
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    for (int pair = 0; pair < 2; ++pair) {
      // ---- dump 64 rows (this i of the waves wm = 2 pair, 2 pair + 1) as bf16: 16-byte chunks of 8 consecutive columns
      if ((wm >> 1) == pair) {
        char* base = lds + ((wm & 1) * 32 + (lane & 31)) * RK_PITCH + (wn * 192 + 8 * hi) * 2;
#pragma unroll
        for (int j = 0; j < RK_JN; ++j)
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(P[j][i][4 * gp + 0], P[j][i][4 * gp + 2], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(P[j][i][4 * gp + 1], P[j][i][4 * gp + 3], false, false);
            u32x4_t v = {s0[0], s1[0], s0[1], s1[1]};
            *(u32x4_t*)(base + (j * 32 + 16 * gp) * 2) = v;
          }
      }
      rk_lds_barrier();
      for (int rr = 0; rr < 8; ++rr) {
        const int s = (i * 2 + pair) * 8 + rr;
        const int lr = rr * 8 + wave;
        const int64_t row = row_of(s);
        const u32x4_t qa = pa, qb = pb, qc = pc;
        const float f0 = pf0, f1 = pf1;
        if (s + 1 < 32) fetch(s + 1);
        u32x4_t tq = {0, 0, 0, 0};
        if (on) tq = *(const u32x4_t*)(lds + lr * RK_PITCH + lane * 16);
        float tv[8];
        unpack8(tq, tv);

        if constexpr (MODE == RK_LNF) {
          if (on && ep.t_out) *(u32x4_t*)(ep.t_out + row * D + c8) = tq;
          float xv[8];
          if (ep.resid) {
            unpack8(qa, xv);
            if (ep.gate) {
#pragma unroll
              for (int e = 0; e < 8; ++e) xv[e] = bf2f(f2bf(xv[e] + k4[e] * tv[e]));  // statistics of what is stored
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) xv[e] = bf2f(f2bf(xv[e] + tv[e]));
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] = tv[e];
          }
          if (on && ep.x_out) *(u32x4_t*)(ep.x_out + row * D + c8) = pack8(xv);
          float sm = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) sm += xv[e];
          const float mu = wave_sum(sm) * invD;
          float q = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = on ? (xv[e] - mu) : 0.f;
            q += d * d;
          }
          const float rs = rsqrtf(wave_sum(q) * invD + ep.eps);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float y = (xv[e] - mu) * rs * k0[e] + k1[e];
            xv[e] = y * (1.0f + k2[e]) + k3[e];
          }
          if (on) *(u32x4_t*)(ep.xm_out + row * D + c8) = pack8(xv);
          if (lane == 0) {
            ep.mean[row] = mu;
            ep.rstd[row] = rs;
          }
        } else if constexpr (MODE == RK_LNB) {
          float xv[8], rv[8];
          unpack8(qa, xv);
#pragma unroll
          for (int e = 0; e < 8; ++e) rv[e] = 0.f;
          if (ep.dres) unpack8(qb, rv);
          const float mu = f0, rs = f1;
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xh = on ? (xv[e] - mu) * rs : 0.f;
            const float d = tv[e];
            k1[e] += d;
            k2[e] += d * xh;
            const float dxh = d * k0[e];
            s1 += dxh;
            s2 += dxh * xh;
            xv[e] = xh;
            tv[e] = dxh;
          }
          const float c1 = wave_sum(s1) * invD, c2 = wave_sum(s2) * invD;
#pragma unroll
          for (int e = 0; e < 8; ++e) rv[e] += rs * (tv[e] - c1 - xv[e] * c2);
          if (on) *(u32x4_t*)(ep.dx + row * D + c8) = pack8(rv);
          if (ep.gt) {
            float gtv[8];
            unpack8(qc, gtv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float dxr = bf2f(f2bf(rv[e]));
              k3[e] += on ? dxr * gtv[e] : 0.f;
              rv[e] = dxr * k4[e];
            }
            if (on) *(u32x4_t*)(ep.gdt + row * D + c8) = pack8(rv);
          }
        } else {
          // pre-norm row of this third of qkv (the backward's input; V is read in place by the attention kernels)
          if (on) *(u32x4_t*)(ep.qkv + row * (3 * D) + tile_n * D + c8) = tq;
          if (tile_n < 2) {
            float s1 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s1 += tv[e] * tv[e];
            const float rq = rsqrtf(wave_sum(s1) * invD + ep.eps);
            if (on) {
              const int dh = ep.dh, h = c8 / dh, d0 = c8 - h * dh;
#pragma unroll
              for (int e = 0; e < 8; ++e) tv[e] = tv[e] * rq * k0[e];
              if (d0 < ep.rot) {
                const f32x4_t cc = __builtin_bit_cast(f32x4_t, qa), ss = __builtin_bit_cast(f32x4_t, qb);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                  const float a = tv[2 * p], b = tv[2 * p + 1];
                  tv[2 * p] = a * cc[p] - b * ss[p];
                  tv[2 * p + 1] = a * ss[p] + b * cc[p];
                }
              }
              const int64_t bb = row / ep.n_tok;
              const int n = (int)(row - bb * ep.n_tok);
              const int64_t o = ((bb * ep.heads + h) * ep.n_dst + ep.n_off + n) * dh + d0;
              *(u32x4_t*)((tile_n == 0 ? ep.qo : ep.ko) + o) = pack8(tv);
            }
            if (lane == 0) ep.rrms[row * 2 + tile_n] = rq;
          }
        }
      }
      rk_lds_barrier();  // the piece is rewritten by the next dump (or by the next tile's operand DMA)
    }
  }

  if constexpr (MODE == RK_LNB) {
    // ---- per-sample column sums: one [3][D] f32 slot per wave, summed in wave order (deterministic), plain stores
    float* red = (float*)lds;
    if (on) {
      float* slot = red + (size_t)wave * 3 * D + c8;
      *(f32x4_t*)(slot) = *(f32x4_t*)&k1[0];
      *(f32x4_t*)(slot + 4) = *(f32x4_t*)&k1[4];
      *(f32x4_t*)(slot + D) = *(f32x4_t*)&k2[0];
      *(f32x4_t*)(slot + D + 4) = *(f32x4_t*)&k2[4];
      *(f32x4_t*)(slot + 2 * D) = *(f32x4_t*)&k3[0];
      *(f32x4_t*)(slot + 2 * D + 4) = *(f32x4_t*)&k3[4];
    }
    rk_lds_barrier();
    const int col = threadIdx.x;
    if (col < D) {
      float S1 = 0.f, S2 = 0.f, S3 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        S1 += red[(size_t)w * 3 * D + col];
        S2 += red[(size_t)w * 3 * D + D + col];
        S3 += red[(size_t)w * 3 * D + 2 * D + col];
      }
      const float wc = ep.ln_w ? ep.ln_w[col] : 1.0f, bc = ep.ln_b ? ep.ln_b[col] : 0.0f;
      const float sc1 = 1.0f + bf2f(ep.scale[g * ep.ld_mod + col]);
      ep.dscale[g * ep.ld_dmod + col] = wc * S2 + bc * S1;
      ep.dshift[g * ep.ld_dmod + col] = S1;
      if (ep.gt) ep.dgate[g * ep.ld_dmod + col] = S3;
      if (ep.dwb) {
        ep.dwb[(size_t)g * 2 * D + col] = sc1 * S2;
        ep.dwb[(size_t)g * 2 * D + D + col] = sc1 * S1;
      }
    }
    rk_lds_barrier();
  }
}

template <int MODE>
__global__ __launch_bounds__(RK_THREADS, 2) void gemm_nt_rows_k(const bf16_t* __restrict__ A, int64_t lda,
                                                                const bf16_t* __restrict__ Bm, int64_t ldb, int M, int N, int K,
                                                                RowEpi ep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = N / RK_BN, ntiles = (M / RK_BM) * tiles_n;
  const int nk = K / RK_BK;
  const int G = gridDim.x;  // multiple of 8, <= ntiles
  const int slot0 = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int cnt = (ntiles - slot0 + G - 1) / G;
  const int total = cnt * nk;

  // DMA cursor: one stage ahead of the compute cursor
  int s_tile = slot0, s_kt = 0, s_it = 0;
  const bf16_t* s_ta = A + (int64_t)((s_tile / tiles_n) * RK_BM) * lda;
  const bf16_t* s_tb = Bm + (int64_t)((s_tile % tiles_n) * RK_BN) * ldb;

  f32x16_t acc[RK_JN][2];
#pragma unroll
  for (int j = 0; j < RK_JN; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  int it = 0;
  for (int t = 0; t < cnt; ++t) {
    const int tile = slot0 + t * G;
    // Per-lane constants of the main loop are re-derived for every tile from an opaque copy of the lane id: kept live across
    // the row epilogue they were spilled, and a scratch reload in the k-loop waits on vmcnt(0), i.e. drains the operand DMA.
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int lhi = ln >> 5;
    uint32_t src_off[RK_CH];  // byte offset of this lane's 16 bytes inside the A / B row panel of a tile (k = 0)
    int lds_off[RK_CH];
    bool is_a[RK_CH];
#pragma unroll
    for (int i = 0; i < RK_CH; ++i) {
      const int c = wave * RK_CH + i;
      is_a[i] = c < RK_BM / 8;
      const int cc = is_a[i] ? c : c - RK_BM / 8;
      const int r = cc * 8 + (ln >> 3);
      const int q = (ln & 7) ^ ((r >> 1) & 7);
      src_off[i] = (uint32_t)(r * (int)(is_a[i] ? lda : ldb) + q * 8) * 2u;
      lds_off[i] = (is_a[i] ? 0 : RK_BM * 128) + cc * 1024;
    }
    auto stage_next = [&]() {
      char* base = smem + (s_it & 1) * RK_STAGE;
      const char* ka = (const char*)(s_ta + s_kt * RK_BK);
      const char* kb = (const char*)(s_tb + s_kt * RK_BK);
#pragma unroll
      for (int i = 0; i < RK_CH; ++i) glds16_s(is_a[i] ? ka : kb, src_off[i], base + lds_off[i]);
      ++s_it;
      if (++s_kt == nk) {
        s_kt = 0;
        s_tile += G;
        s_ta = A + (int64_t)((s_tile / tiles_n) * RK_BM) * lda;
        s_tb = Bm + (int64_t)((s_tile % tiles_n) * RK_BN) * ldb;
      }
    };
    int xrow[2], wrow[RK_JN];
#pragma unroll
    for (int i = 0; i < 2; ++i) xrow[i] = wm * 64 + i * 32 + (ln & 31);
#pragma unroll
    for (int j = 0; j < RK_JN; ++j) wrow[j] = wn * (RK_BN / 2) + j * 32 + (ln & 31);

    if (t == 0) stage_next();
    for (int kt = 0; kt < nk; ++kt, ++it) {
      rk_wait_vmcnt<0>();             // stage `it` has landed (two-slot ring: nothing younger is in flight)
      __builtin_amdgcn_s_barrier();   // raw barrier: every wave's DMA share has landed and the other slot's readers are done
      if (s_it < total) stage_next();
      const char* sa = smem + (it & 1) * RK_STAGE;
      const char* sb = sa + RK_BM * 128;
      // fragment software pipeline of gemm_nt_big_k: weight fragment two MFMA pairs ahead, activation fragments one sub-step ahead
      bf16x8_t xq[2][2], wq[3];
      auto rd_x = [&](int kk, int i) -> bf16x8_t {
        return *(const bf16x8_t*)(sa + xrow[i] * 128 + ((((kk << 1) | lhi) ^ ((xrow[i] >> 1) & 7)) << 4));
      };
      auto rd_w = [&](int kk, int j) -> bf16x8_t {
        return *(const bf16x8_t*)(sb + wrow[j] * 128 + ((((kk << 1) | lhi) ^ ((wrow[j] >> 1) & 7)) << 4));
      };
      xq[0][0] = rd_x(0, 0);
      xq[0][1] = rd_x(0, 1);
      wq[0] = rd_w(0, 0);
      wq[1] = rd_w(0, 1);
#pragma unroll
      for (int s = 0; s < 4 * RK_JN; ++s) {
        const int kk = s / RK_JN, j = s % RK_JN;
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < 4 * RK_JN) wq[(s + 2) % 3] = rd_w((s + 2) / RK_JN, (s + 2) % RK_JN);
        if (kk < 3 && j < 2) xq[(kk + 1) & 1][j] = rd_x(kk + 1, j);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[s % 3], xq[kk & 1][i], acc[j][i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    uint32_t P[RK_JN][2][8];
#pragma unroll
    for (int j = 0; j < RK_JN; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int q = 0; q < 8; ++q) P[j][i][q] = pack2bf(acc[j][i][2 * q], acc[j][i][2 * q + 1]);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
      }
    // the slot consumed by the last k-step; the other one may already be receiving the next tile's first stage
    rk_row_epilogue<MODE>(P, smem + ((it - 1) & 1) * RK_STAGE, tile / tiles_n, tile % tiles_n, lane, wave, ep);
  }
}

template <int MODE>
static int rk_launch(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int64_t N, int64_t K, const RowEpi& ep,
                     hipStream_t stream) {
  constexpr int LDS = 2 * RK_STAGE;
  static DevOnce once;
  const int n_cu = dev_cus(once, [] {
    (void)hipFuncSetAttribute((const void*)gemm_nt_rows_k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  });
  const int ntiles = (int)((M / RK_BM) * (N / RK_BN));
  const int budget = dl_wg_budget(n_cu);
  int grid = budget < ntiles ? budget : ntiles;
  grid &= ~7;
  hipLaunchKernelGGL((gemm_nt_rows_k<MODE>), grid, RK_THREADS, LDS, stream, (const bf16_t*)A, lda, (const bf16_t*)W, ldw, (int)M,
                     (int)N, (int)K, ep);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// shapes the row-complete tiles serve: rows are exactly one tile wide, a modulation group is exactly one tile high
static bool rk_shape_ok(int64_t M, int64_t D, int64_t K, int64_t rows_per_mod) {
  return D == RK_BN && M % RK_BM == 0 && M / RK_BM >= 8 && K % RK_BK == 0 && K > 0 && rows_per_mod == RK_BM && M < (1ll << 31);
}
#define RK_ALIGNED16(p) ((((uintptr_t)(p)) & 15) == 0)

extern "C" int dl_ln_modulate_gemm_fwd(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int64_t K,
                                       const void* resid, const void* gate, int64_t ld_gate, const float* ln_w, const float* ln_b,
                                       const void* scale, const void* shift, int64_t ld_mod, int64_t rows_per_mod, float eps,
                                       void* t_out, void* x_out, void* xm_out, float* mean, float* rstd, int64_t D,
                                       dl_stream_t stream) {
  DL_CHECK_ARG(A && W && scale && shift && xm_out && mean && rstd && M > 0, "dl_ln_modulate_gemm_fwd: null operand");
  DL_CHECK_ARG((ln_w == nullptr) == (ln_b == nullptr), "dl_ln_modulate_gemm_fwd: w and b must both be given or both NULL");
  DL_CHECK_ARG(!gate || resid, "dl_ln_modulate_gemm_fwd: a gate needs the residual it gates into");
  DL_CHECK_ARG(!resid || x_out, "dl_ln_modulate_gemm_fwd: the updated residual stream needs x_out");
  if (!rk_shape_ok(M, D, K, rows_per_mod)) {
    dl_set_error("dl_ln_modulate_gemm_fwd: no row-complete tile for M=%lld D=%lld K=%lld rows_per_mod=%lld", (long long)M,
                 (long long)D, (long long)K, (long long)rows_per_mod);
    return DL_ERR_UNSUPPORTED;
  }
  DL_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K && ld_mod % 8 == 0 && (!gate || ld_gate % 8 == 0),
               "dl_ln_modulate_gemm_fwd: leading dimensions");
  DL_CHECK_ARG(RK_ALIGNED16(A) && RK_ALIGNED16(W) && RK_ALIGNED16(resid) && RK_ALIGNED16(gate) && RK_ALIGNED16(scale) &&
                   RK_ALIGNED16(shift) && RK_ALIGNED16(t_out) && RK_ALIGNED16(x_out) && RK_ALIGNED16(xm_out) && RK_ALIGNED16(ln_w) &&
                   RK_ALIGNED16(ln_b),
               "dl_ln_modulate_gemm_fwd: 16-byte alignment");
  RowEpi ep{};
  ep.resid = (const bf16_t*)resid;
  ep.gate = (const bf16_t*)gate;
  ep.ld_gate = ld_gate;
  ep.ln_w = ln_w;
  ep.ln_b = ln_b;
  ep.scale = (const bf16_t*)scale;
  ep.shift = (const bf16_t*)shift;
  ep.ld_mod = ld_mod;
  ep.eps = eps;
  ep.t_out = (bf16_t*)t_out;
  ep.x_out = (bf16_t*)x_out;
  ep.xm_out = (bf16_t*)xm_out;
  ep.mean = mean;
  ep.rstd = rstd;
  return rk_launch<RK_LNF>(A, lda, W, ldw, M, D, K, ep, (hipStream_t)stream);
}

extern "C" int dl_ln_modulate_gemm_bwd(const void* A, int64_t lda, const void* Wt, int64_t ldw, int64_t M, int64_t K,
                                       const void* x, const float* ln_w, const float* ln_b, const void* scale, int64_t ld_mod,
                                       int64_t rows_per_mod, const float* mean, const float* rstd, const void* dres, void* dx,
                                       float* dscale, float* dshift, int64_t ld_dmod, float* dwb, const void* gate_t,
                                       const void* gate, int64_t ld_gate, void* dt, float* dgate, int64_t D, dl_stream_t stream) {
  DL_CHECK_ARG(A && Wt && x && scale && mean && rstd && dx && dscale && dshift && M > 0, "dl_ln_modulate_gemm_bwd: null operand");
  DL_CHECK_ARG((ln_w == nullptr) == (ln_b == nullptr), "dl_ln_modulate_gemm_bwd: w and b must both be given or both NULL");
  DL_CHECK_ARG(!gate_t || (gate && dt && dgate), "dl_ln_modulate_gemm_bwd: the fused gate backward needs gate_t, gate, dt and dgate");
  if (!rk_shape_ok(M, D, K, rows_per_mod)) {
    dl_set_error("dl_ln_modulate_gemm_bwd: no row-complete tile for M=%lld D=%lld K=%lld rows_per_mod=%lld", (long long)M,
                 (long long)D, (long long)K, (long long)rows_per_mod);
    return DL_ERR_UNSUPPORTED;
  }
  DL_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K && ld_mod % 8 == 0 && (!gate_t || ld_gate % 8 == 0),
               "dl_ln_modulate_gemm_bwd: leading dimensions");
  DL_CHECK_ARG(RK_ALIGNED16(A) && RK_ALIGNED16(Wt) && RK_ALIGNED16(x) && RK_ALIGNED16(scale) && RK_ALIGNED16(dres) &&
                   RK_ALIGNED16(dx) && RK_ALIGNED16(gate_t) && RK_ALIGNED16(gate) && RK_ALIGNED16(dt) && RK_ALIGNED16(ln_w),
               "dl_ln_modulate_gemm_bwd: 16-byte alignment");
  RowEpi ep{};
  ep.x = (const bf16_t*)x;
  ep.ln_w = ln_w;
  ep.ln_b = ln_b;
  ep.scale = (const bf16_t*)scale;
  ep.ld_mod = ld_mod;
  ep.mean_i = mean;
  ep.rstd_i = rstd;
  ep.dres = (const bf16_t*)dres;
  ep.dx = (bf16_t*)dx;
  ep.dscale = dscale;
  ep.dshift = dshift;
  ep.ld_dmod = ld_dmod;
  ep.dwb = dwb;
  ep.gt = (const bf16_t*)gate_t;
  ep.gate = (const bf16_t*)gate;
  ep.ld_gate = ld_gate;
  ep.gdt = (bf16_t*)dt;
  ep.dgate = dgate;
  return rk_launch<RK_LNB>(A, lda, Wt, ldw, M, D, K, ep, (hipStream_t)stream);
}

extern "C" int dl_gemm_nt_qk_norm_rope(const void* A, int64_t lda, const void* Wqkv, int64_t ldw, int64_t B, int64_t N, int64_t H,
                                       int64_t dh, int64_t rot, float eps, const float* scale_q, const float* scale_k,
                                       const float* cos, const float* sin, void* qkv, void* q, void* k, float* rrms, int64_t n_dst,
                                       int64_t n_off, dl_stream_t stream) {
  DL_CHECK_ARG(A && Wqkv && scale_q && scale_k && cos && sin && qkv && q && k && rrms && B > 0 && N > 0,
               "dl_gemm_nt_qk_norm_rope: null operand");
  const int64_t D = H * dh, M = B * N;
  if (!rk_shape_ok(M, D, D, RK_BM) || dh % 8 || rot % 8 || rot > dh) {
    dl_set_error("dl_gemm_nt_qk_norm_rope: no row-complete tile for M=%lld D=%lld dh=%lld", (long long)M, (long long)D, (long long)dh);
    return DL_ERR_UNSUPPORTED;
  }
  DL_CHECK_ARG(n_off >= 0 && n_off + N <= n_dst, "dl_gemm_nt_qk_norm_rope: row window outside n_dst");
  DL_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && lda >= D && ldw >= D, "dl_gemm_nt_qk_norm_rope: leading dimensions");
  DL_CHECK_ARG(RK_ALIGNED16(A) && RK_ALIGNED16(Wqkv) && RK_ALIGNED16(qkv) && RK_ALIGNED16(q) && RK_ALIGNED16(k) &&
                   RK_ALIGNED16(scale_q) && RK_ALIGNED16(scale_k) && RK_ALIGNED16(cos) && RK_ALIGNED16(sin),
               "dl_gemm_nt_qk_norm_rope: 16-byte alignment");
  RowEpi ep{};
  ep.eps = eps;
  ep.qkv = (bf16_t*)qkv;
  ep.qo = (bf16_t*)q;
  ep.ko = (bf16_t*)k;
  ep.rrms = rrms;
  ep.sq = scale_q;
  ep.sk = scale_k;
  ep.cs = cos;
  ep.sn = sin;
  ep.n_tok = (int)N;
  ep.heads = (int)H;
  ep.dh = (int)dh;
  ep.rot = (int)rot;
  ep.n_dst = (int)n_dst;
  ep.n_off = (int)n_off;
  return rk_launch<RK_QK>(A, lda, Wqkv, ldw, M, 3 * D, D, ep, (hipStream_t)stream);
}
