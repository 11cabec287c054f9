// bf16 MFMA GEMMs for the DiT linears (gfx950).
//
//   gemm_nt : C[M,N] = epilogue(A[M,K] . B[N,K]^T)         forward + dgrad (with transposed weight shadows)
//   gemm_tn : C[M,N] += A[R,M]^T . B[R,N]                   wgrad (reduction over tokens, split over R)
//
// Both: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each 64x64 = 2x2 MFMA 32x32x16 tiles),
// 64-deep K steps, operands streamed HBM/L2 -> LDS with direct-to-LDS 16-byte loads (global_load_lds_dwordx4),
// double buffered (one barrier per K step).  The LDS image written by the DMA is lane-linear, so the bank
// swizzle is applied to the per-lane SOURCE address and again on the ds_read side (guide rule 21).
//   nt: rows are 128 B (64 k);  16-B slot' = slot ^ ((row >> 1) & 7)  -> ds_read_b128 conflict-free per 16-lane group
//   tn: rows are 256 B (128 m); 16-B slot' = slot ^ (4 * (row & 3))   -> the 4 rows of one ds_read_b64_tr_b16
//       group land on 4 distinct 32-B bank groups
// Workgroup ids are remapped so that each XCD (own L2) walks a contiguous run of tiles sharing the A panel.
#include "common.h"

#define BM 128
#define BN 128
#define BK 64
#define NT_THREADS 256
#define STAGE_PITCH 68  // floats; 64 + 4 keeps 16-B alignment and skews rows across banks
#define NT_LDS_BYTES (4 * 64 * STAGE_PITCH * 4)  // 69632 >= 2 * 32 KiB of operand buffers

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void_t*)gsrc, (lds_void_t*)lds_dst_wave_base, 16, 0, 0);
}

struct NtEpilogue {
  const float* bias;
  int act;
  int out_f32;
  bf16_t* pre_out;
  const bf16_t* resid;
  int64_t ldr;
  const bf16_t* gate;
  int64_t ldg;
  int64_t rows_per_gate;
};

__global__ __launch_bounds__(NT_THREADS, 2) void gemm_nt_k(const bf16_t* __restrict__ A, int64_t lda,
                                                             const bf16_t* __restrict__ Bm, int64_t ldb,
                                                             void* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                             NtEpilogue ep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = (N + BN - 1) / BN;
  const int tiles_m = (M + BM - 1) / BM;
  const int lid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // ---- staging addresses: wave w DMA-copies chunks 4w..4w+3 (8 rows each) of the A tile and of the B tile
  const int srow = lane >> 3, sslot = lane & 7;
  const bf16_t* a_src[4];
  const bf16_t* b_src[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int r = (wave * 4 + c) * 8 + srow;
    const int q = sslot ^ ((r >> 1) & 7);
    int gm = m0 + r;
    gm = gm < M ? gm : M - 1;
    int gn = n0 + r;
    gn = gn < N ? gn : N - 1;
    a_src[c] = A + (int64_t)gm * lda + q * 8;
    b_src[c] = Bm + (int64_t)gn * ldb + q * 8;
  }
  auto stage = [&](int kt, int buf) {
    char* ba = smem + buf * 32768 + wave * 4096;
    char* bb = ba + 16384;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      glds16(a_src[c] + kt * BK, ba + c * 1024);
      glds16(b_src[c] + kt * BK, bb + c * 1024);
    }
  };

  // ---- fragment read offsets (bytes inside a 16 KiB operand tile), constant over K
  int a_off[2], b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = wr * 64 + i * 32 + (lane & 31);
    const int rb = wc * 64 + i * 32 + (lane & 31);
    a_off[i] = ra * 128 + ((((lane >> 5)) ^ ((ra >> 1) & 7)) << 4);
    b_off[i] = rb * 128 + ((((lane >> 5)) ^ ((rb >> 1) & 7)) << 4);
  }
  // slot for sub-step kk is (2kk + hi) ^ sw == ((2kk) ^ (hi ^ sw)) only when bit0 handling is separate; keep it simple:
  auto frag_off = [&](int base_row_off, int row, int kk) -> int {
    return base_row_off + ((((kk << 1) | (lane >> 5)) ^ ((row >> 1) & 7)) << 4);
  };
  (void)a_off;
  (void)b_off;

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const char* ta = smem + (kt & 1) * 32768;
    const char* tb = ta + 16384;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8_t af[2], bfg[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ra = wr * 64 + i * 32 + (lane & 31);
        const int rb = wc * 64 + i * 32 + (lane & 31);
        af[i] = *(const bf16x8_t*)(ta + frag_off(ra * 128, ra, kk));
        bfg[i] = *(const bf16x8_t*)(tb + frag_off(rb * 128, rb, kk));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfg[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: accumulators -> wave-private LDS slab (f32) -> row-contiguous 16-byte global stores
  __syncthreads();  // every wave is done reading the operand buffers
  float* slab = (float*)smem + wave * 64 * STAGE_PITCH;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        slab[row * STAGE_PITCH + j * 32 + (lane & 31)] = acc[i][j][r];
      }
  __syncthreads();
  const int erow = lane >> 3, ecol = (lane & 7) * 8;
#pragma unroll 2
  for (int p = 0; p < 8; ++p) {
    const int row = p * 8 + erow;
    const int m = m0 + wr * 64 + row;
    const int n = n0 + wc * 64 + ecol;
    if (m >= M || n >= N) continue;
    float v[8];
    *(f32x4_t*)&v[0] = *(const f32x4_t*)&slab[row * STAGE_PITCH + ecol];
    *(f32x4_t*)&v[4] = *(const f32x4_t*)&slab[row * STAGE_PITCH + ecol + 4];
    const bool full = (n + 8 <= N);
    if (ep.bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (full || n + e < N) v[e] += ep.bias[n + e];
    }
    if (ep.pre_out) {
      bf16_t* po = ep.pre_out + (int64_t)m * ldc + n;
      if (full) {
        *(u32x4_t*)po = pack8(v);
      } else {
        for (int e = 0; e < 8 && n + e < N; ++e) po[e] = f2bf(v[e]);
      }
    }
    if (ep.act == DL_ACT_SILU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
    }
    if (ep.resid) {
      const bf16_t* rp = ep.resid + (int64_t)m * ep.ldr + n;
      float rr[8], gg[8];
      if (full) {
        unpack8(*(const u32x4_t*)rp, rr);
      } else {
        for (int e = 0; e < 8; ++e) rr[e] = (n + e < N) ? bf2f(rp[e]) : 0.f;
      }
      if (ep.gate) {
        const bf16_t* gp = ep.gate + (int64_t)(m / ep.rows_per_gate) * ep.ldg + n;
        if (full) {
          unpack8(*(const u32x4_t*)gp, gg);
        } else {
          for (int e = 0; e < 8; ++e) gg[e] = (n + e < N) ? bf2f(gp[e]) : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rr[e] + gg[e] * v[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rr[e] + v[e];
      }
    }
    if (ep.out_f32) {
      float* cp = (float*)C + (int64_t)m * ldc + n;
      if (full) {
        *(f32x4_t*)cp = *(f32x4_t*)&v[0];
        *(f32x4_t*)(cp + 4) = *(f32x4_t*)&v[4];
      } else {
        for (int e = 0; e < 8 && n + e < N; ++e) cp[e] = v[e];
      }
    } else {
      bf16_t* cp = (bf16_t*)C + (int64_t)m * ldc + n;
      if (full) {
        *(u32x4_t*)cp = pack8(v);
      } else {
        for (int e = 0; e < 8 && n + e < N; ++e) cp[e] = f2bf(v[e]);
      }
    }
  }
}

extern "C" int dl_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                          int64_t N, int64_t K, const float* bias, int act, int out_dtype, void* pre_out,
                          const void* resid, int64_t ldr, const void* gate, int64_t ldg, int64_t rows_per_gate,
                          dl_stream_t stream) {
  DL_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0, "dl_gemm_nt: null/empty operand");
  DL_CHECK_ARG(K % BK == 0, "dl_gemm_nt: K=%lld must be a multiple of %d (zero-pad)", (long long)K, BK);
  DL_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && lda >= K && ldb >= K && ldc >= N,
               "dl_gemm_nt: leading dims must be multiples of 8 and cover the row (lda=%lld ldb=%lld ldc=%lld)",
               (long long)lda, (long long)ldb, (long long)ldc);
  DL_CHECK_ARG((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0, "dl_gemm_nt: 16-byte alignment");
  DL_CHECK_ARG(!resid || (ldr % 8 == 0 && ((uintptr_t)resid & 15) == 0), "dl_gemm_nt: resid alignment");
  DL_CHECK_ARG(!gate || (resid && rows_per_gate > 0 && ldg % 8 == 0 && ((uintptr_t)gate & 15) == 0),
               "dl_gemm_nt: gate needs resid, rows_per_gate>0, aligned rows");
  DL_CHECK_ARG(M < (1ll << 31) && N < (1ll << 31), "dl_gemm_nt: dims too large");
  NtEpilogue ep{bias, act, out_dtype == DL_F32, (bf16_t*)pre_out, (const bf16_t*)resid, ldr, (const bf16_t*)gate,
                ldg, rows_per_gate > 0 ? rows_per_gate : 1};
  const int nwg = cdiv(M, BM) * cdiv(N, BN);
  hipLaunchKernelGGL(gemm_nt_k, nwg, NT_THREADS, NT_LDS_BYTES, (hipStream_t)stream, (const bf16_t*)A, lda,
                     (const bf16_t*)B, ldb, C, ldc, (int)M, (int)N, (int)K, ep);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// =====================================================================================================
// TN (wgrad): C[m,n] += sum_r A[r,m] B[r,n].  Operand tiles are [64 r][128 cols] row-major in LDS (DMA'd
// straight from the row-major activations), MFMA fragments come out of ds_read_b64_tr_b16 transposing reads.
// =====================================================================================================
__device__ __forceinline__ s16x4_t lds_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}

__global__ __launch_bounds__(NT_THREADS, 2) void gemm_tn_k(const bf16_t* __restrict__ A, int64_t lda,
                                                             const bf16_t* __restrict__ Bm, int64_t ldb,
                                                             float* __restrict__ C, int64_t ldc, int M, int N, int R,
                                                             int steps_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = (N + BN - 1) / BN;
  const int tiles_m = (M + BM - 1) / BM;
  const int ntile = tiles_m * tiles_n;
  const int tile = blockIdx.x % ntile, split = blockIdx.x / ntile;
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int nsteps_total = R / BK;
  const int s_begin = split * steps_per_split;
  int s_end = s_begin + steps_per_split;
  s_end = s_end < nsteps_total ? s_end : nsteps_total;
  if (s_begin >= s_end) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // staging: a tile is 16 chunks of 4 rows x 256 B; wave w copies chunks 4w..4w+3 of both tiles
  const int srow = lane >> 4, sslot = lane & 15;
  const bf16_t* a_src[4];
  const bf16_t* b_src[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int r = (wave * 4 + c) * 4 + srow;
    const int q = sslot ^ ((r & 3) << 2);
    int cm = m0 + q * 8;
    cm = (cm + 8 <= M) ? cm : (M - 8);
    int cn = n0 + q * 8;
    cn = (cn + 8 <= N) ? cn : (N - 8);
    a_src[c] = A + (int64_t)r * lda + cm;
    b_src[c] = Bm + (int64_t)r * ldb + cn;
  }
  auto stage = [&](int st, int buf) {
    char* ba = smem + buf * 32768 + wave * 4096;
    char* bb = ba + 16384;
    const int64_t r0 = (int64_t)st * BK;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      glds16(a_src[c] + r0 * lda, ba + c * 1024);
      glds16(b_src[c] + r0 * ldb, bb + c * 1024);
    }
  };

  // transposing fragment read: 16-lane group g = lane>>4 reads the [4 r][16 cols] block
  //   rows  rbase + (li>>2), cols cb + 4*(li&3) .. +3        (li = lane & 15)
  // and lane li receives column cb+li of those 4 rows.  g&1 selects the 16-col half of the 32-wide MFMA
  // tile, g>>1 the k half (k = 8*(lane>>5) + j).
  const int li = lane & 15, g = lane >> 4;
  auto tr_off = [&](int col_tile_base, int kk, int half) -> int {
    const int r = kk * 16 + (g >> 1) * 8 + half * 4 + (li >> 2);
    const int col = col_tile_base + (g & 1) * 16 + (li & 3) * 4;
    const int slot = (col >> 3) ^ ((r & 3) << 2);
    return r * 256 + slot * 16 + (col & 7) * 2;
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  stage(s_begin, 0);
  for (int st = s_begin; st < s_end; ++st) {
    const int it = st - s_begin;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (st + 1 < s_end) stage(st + 1, (it + 1) & 1);
    const char* ta = smem + (it & 1) * 32768;
    const char* tb = ta + 16384;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8_t af[2], bfg[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        union {
          s16x4_t h[2];
          bf16x8_t v;
        } ua, ub;
        ua.h[0] = lds_tr16(ta + tr_off(wr * 64 + i * 32, kk, 0));
        ua.h[1] = lds_tr16(ta + tr_off(wr * 64 + i * 32, kk, 1));
        ub.h[0] = lds_tr16(tb + tr_off(wc * 64 + i * 32, kk, 0));
        ub.h[1] = lds_tr16(tb + tr_off(wc * 64 + i * 32, kk, 1));
        af[i] = ua.v;
        bfg[i] = ub.v;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfg[j], acc[i][j], 0, 0, 0);
    }
  }

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wc * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M && n < N) unsafeAtomicAdd(&C[(int64_t)m * ldc + n], acc[i][j][r]);
      }
    }
}

extern "C" int dl_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                          int64_t N, int64_t R, dl_stream_t stream) {
  DL_CHECK_ARG(A && B && C && M > 0 && N > 0 && R > 0, "dl_gemm_tn: null/empty operand");
  DL_CHECK_ARG(R % BK == 0, "dl_gemm_tn: R=%lld must be a multiple of %d", (long long)R, BK);
  DL_CHECK_ARG(M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= M && ldb >= N && ldc >= N,
               "dl_gemm_tn: M,N,lda,ldb must be multiples of 8 (M=%lld N=%lld)", (long long)M, (long long)N);
  DL_CHECK_ARG((((uintptr_t)A | (uintptr_t)B) & 15) == 0, "dl_gemm_tn: 16-byte alignment");
  const int ntile = cdiv(M, BM) * cdiv(N, BN);
  const int nsteps = (int)(R / BK);
  int splits = (1024 + ntile - 1) / ntile;  // aim at >= 4 workgroups per CU
  if (splits > nsteps) splits = nsteps;
  if (splits < 1) splits = 1;
  const int sps = (nsteps + splits - 1) / splits;
  splits = (nsteps + sps - 1) / sps;
  hipLaunchKernelGGL(gemm_tn_k, ntile * splits, NT_THREADS, 65536, (hipStream_t)stream, (const bf16_t*)A, lda,
                     (const bf16_t*)B, ldb, C, ldc, (int)M, (int)N, (int)R, sps);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// =====================================================================================================
// probe: raw lane map of ds_read_b64_tr_b16 (tests/test_gpu_probe.py pins the semantics gemm_tn / attention rely on)
// =====================================================================================================
__global__ void probe_tr16_k(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t img[256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) img[i] = (uint16_t)i;
  __syncthreads();
  s16x4_t v = lds_tr16((const char*)img + lane * 8);
#pragma unroll
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}
extern "C" int dl_probe_tr16(uint16_t* out, dl_stream_t stream) {
  DL_CHECK_ARG(out, "dl_probe_tr16: null");
  hipLaunchKernelGGL(probe_tr16_k, 1, 64, 0, (hipStream_t)stream, out);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
