// Ring-buffered weight-gradient GEMM (gfx950): 384 x 192 output tiles, operand stages in a four-slot LDS ring.
//
// Why a new kernel: the older weight-gradient kernels of gemm.hip (384 x 128 tile, 96 x 64 per wave, two ring slots) (a) read
// 160 KiB of LDS fragments and stage 64 KiB of operands per 6.3 MFLOP and (b) never overlapped their operand DMA with their MFMAs
// at all: behind a direct-to-LDS DMA the compiler's waitcnt pass puts `s_waitcnt vmcnt(0)` in front of every compiler-visible
// transposing LDS read (it cannot tell the read from the DMA's LDS target), so the stage prefetched for the NEXT k-step was waited
// for before the current one was consumed.  Here:
//   * the transposing reads are inline assembly without a memory operand, their lgkmcnt wait is explicit (W4_LANDED);
//   * the wave tile is 96 x 96 (8 waves, two per SIMD, 212 VGPRs): 268 KiB of fragment reads and 144 KiB of staged operands per
//     18.9 MFLOP (x0.56 / x0.75 per FLOP of the old kernel);
//   * both fragment sets of a 16-deep sub-step are double buffered IN REGISTERS and the reads of sub-step p+1 are dealt out over
//     the MFMA rows of sub-step p (one scheduling region per row);
//   * operand stages are 32 tokens deep in a FOUR-slot ring: three stages (108 KiB) stay in flight under counted vmcnt waits;
//   * one barrier per stage, placed after the first MFMA row of the sub-step so the matrix pipe runs through the barrier skew;
//   * all tiles of one token range sit on ONE XCD: every operand byte crosses into exactly one L2.
// Two launch forms share the tile body (w4_tile):
//   gemm_tn_w4_k        one problem, token ranges meet in C through f32 atomics (dl_gemm_tn / dl_gemm_tn_ex: any caller, any shape
//                       the tile divides);
//   gemm_tn_group_k     up to four problems over the SAME token rows in ONE launch (the four weight gradients of a transformer
//                       block: 32 tiles at D = 384), every (token range, tile) writes its f32 partial tile with plain stores into
//                       slab[range], and tn_group_fold_k adds the ranges in a fixed order: g += ((p0 + p1) + p2) + ...  No atomics:
//                       the result is bit-reproducible, the 68 M atomics per block of the one-problem launches (round 2: 50 of the
//                       176 us of the MLP-up weight gradient, 33 of 58 us of the projection's) become 75 MB of plain stores and a
//                       fold, and a workgroup runs 256 - 512 stages instead of 25 - 128.  dl_gemm_tn_group.
// (A 4-wave, one-wave-per-SIMD form with 192 x 96 wave tiles was measured in round 2 -- 275 vs 176 us: the compiler keeps MFMA
// accumulators in the 256 AGPRs only and shuttles the other 32 through v_accvgpr moves -- and is not kept.)
#include <stdlib.h>

#include <type_traits>

#include "common.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ void glds16_w4(const void* gsrc, void* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void_t*)gsrc, (lds_void_t*)lds_dst_wave_base, 16, 0, 0);
}
// Transposing LDS read as inline assembly ON PURPOSE: for a compiler-visible LDS load that follows a direct-to-LDS DMA the
// waitcnt pass inserts s_waitcnt vmcnt(0) (it cannot prove that the read does not alias the DMA's LDS target), which drains the
// whole operand ring before every fragment read -- the DMA then never overlaps the MFMAs.  The asm read carries no memory
// operand; the lgkmcnt wait for its result is explicit (w4_frags_landed).
template <int OFF>
__device__ __forceinline__ s16x4_t lds_tr16_w4(unsigned addr) {
  s16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt_w4() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#define W4_THREADS 256
#define W4_BK 32    // tokens per operand stage
#define W4_NST 4    // ring slots
#define W4_AM 384   // output rows (A-operand columns) per tile

// =====================================================================================================
// w4_tile: one 384 (m) x BNW (n) output tile of C = A[R, Mo]^T . B[R, No] over the operand stages [s_begin, s_end) (32 tokens each),
// left in the accumulators.  BNW = 192; 8 waves as 4 (m) x 2 (n), each 96 x 96 = 3 x 3 MFMA 32x32x16 tiles.
// LDS images are the row-major [32 r][384 | BNW] slabs written by the DMA; fragments come out of ds_read_b64_tr_b16.  16-byte slot
// swizzle per image row so that the 4 rows of one transposing read fall into 4 distinct 64-byte bank groups:
//   768- and 512-byte rows (multiples of the 256-byte bank period): slot ^= (r & 3) << 2
//   384-byte rows (r and r+2 alias):                                slot ^= ((r >> 1) & 1) << 2
// A, Bm point at column 0 of the operands (the tile's column offsets m0 / n0 are applied here).
// =====================================================================================================
#define W4_WAVES 8
#define W4_BNW 192
#define W4_LDS (W4_NST * W4_BK * (W4_AM + W4_BNW) * 2)
template <int AM, int BNW>
__device__ __forceinline__ void w4_tile(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ Bm, int64_t ldb,
                                        int m0, int n0, int s_begin, int s_end, char* smem,
                                        f32x16_t (&acc)[AM / (W4_WAVES / 2) / 32][BNW / 64]) {
  constexpr int WAVES = W4_WAVES, WM = WAVES / 2;     // waves along m (2 along n)
  constexpr int IM = AM / WM / 32, JN = BNW / 64;    // MFMA tiles per wave
  constexpr int PA = AM * 2, PB = BNW * 2;           // image row pitches (bytes)
  constexpr int A_BYTES = W4_BK * PA;                // 24 KiB
  constexpr int STAGE = W4_BK * (PA + PB);           // 36 | 40 KiB
  constexpr int ACH = A_BYTES / 1024;                // 1 KiB DMA chunks of the A image
  constexpr int NCH = STAGE / 1024;                  // 36 | 40
  constexpr int CH = (NCH + WAVES - 1) / WAVES;      // chunks per wave and stage (the last waves may own one less)
  constexpr int CHMIN = NCH / WAVES;                 // what the counted waits assume is outstanding per younger stage
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;  // LDS byte address of the ring (asm reads take addresses)
  const int n = s_end - s_begin;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- DMA: the stage image is NCH chunks of 1 KiB (lane-linear in LDS); wave w copies chunks [w*CH, (w+1)*CH)
  // everything that depends on which operand a chunk belongs to is resolved here, so that the issue inside the main loop is
  // straight-line: per-lane source pointer at relative stage 0, uniform byte... element step per stage, uniform LDS offset
  const bf16_t* cptr[CH];
  int64_t cstep[CH];
  int clds[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    int c = i * WAVES + wave;
    if (c >= NCH) c = NCH - 1;  // (never issued: see `extra`)
    const bool a = c < ACH;
    const int cc = a ? c : c - ACH;
    const int pitch = a ? PA : PB;
    const int o = cc * 1024 + lane * 16;
    const int r = o / pitch, s = (o - r * pitch) >> 4;
    const int q = s ^ ((a || BNW == 256) ? ((r & 3) << 2) : (((r >> 1) & 1) << 2));
    const int64_t ld = a ? lda : ldb;
    cptr[i] = (a ? A + m0 : Bm + n0) + ((int64_t)s_begin * W4_BK + r) * ld + q * 8;
    cstep[i] = (int64_t)W4_BK * ld;
    clds[i] = (a ? 0 : A_BYTES) + cc * 1024;
  }
  auto dma_chunk = [&](int st, int i) {  // chunk i of this wave of (relative) stage st
    glds16_w4(cptr[i] + (int64_t)st * cstep[i], smem + (st & (W4_NST - 1)) * STAGE + clds[i]);
  };

  // ---- per-lane byte offsets of the transposing fragment reads inside a stage image (sub-step and half add constants)
  const int li = lane & 15, g = lane >> 4;
  const int rl = (g >> 1) * 8 + (li >> 2);
  int aoff[IM], boff[JN];
#pragma unroll
  for (int i = 0; i < IM; ++i) {
    const int col = wm * (AM / WM) + i * 32 + (g & 1) * 16 + (li & 3) * 4;
    aoff[i] = rl * PA + (((col >> 3) ^ ((rl & 3) << 2)) << 4) + (col & 7) * 2;
  }
#pragma unroll
  for (int j = 0; j < JN; ++j) {
    const int col = wn * (BNW / 2) + j * 32 + (g & 1) * 16 + (li & 3) * 4;
    const int sw = BNW == 256 ? ((rl & 3) << 2) : (((rl >> 1) & 1) << 2);
    boff[j] = A_BYTES + rl * PB + (((col >> 3) ^ sw) << 4) + (col & 7) * 2;
  }

#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8_t af[2][IM], bfr[2][JN];
  union Frag {
    s16x4_t h[2];
    bf16x8_t v;
  };
  // fragment f of a sub-step: f < IM -> A tile f, else B tile f - IM (two 4-row halves of the 16-deep sub-step each)
#define W4_READ1(BUF, F, SBASE, KK)                                                  \
  do {                                                                               \
    Frag u;                                                                          \
    if ((F) < IM) {                                                                  \
      u.h[0] = lds_tr16_w4<((KK) * 16) * PA>((SBASE) + aoff[(F) < IM ? (F) : 0]);      \
      u.h[1] = lds_tr16_w4<((KK) * 16 + 4) * PA>((SBASE) + aoff[(F) < IM ? (F) : 0]);  \
      af[BUF][(F) < IM ? (F) : 0] = u.v;                                             \
    } else {                                                                         \
      u.h[0] = lds_tr16_w4<((KK) * 16) * PB>((SBASE) + boff[(F) < IM ? 0 : (F) - IM]);     \
      u.h[1] = lds_tr16_w4<((KK) * 16 + 4) * PB>((SBASE) + boff[(F) < IM ? 0 : (F) - IM]); \
      bfr[BUF][(F) < IM ? 0 : (F) - IM] = u.v;                                       \
    }                                                                                \
  } while (0)
  // every fragment of set BUF has landed: the wait is tied to the registers so that no MFMA on them can be scheduled above it
#define W4_LANDED(BUF)                                                                          \
  do {                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                          \
    _Pragma("unroll") for (int i = 0; i < IM; ++i) asm volatile("" : "+v"(af[BUF][i]));          \
    _Pragma("unroll") for (int j = 0; j < JN; ++j) asm volatile("" : "+v"(bfr[BUF][j]));         \
  } while (0)
#define W4_READ(BUF, SBASE, KK) \
  _Pragma("unroll") for (int f = 0; f < IM + JN; ++f) W4_READ1(BUF, f, SBASE, KK)
#define W4_MFMA_ROW(BUF, I)                                             \
  _Pragma("unroll") for (int j = 0; j < JN; ++j) {                      \
    acc[I][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[BUF][I], bfr[BUF][j], acc[I][j], 0, 0, 0); \
  }

  // ---- prologue: fill the ring, wait for stage 0, fetch its first fragment set
  constexpr int CHF = NCH / WAVES;  // chunks every wave owns; with a remainder the first NCH % WAVES waves own one more
  const bool extra = (NCH % WAVES) && wave < NCH % WAVES;
  {
    const int pre = n < W4_NST ? n : W4_NST;
    for (int st = 0; st < pre; ++st) {
#pragma unroll
      for (int i = 0; i < CHF; ++i) dma_chunk(st, i);
      if (extra) dma_chunk(st, CHF);
    }
    if (n >= W4_NST) wait_vmcnt_w4<(W4_NST - 1) * CHMIN>();
    else wait_vmcnt_w4<0>();
    __builtin_amdgcn_s_barrier();
    W4_READ(0, lds0, 0);
  }

  // one operand stage = two 16-deep sub-steps.  FULL: steady state (stages it+1 .. it+NST exist), no conditions inside, so
  // each sub-step is ONE scheduling region and the interleave below is what the hardware sees
  auto stage_body = [&](auto full_c, int it) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_c)::value;
    constexpr int NF = IM + JN;
    const unsigned sb = lds0 + (it & (W4_NST - 1)) * STAGE;
    W4_LANDED(0);
    // ---- sub-step 0: MFMAs on set 0; the transposing reads of set 1 (same stage) are dealt out over the MFMA rows, one
    //      scheduling region per row, so every group of JN MFMAs has its share of LDS reads next to it
    constexpr int FPR0 = (NF + IM - 1) / IM;
#pragma unroll
    for (int i = 0; i < IM; ++i) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = i * FPR0; f < (i + 1) * FPR0 && f < NF; ++f) W4_READ1(1, f, sb, 1);
      W4_MFMA_ROW(0, i);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- sub-step 1: the last reads of this stage have landed -> the slot can be refilled once every wave is here
    W4_LANDED(1);
    W4_MFMA_ROW(1, 0);  // (the matrix pipe runs through the barrier skew)
    if (FULL) wait_vmcnt_w4<(W4_NST - 2) * CHMIN>();  // stage it+1 landed; it+2, it+3 may stay in flight
    else wait_vmcnt_w4<0>();
    __builtin_amdgcn_s_barrier();
    const unsigned sn = lds0 + ((it + 1) & (W4_NST - 1)) * STAGE;
    constexpr int FPR1 = (NF + IM - 2) / (IM - 1), CPR = (CHF + IM - 2) / (IM - 1);
#pragma unroll
    for (int i = 1; i < IM; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      if (FULL || it + 1 < n) {
#pragma unroll
        for (int f = (i - 1) * FPR1; f < i * FPR1 && f < NF; ++f) W4_READ1(0, f, sn, 0);
      }
      if (FULL) {
#pragma unroll
        for (int c = (i - 1) * CPR; c < i * CPR && c < CHF; ++c) dma_chunk(it + W4_NST, c);
      }
      W4_MFMA_ROW(1, i);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (FULL && extra) dma_chunk(it + W4_NST, CHF);
  };
  int it = 0;
  for (; it + W4_NST < n; ++it) stage_body(std::true_type{}, it);
  for (; it < n; ++it) stage_body(std::false_type{}, it);
#undef W4_READ
#undef W4_READ1
#undef W4_LANDED
#undef W4_MFMA_ROW

}

// acc[i][j][r] = C[m][n]: n = lane & 31 (128 contiguous bytes per half-wave), m from r and the lane half; AM_ / BNW_ = the tile
#define W4_FOR_ACC(AM_, BNW_, STMT)                                                                                    \
  _Pragma("unroll") for (int i = 0; i < (AM_) / 128; ++i) _Pragma("unroll") for (int j = 0; j < (BNW_) / 64; ++j) {    \
    const int nn = n0 + wn * ((BNW_) / 2) + j * 32 + (lane & 31);                                                      \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                   \
      const int m = m0 + wm * ((AM_) / 4) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);                         \
      STMT;                                                                                                            \
    }                                                                                                                  \
  }

// =====================================================================================================
// gemm_tn_w4_k: C[Mo, No] (f32) += A[R, Mo]^T . B[R, No], reduction over tokens split over workgroups, f32 atomics.
// Requires Mo % 384 == 0, No % 192 == 0, R % 32 == 0.
// =====================================================================================================
__global__ __launch_bounds__(W4_WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_tn_w4_k(
    const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ Bm, int64_t ldb, float* __restrict__ C, int64_t ldc,
    int M, int N, int R, int steps_per_split, int spx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // block -> (A-panel unit = (split, m-tile), n-tile): the workgroups that read the same A panel get block ids congruent mod 8,
  // i.e. they sit on one XCD and share the panel in its L2
  const int tiles_n = N / W4_BNW, tiles_m = M / W4_AM;
  int split, mt, nt;
  if (spx > 0) {
    // ALL tiles of one token range on ONE XCD (blockIdx & 7), spx ranges per XCD: every operand byte crosses into exactly one L2
    // and is fetched from HBM once; its other readers (tiles_n sharers of an A panel, tiles_m sharers of a B panel) hit that L2
    const int local = blockIdx.x >> 3, ntile = tiles_m * tiles_n;
    const int sl = local / ntile, tile = local - sl * ntile;
    split = (blockIdx.x & 7) * spx + sl;
    mt = tile / tiles_n;
    nt = tile - mt * tiles_n;
  } else {
    const int grp = blockIdx.x / (8 * tiles_n), rem = blockIdx.x - grp * 8 * tiles_n;
    const int unit = grp * 8 + (rem & 7);
    nt = rem >> 3;
    split = unit / tiles_m;
    mt = unit - split * tiles_m;
  }
  const int m0 = mt * W4_AM, n0 = nt * W4_BNW;
  const int nsteps_total = R / W4_BK;
  const int s_begin = split * steps_per_split;
  int s_end = s_begin + steps_per_split;
  s_end = s_end < nsteps_total ? s_end : nsteps_total;
  if (s_begin >= s_end) return;
  f32x16_t acc[3][3];
  w4_tile<W4_AM, W4_BNW>(A, lda, Bm, ldb, m0, n0, s_begin, s_end, smem, acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  W4_FOR_ACC(W4_AM, W4_BNW, unsafeAtomicAdd(&C[(int64_t)m * ldc + nn], acc[i][j][r]));
}

// =====================================================================================================
// gemm_tn_group_k: up to four problems C_p = A_p[R, Mo_p]^T . B_p[R, No_p] over the same R token rows, S token ranges.
// Workgroup u (XCD-major: the workgroups of one XCD are consecutive u) -> range u / ntile, tile u % ntile, so a token range sits on
// one XCD (S = 8) or on 8 / S neighbouring ones and every operand byte crosses into as few L2s as the grid allows.  The partial
// tile goes to slab[range] (slab-relative offset of the problem + m * No + n) with plain stores.
// =====================================================================================================
struct W4Prob {
  const bf16_t* A;
  const bf16_t* B;
  int64_t off;  // float offset of the problem's [Mo, No] image inside a slab
  int lda, ldb, No, tiles_n, tile0;
  int Mo;       // rows of the image: a last tile that would reach past Mo / No is shifted back to END at the edge (it recomputes a
                // strip of its neighbour: the same products in the same order, i.e. the same bits, stored twice) -- widths that are
                // not whole tiles (640-wide models on 256 x 256 tiles) need no masking and no padded operands
};
struct W4Group {
  W4Prob p[4];
  int nprob, ntile, steps_per_split, nsteps;
  int64_t slab_stride;  // floats between the slabs of two token ranges
  float* slab;
};
// AM x BNW = 384 x 192 (inner widths that are multiples of 384: 3 x 3 MFMA tiles per wave, 128 FLOP per staged byte) or 256 x 256
// (multiples of 256 -- the 512-wide configurations: 2 x 4 tiles per wave, 102 FLOP per staged byte, 32 KiB stages)
template <int AM, int BNW>
__global__ __launch_bounds__(W4_WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_tn_group_k(W4Group g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int u = xcd_remap(blockIdx.x, gridDim.x);
  const int split = u / g.ntile, tile = u - split * g.ntile;
  W4Prob pr = g.p[0];
  if (g.nprob > 1 && tile >= g.p[1].tile0) pr = g.p[1];
  if (g.nprob > 2 && tile >= g.p[2].tile0) pr = g.p[2];
  if (g.nprob > 3 && tile >= g.p[3].tile0) pr = g.p[3];
  const int lt = tile - pr.tile0, mt = lt / pr.tiles_n, nt = lt - mt * pr.tiles_n;
  int m0 = mt * AM, n0 = nt * BNW;
  m0 = m0 + AM > pr.Mo ? pr.Mo - AM : m0;
  n0 = n0 + BNW > pr.No ? pr.No - BNW : n0;
  const int s_begin = split * g.steps_per_split;
  int s_end = s_begin + g.steps_per_split;
  s_end = s_end < g.nsteps ? s_end : g.nsteps;  // (the host sizes the grid so that every range owns at least one stage)
  f32x16_t acc[AM / 128][BNW / 64];
  w4_tile<AM, BNW>(pr.A, pr.lda, pr.B, pr.ldb, m0, n0, s_begin, s_end, smem, acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  float* C = g.slab + (int64_t)split * g.slab_stride + pr.off;
  const int64_t ldc = pr.No;
  W4_FOR_ACC(AM, BNW, C[(int64_t)m * ldc + nn] = acc[i][j][r]);
}

// g_p[i] += ((slab_0[off_p + i] + slab_1[..]) + ...) + slab_{S-1}[..]: a fixed summation order, one pass
struct W4Fold {
  float* dst[4];
  int64_t off[5];  // off[p] = first float of problem p inside a slab, off[nprob] = floats per slab that are in use
  int nprob;
};
__global__ __launch_bounds__(256) void tn_group_fold_k(const float* __restrict__ slab, int64_t stride, int splits, W4Fold f) {
  const int64_t n4 = f.off[f.nprob] >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = i << 2;
    int p = 0;
    if (f.nprob > 1 && e >= f.off[1]) p = 1;
    if (f.nprob > 2 && e >= f.off[2]) p = 2;
    if (f.nprob > 3 && e >= f.off[3]) p = 3;
    float* d = (p == 0 ? f.dst[0] : p == 1 ? f.dst[1] : p == 2 ? f.dst[2] : f.dst[3]) + (e - f.off[p]);
    f32x4_t s = *(const f32x4_t*)(slab + e);
    for (int k = 1; k < splits; ++k) s += *(const f32x4_t*)(slab + k * stride + e);
    *(f32x4_t*)d = *(const f32x4_t*)d + s;
  }
}

static int w4_cus() {
  static DevOnce once;
  return dev_cus(once, [] {
    (void)hipFuncSetAttribute((const void*)gemm_tn_w4_k, hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_tn_group_k<384, 192>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_tn_group_k<256, 256>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              W4_NST * W4_BK * (256 + 256) * 2);
  });
}

// returns 1 when the shape is not this kernel's (the caller keeps its own path), else the launch status
int launch_tn_w4(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N, int64_t R,
                 int max_workgroups, hipStream_t stream) {
  const int n_cu = w4_cus(), min_steps = 32;
  if (M % W4_AM || N % W4_BNW || R % W4_BK || R / W4_BK < 64) return 1;
  const int nsteps = (int)(R / W4_BK);
  const int tiles_m = (int)(M / W4_AM), tiles_n = (int)(N / W4_BNW), ntile = tiles_m * tiles_n;
  const int budget = (max_workgroups > 0 && max_workgroups < n_cu) ? max_workgroups : n_cu;
  int grid, sps, spx = 0;
  if (ntile <= budget / 8) {
    spx = (budget / 8) / ntile;                       // token ranges per XCD
    while (spx > 1 && nsteps / (8 * spx) < min_steps) --spx;
    sps = (nsteps + 8 * spx - 1) / (8 * spx);
    grid = 8 * spx * ntile;
  } else {
    int padded_max = (budget / tiles_n) & ~7;
    if (padded_max < 8) padded_max = 8;
    int splits = padded_max / tiles_m;
    if (splits < 1) splits = 1;
    if (splits > nsteps / min_steps) splits = nsteps / min_steps;
    sps = (nsteps + splits - 1) / splits;
    splits = (nsteps + sps - 1) / sps;
    grid = ((tiles_m * splits + 7) & ~7) * tiles_n;  // surplus units exit at once
  }
  hipLaunchKernelGGL(gemm_tn_w4_k, grid, W4_WAVES * 64, W4_LDS, stream, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, (int)M,
                     (int)N, (int)R, sps, spx);
  if (hipGetLastError() != hipSuccess) return DL_ERR_LAUNCH;
  return DL_OK;
}

extern "C" int dl_gemm_tn_group(const dl_wgrad_t* probs, int n_probs, int64_t R, float* slab, int64_t slab_floats, int max_workgroups,
                                dl_stream_t stream) {
  DL_CHECK_ARG(probs && n_probs >= 1 && n_probs <= 4 && slab && R > 0, "dl_gemm_tn_group: 1..4 problems and a slab");
  const int n_cu = w4_cus();
  if (R % W4_BK || R / W4_BK < 64 || R >= (1ll << 31)) {
    dl_set_error("dl_gemm_tn_group: R=%lld must be a multiple of %d, >= %d and < 2^31", (long long)R, W4_BK, 64 * W4_BK);
    return DL_ERR_UNSUPPORTED;
  }
  W4Group g{};
  W4Fold f{};
  int ntile = 0;
  int64_t total = 0;
  bool t384 = true, t256 = true;  // which tile divides every problem
  for (int i = 0; i < n_probs; ++i) {
    t384 = t384 && probs[i].m_out % 384 == 0 && probs[i].n_in % 192 == 0;
    t256 = t256 && probs[i].m_out % 256 == 0 && probs[i].n_in % 256 == 0;
  }
  if (!t384 && !t256) {  // 256 x 256 tiles with the last tile of a row / column shifted back to the edge (see W4Prob::Mo)
    bool edge = true;
    for (int i = 0; i < n_probs; ++i)
      edge = edge && probs[i].m_out % 8 == 0 && probs[i].n_in % 8 == 0 && probs[i].m_out >= 256 && probs[i].n_in >= 256;
    if (!edge) {
      dl_set_error("dl_gemm_tn_group: the problems are neither whole 384 x 192 tiles nor at least 256 x 256 with widths %% 8 == 0");
      return DL_ERR_UNSUPPORTED;
    }
  }
  const int AMt = t384 ? 384 : 256, BNt = t384 ? 192 : 256;
  for (int i = 0; i < n_probs; ++i) {
    const dl_wgrad_t& q = probs[i];
    DL_CHECK_ARG(q.dy && q.x && q.g && q.m_out > 0 && q.n_in > 0, "dl_gemm_tn_group: null operand in problem %d", i);
    DL_CHECK_ARG(q.ld_dy % 8 == 0 && q.ld_x % 8 == 0 && q.ld_dy >= q.m_out && q.ld_x >= q.n_in && q.ld_dy < (1ll << 31) &&
                     q.ld_x < (1ll << 31),
                 "dl_gemm_tn_group: leading dimensions of problem %d", i);
    DL_CHECK_ARG((((uintptr_t)q.dy | (uintptr_t)q.x | (uintptr_t)q.g) & 15) == 0, "dl_gemm_tn_group: 16-byte alignment");
    W4Prob& p = g.p[i];
    p.A = (const bf16_t*)q.dy;
    p.B = (const bf16_t*)q.x;
    p.lda = (int)q.ld_dy;
    p.ldb = (int)q.ld_x;
    p.No = (int)q.n_in;
    p.Mo = (int)q.m_out;
    p.tiles_n = (int)((q.n_in + BNt - 1) / BNt);
    p.tile0 = ntile;
    p.off = total;
    f.dst[i] = q.g;
    f.off[i] = total;
    ntile += (int)((q.m_out + AMt - 1) / AMt) * p.tiles_n;
    total += q.m_out * q.n_in;
  }
  f.off[n_probs] = total;
  f.nprob = g.nprob = n_probs;
  DL_CHECK_ARG((((uintptr_t)slab) & 15) == 0 && slab_floats >= total, "dl_gemm_tn_group: slab of %lld floats < one partial image (%lld)",
               (long long)slab_floats, (long long)total);
  const int nsteps = (int)(R / W4_BK);
  const int budget = (max_workgroups > 0 && max_workgroups < n_cu) ? max_workgroups : n_cu;
  int splits = budget / ntile;
  if (splits < 1) splits = 1;
  if (splits > nsteps / 32) splits = nsteps / 32;
  if ((int64_t)splits * total > slab_floats) splits = (int)(slab_floats / total);
  const int sps = (nsteps + splits - 1) / splits;
  splits = (nsteps + sps - 1) / sps;  // every range owns at least one stage
  g.ntile = ntile;
  g.steps_per_split = sps;
  g.nsteps = nsteps;
  g.slab_stride = total;
  g.slab = slab;
  if (t384)
    hipLaunchKernelGGL((gemm_tn_group_k<384, 192>), splits * ntile, W4_WAVES * 64, W4_LDS, (hipStream_t)stream, g);
  else
    hipLaunchKernelGGL((gemm_tn_group_k<256, 256>), splits * ntile, W4_WAVES * 64, W4_NST * W4_BK * (256 + 256) * 2, (hipStream_t)stream, g);
  int64_t fg = ((total >> 2) + 255) / 256;
  if (fg > 1024) fg = 1024;
#ifndef W4_LAB_NO_FOLD  // LAB: what the step costs without the per-block fold launch (WRONG gradients)
  hipLaunchKernelGGL(tn_group_fold_k, (int)fg, 256, 0, (hipStream_t)stream, slab, total, splits, f);
#endif
  DL_LAUNCH_CHECK();
  return DL_OK;
}
