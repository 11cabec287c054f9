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
// WAVES = 4 (one wave per SIMD, 192 x 96 per wave, 288 accumulator registers) is kept for the record: the compiler holds MFMA
// accumulators in the 256 AGPRs only and shuttles the remaining 32 through v_accvgpr moves every iteration (275 us vs 176 us on the
// MLP-up weight gradient); DL_GEMM_TN_W4_WAVES=4 selects it.
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
// gemm_tn_w4_k: C[Mo, No] (f32) += A[R, Mo]^T . B[R, No], reduction over tokens split over workgroups, f32 atomics.
// Tile 384 (m) x BNW (n), BNW = 192 (256 compiles for WAVES = 8 only on paper: 278 registers); WAVES/2 (m) x 2 (n) waves, each
// 768/WAVES x BNW/2 = (24/WAVES) x (BNW/64) MFMA 32x32x16 tiles.
// LDS images are the row-major [32 r][384 | BNW] slabs written by the DMA; fragments come out of ds_read_b64_tr_b16.  16-byte slot
// swizzle per image row so that the 4 rows of one transposing read fall into 4 distinct 64-byte bank groups:
//   768- and 512-byte rows (multiples of the 256-byte bank period): slot ^= (r & 3) << 2
//   384-byte rows (r and r+2 alias):                                slot ^= ((r >> 1) & 1) << 2
// Requires Mo % 384 == 0, No % BNW == 0, R % 32 == 0.
// =====================================================================================================
// PROBE (tuning builds, DL_GEMM_TN_W4_PROBE): 2 = no MFMAs, 4 = no DMA after the ring fill, 8 = no fragment reads
template <int BNW, int WAVES, int PROBE = 0>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(WAVES / 4, WAVES / 4))) void gemm_tn_w4_k(
    const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ Bm, int64_t ldb, float* __restrict__ C, int64_t ldc,
    int M, int N, int R, int steps_per_split, int spx) {
  constexpr int WM = WAVES / 2;                      // waves along m (2 along n)
  constexpr int IM = W4_AM / WM / 32, JN = BNW / 64;  // MFMA tiles per wave
  constexpr int PA = W4_AM * 2, PB = BNW * 2;        // image row pitches (bytes)
  constexpr int A_BYTES = W4_BK * PA;                // 24 KiB
  constexpr int STAGE = W4_BK * (PA + PB);           // 36 | 40 KiB
  constexpr int ACH = A_BYTES / 1024;                // 1 KiB DMA chunks of the A image
  constexpr int NCH = STAGE / 1024;                  // 36 | 40
  constexpr int CH = (NCH + WAVES - 1) / WAVES;      // chunks per wave and stage (the last waves may own one less)
  constexpr int CHMIN = NCH / WAVES;                 // what the counted waits assume is outstanding per younger stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;  // LDS byte address of the ring (asm reads take addresses)
  // block -> (A-panel unit = (split, m-tile), n-tile): the workgroups that read the same A panel get block ids congruent mod 8,
  // i.e. they sit on one XCD and share the panel in its L2
  const int tiles_n = N / BNW, tiles_m = M / W4_AM;
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
  const int m0 = mt * W4_AM, n0 = nt * BNW;
  const int nsteps_total = R / W4_BK;
  const int s_begin = split * steps_per_split;
  int s_end = s_begin + steps_per_split;
  s_end = s_end < nsteps_total ? s_end : nsteps_total;
  if (s_begin >= s_end) return;
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
    if ((PROBE & 4) && st >= W4_NST) return;
    glds16_w4(cptr[i] + (int64_t)st * cstep[i], smem + (st & (W4_NST - 1)) * STAGE + clds[i]);
  };

  // ---- per-lane byte offsets of the transposing fragment reads inside a stage image (sub-step and half add constants)
  const int li = lane & 15, g = lane >> 4;
  const int rl = (g >> 1) * 8 + (li >> 2);
  int aoff[IM], boff[JN];
#pragma unroll
  for (int i = 0; i < IM; ++i) {
    const int col = wm * (W4_AM / WM) + i * 32 + (g & 1) * 16 + (li & 3) * 4;
    aoff[i] = rl * PA + (((col >> 3) ^ ((rl & 3) << 2)) << 4) + (col & 7) * 2;
  }
#pragma unroll
  for (int j = 0; j < JN; ++j) {
    const int col = wn * (BNW / 2) + j * 32 + (g & 1) * 16 + (li & 3) * 4;
    const int sw = BNW == 256 ? ((rl & 3) << 2) : (((rl >> 1) & 1) << 2);
    boff[j] = A_BYTES + rl * PB + (((col >> 3) ^ sw) << 4) + (col & 7) * 2;
  }

  f32x16_t acc[IM][JN];
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
    if (PROBE & 8) {                                                                 \
    } else if ((F) < IM) {                                                                  \
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
    if (PROBE & 2) {                                                    \
      asm volatile("" ::"v"(af[BUF][I]), "v"(bfr[BUF][j]));             \
    } else {                                                            \
      acc[I][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[BUF][I], bfr[BUF][j], acc[I][j], 0, 0, 0); \
    }                                                                   \
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
    if (FULL && !(PROBE & 4)) wait_vmcnt_w4<(W4_NST - 2) * CHMIN>();  // stage it+1 landed; it+2, it+3 may stay in flight
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

  // ---- f32 atomics: acc[i][j][r] = C[m][n], n = lane & 31 (128 contiguous bytes per half-wave), m from r and the lane half
#pragma unroll
  for (int i = 0; i < IM; ++i)
#pragma unroll
    for (int j = 0; j < JN; ++j) {
      const int nn = n0 + wn * (BNW / 2) + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (W4_AM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        unsafeAtomicAdd(&C[(int64_t)m * ldc + nn], acc[i][j][r]);
      }
    }
}

// returns 1 when the shape is not this kernel's (the caller keeps its own path), else the launch status
int launch_tn_w4(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N, int64_t R,
                 int max_workgroups, hipStream_t stream) {
  static int n_cu = 0, waves = 8, xcd_map = 1, min_steps = 32;
  if (n_cu == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (n_cu <= 0) n_cu = 256;
    (void)hipFuncSetAttribute((const void*)gemm_tn_w4_k<192, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_NST * W4_BK * (384 + 192) * 2);
    (void)hipFuncSetAttribute((const void*)gemm_tn_w4_k<192, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_NST * W4_BK * (384 + 192) * 2);
    const char* e = getenv("DL_GEMM_TN_W4_WAVES");
    if (e) waves = atoi(e);
    e = getenv("DL_GEMM_TN_W4_XCD");  // 0: A-panel sharers on one XCD, token ranges spread (the mapping of gemm_tn_big_k)
    if (e) xcd_map = atoi(e);
    e = getenv("DL_GEMM_TN_W4_MINSTEPS");
    if (e) min_steps = atoi(e);
  }
  const int BNW = 192;
  if (M % W4_AM || N % BNW || R % W4_BK || R / W4_BK < 64) return 1;
  const int nsteps = (int)(R / W4_BK);
  const int tiles_m = (int)(M / W4_AM), tiles_n = (int)(N / BNW), ntile = tiles_m * tiles_n;
  const int budget = (max_workgroups > 0 && max_workgroups < n_cu) ? max_workgroups : n_cu;
  int grid, sps, spx = 0;
  if (xcd_map && ntile <= budget / 8) {
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
  static int probe = -1;
  if (probe < 0) {
    const char* e = getenv("DL_GEMM_TN_W4_PROBE");
    probe = e ? atoi(e) : 0;
    (void)hipFuncSetAttribute((const void*)gemm_tn_w4_k<192, 8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_NST * W4_BK * (384 + 192) * 2);
    (void)hipFuncSetAttribute((const void*)gemm_tn_w4_k<192, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_NST * W4_BK * (384 + 192) * 2);
    (void)hipFuncSetAttribute((const void*)gemm_tn_w4_k<192, 8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_NST * W4_BK * (384 + 192) * 2);
    (void)hipFuncSetAttribute((const void*)gemm_tn_w4_k<192, 8, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_NST * W4_BK * (384 + 192) * 2);
    (void)hipFuncSetAttribute((const void*)gemm_tn_w4_k<192, 8, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_NST * W4_BK * (384 + 192) * 2);
    (void)hipFuncSetAttribute((const void*)gemm_tn_w4_k<192, 8, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_NST * W4_BK * (384 + 192) * 2);
  }
#define W4_PROBE_GO(P)                                                                                                       \
  hipLaunchKernelGGL((gemm_tn_w4_k<192, 8, P>), grid, 512, W4_NST * W4_BK * (384 + 192) * 2, stream, (const bf16_t*)A, lda, \
                     (const bf16_t*)B, ldb, C, ldc, (int)M, (int)N, (int)R, sps, spx)
  if (probe == 2) W4_PROBE_GO(2);
  else if (probe == 4) W4_PROBE_GO(4);
  else if (probe == 8) W4_PROBE_GO(8);
  else if (probe == 10) W4_PROBE_GO(10);
  else if (probe == 12) W4_PROBE_GO(12);
  else if (probe == 6) W4_PROBE_GO(6);
  else if (waves == 4)
    hipLaunchKernelGGL((gemm_tn_w4_k<192, 4>), grid, 256, W4_NST * W4_BK * (384 + 192) * 2, stream, (const bf16_t*)A, lda,
                       (const bf16_t*)B, ldb, C, ldc, (int)M, (int)N, (int)R, sps, spx);
  else
    hipLaunchKernelGGL((gemm_tn_w4_k<192, 8>), grid, 512, W4_NST * W4_BK * (384 + 192) * 2, stream, (const bf16_t*)A, lda,
                       (const bf16_t*)B, ldb, C, ldc, (int)M, (int)N, (int)R, sps, spx);
  if (hipGetLastError() != hipSuccess) return DL_ERR_LAUNCH;
  return DL_OK;
}
