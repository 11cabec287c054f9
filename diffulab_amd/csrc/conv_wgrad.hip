// Tap-reusing weight gradient of the 3x3 / pad 1 convolution (gfx950):  g[(tap, ci), co] += sum_p x[p + shift(tap), ci] * dY[p, co]
// over NHWC rows (reference: the autograd of nn.Conv2d in diffulab/networks/denoisers/unet.py:187,208).
//
// Why a new kernel: the implicit-GEMM forms of gemm.hip (gemm_tn_big_k<true>, gemm_tn_k<true>) treat the nine taps as nine
// independent column blocks of an im2col matrix: a 384 x 128 output tile stages 64 KiB of operands per 6.3 MFLOP (96 FLOP/B),
// the same x rows once per tap, and the loop runs at what a CU pulls out of L2 (25 steps x 64 KiB = 60 of the 90 us of the
// 16 x 16 / 256 -> 256 layer at B = 128; 473 TFLOP/s over the UNet's layers against 800 for the forward convolution).  Here
//   * an output tile is ALL NINE TAPS of a 64-channel chunk x 128 output channels (576 x 128 f32, 144 accumulator registers per lane);
//   * a 64-pixel step is whole image rows (or whole images), so the nine taps read nine SHIFTED windows of the same pixels: the
//     step's rows plus a one-pixel border -- the halo, (nrow + 2) x (W + 2) rows per image segment, 100 - 144 rows of 128 bytes --
//     are staged ONCE (zero line outside the image) beside the 64 x 128 dY rows: 29 - 34 KiB per 9.4 MFLOP (280 - 320 FLOP/B);
//   * fragments come out of ds_read_b64_tr_b16 (both operands are contraction-major in memory); the halo row of (pixel, tap) is
//     lane part + compile-time constant, and the 16-byte-slot swizzle is arranged so that the constant lands in the instruction's
//     offset field: no address arithmetic per MFMA (below, "swizzle");
//   * four-slot LDS ring (<= 136 KiB), stages in flight under counted vmcnt waits, ONE barrier per 36 MFMAs of a wave -- in mid-step,
//     where it admits the NEXT stage: the fragment read stream runs five fragments ahead of the MFMAs (counted lgkmcnt waits) and
//     does not stop at a step boundary;
//   * the pixel range is split over workgroups (one per CU at most, all tiles of a range on one XCD).  Two epilogues:
//     dl_conv3x3_wgrad_tn -- the ranges meet in g through f32 atomics (its contract since round 2); dl_conv3x3_wgrad_tn_parts --
//     every range STORES its image, the batched fold adds them in image order.  Measured (profiles/r06_p_*, r06_w_*): the loop runs
//     at 1.19-1.3 PFLOP/s; the atomics of the first form are 56 % of a launch (they drain at 1.45 TB/s, plain rows at ~6), which is
//     why the engine takes the second.
// 8 waves as 2 (32-channel halves of the chunk) x 4 (32 output channels): a wave owns one 32 x 32 MFMA tile per tap.
//
// Swizzle.  Halo image: 128-byte rows; 16-byte slot s of row rho is stored at slot s ^ (key(rho) << 2), key(rho) = bit 1 of rho, so
// the four consecutive rows of one transposing read fall into four distinct 64-byte bank groups.  A fragment row is rho = a + b with
// a = the lane's pixel inside the 16-pixel block (+ 2 per image row it crosses) and b = block origin + tap displacement, a
// compile-time constant: key(a + b) = a1 ^ b1 ^ (a0 & b0), i.e. one of FOUR per-lane byte offsets (by b0, b1) plus b * 128 in the
// offset field.
#include <type_traits>

#include "common.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

namespace {  // helpers only: the kernel itself keeps external linkage so that profilers print its name

__device__ __forceinline__ void glds16_cw(const void* gsrc, void* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void_t*)gsrc, (lds_void_t*)lds_dst_wave_base, 16, 0, 0);
}
// inline assembly on purpose (gemm_w4.hip): a compiler-visible LDS read behind a direct-to-LDS DMA gets s_waitcnt vmcnt(0)
template <int OFF>
__device__ __forceinline__ s16x4_t tr16_cw(unsigned addr) {
  s16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt_cw() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lgkm_cw() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

#define CW_THREADS 512
#define CW_NST 4
#define CW_CK 64    // input channels per tile
#define CW_CN 128   // output channels per tile

// geometry of a 64-pixel step for W = 2^LW and NROW image rows per segment (NROW = H when a step holds whole images)
template <int LW, int NROW>
struct CwCfg {
  static constexpr int W = 1 << LW, W2 = W + 2;
  static constexpr int NSEG = 64 / (NROW * W);
  static constexpr int SEG_ROWS = (NROW + 2) * W2;
  static constexpr int HP = NSEG * SEG_ROWS;      // halo rows
  static constexpr int XCH = (HP + 7) / 8;        // 1 KiB DMA chunks of the halo image
  static constexpr int NC = XCH + 16;             // + the 64 x 128 dY rows
  static constexpr int STAGE = NC * 1024;
  static constexpr int CH = (NC + 7) / 8, CHMIN = NC / 8;  // chunks per wave and stage: the first NC % 8 waves own one more
  static constexpr bool HALF_IMM = LW >= 3;       // the second four pixels of a lane's eight are the same image row: + 4 halo rows
  static_assert(NSEG * NROW * W == 64, "a step is 64 pixels");
  static_assert(CW_NST * STAGE <= 160 * 1024, "ring fits the LDS");
  // halo row of the first pixel of 16-pixel block kk
  static constexpr int urow(int kk) {
    const int p = kk * 16, seg = p / (NROW * W), rem = p % (NROW * W);
    return seg * SEG_ROWS + (rem / W + 1) * W2 + (rem % W) + 1;
  }
  static constexpr int delta(int tap) { return (tap / 3 - 1) * W2 + (tap % 3 - 1); }
  static constexpr int brow(int kk, int tap) { return urow(kk) + delta(tap); }
};

// Read stream.  Item s = 9 kk + tap of a step is one MFMA: acc[tap] += A(s) . B(kk).  Fragments (two transposing reads each) return
// in order.  The stream runs CW_D items ahead of the MFMAs and does not stop at a step boundary: item s issues A(s + D) -- of the
// NEXT step's stage once s + D >= 36 -- and, at tap CW_BT, B(kk + 1) (kk = 3: the next step's B(0)), which therefore sits just in
// front of the first A fragment of its block.  Before item s's MFMA at most `cw_younger_reads(s)` reads may be outstanding: the D
// younger A fragments and the B fragments issued since A(s).  The ring of D + 1 A registers divides 36, so the register a fragment
// lands in does not depend on the step.
#define CW_D 5
#define CW_BT (8 - CW_D)
#define CW_MB 13  // item in front of which the NEXT stage is waited for (every wave's share) and the stage three steps ahead is requested
constexpr int cw_mod9(int u) { return ((u % 9) + 9) % 9; }
constexpr int cw_younger_reads(int s) {
  int nb = 0;
  for (int u = s - CW_D + 1; u <= s; ++u) nb += cw_mod9(u) == CW_BT ? 1 : 0;
  return 2 * (CW_D + nb);
}
static_assert(36 % (CW_D + 1) == 0 && CW_BT >= 0 && 2 * (CW_D + 1) <= 15, "read stream");
static_assert(CW_MB < 27 + CW_BT && CW_MB < 36 - CW_D, "the next stage is awaited before its first fragment is requested");

}  // namespace

struct CwArgs {
  const bf16_t* X;
  const bf16_t* dY;
  float* G;
  const bf16_t* zero;  // >= 16 bytes of zeros
  int64_t ldx, ldy, ldg, npix, part_stride;
  int H, Ci, tiles_n, ntile, nsteps, sps, xcd_major;
};

template <int LW, int NROW>
__global__ __launch_bounds__(CW_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_wgrad_halo_k(CwArgs a) {
  using Cfg = CwCfg<LW, NROW>;
  constexpr int W = Cfg::W, W2 = Cfg::W2, XCH = Cfg::XCH, NC = Cfg::NC, STAGE = Cfg::STAGE, CH = Cfg::CH, CHMIN = Cfg::CHMIN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_t*)smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wi = wave >> 2, wj = wave & 3;

  // ---- workgroup -> (pixel range, tile).  xcd_major: all tiles of one range on ONE XCD (blockIdx & 7), so every operand byte crosses
  // into one L2; otherwise (fewer than 8 ranges) tile-minor
  int split, tile;
  if (a.xcd_major > 0) {
    const int local = blockIdx.x >> 3, sl = local / a.ntile;
    tile = local - sl * a.ntile;
    split = (blockIdx.x & 7) * a.xcd_major + sl;
  } else {
    split = blockIdx.x / a.ntile;
    tile = blockIdx.x - split * a.ntile;
  }
  const int tm = tile / a.tiles_n, tn = tile - tm * a.tiles_n;
  const int ci0 = tm * CW_CK, n0 = tn * CW_CN;
  const int s_begin = split * a.sps;
  int s_end = s_begin + a.sps;
  s_end = s_end < a.nsteps ? s_end : a.nsteps;
  if (s_begin >= s_end) return;
  const int n = s_end - s_begin;

  // ---- DMA tables: chunk c = i * 8 + wave of a stage; c < XCH: eight halo rows, else four dY rows
  int d_off[CH], d_hy[CH], d_seg[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = i * 8 + wave;
    d_off[i] = 0, d_hy[i] = -(1 << 20), d_seg[i] = 0;
    if (c < XCH) {
      const int rho = c * 8 + (lane >> 3);
      const int q = (lane & 7) ^ (((rho >> 1) & 1) << 2);
      if (rho < Cfg::HP) {
        const int seg = rho / Cfg::SEG_ROWS, rem = rho - seg * Cfg::SEG_ROWS;
        const int hy = rem / W2, px = rem - hy * W2 - 1;
        if (px >= 0 && px < W) {
          d_hy[i] = hy - 1;
          d_seg[i] = seg * a.H * W;
          d_off[i] = (int)((d_seg[i] + (hy - 1) * W + px) * a.ldx) + q * 8;
        }
      }
    } else if (c < NC) {
      const int o = (c - XCH) * 1024 + lane * 16;
      const int r = o >> 8, s = (o & 255) >> 4;
      d_hy[i] = 0;
      d_off[i] = (int)(r * a.ldy) + ((s ^ ((r & 3) << 2)) << 3);
    }
  }
  auto issue = [&](int st) {  // operand stage of step s_begin + st into ring slot st % NST
    char* base = smem + (st & (CW_NST - 1)) * STAGE;
    const int64_t p0 = (int64_t)(s_begin + st) * 64;
    const int py0 = Cfg::NSEG == 1 ? (int)(p0 >> LW) & (a.H - 1) : 0;
    const bf16_t* xb = a.X + p0 * a.ldx + ci0;
    const bf16_t* yb = a.dY + p0 * a.ldy + n0;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = i * 8 + wave;
      if (c < NC) {
        const bool ok = (unsigned)(py0 + d_hy[i]) < (unsigned)a.H && p0 + d_seg[i] < a.npix;
        const bf16_t* src = (c < XCH ? xb : yb) + d_off[i];
        glds16_cw(ok ? src : a.zero, base + c * 1024);
      }
    }
  };

  // ---- fragment addresses
  const int li = lane & 15, g = lane >> 4;
  const int rl = (g >> 1) * 8 + (li >> 2);
  auto lane_off = [&](int r, int v) -> unsigned {  // byte offset of this lane's read for in-block pixel r, constant-row parity class v
    const int ar = r + 2 * (r >> LW);
    const int cb = (wi * 32 + (g & 1) * 16 + (li & 3) * 4) * 2;
    const int key = ((ar >> 1) & 1) ^ ((v >> 1) & 1) ^ (ar & v & 1);
    return (unsigned)(ar * 128 + (cb ^ (key << 6)));
  };
  unsigned lx[4], lx1[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    lx[v] = lane_off(rl, v);
    lx1[v] = lane_off(rl + 4, v);  // (used when the second half is not + 4 rows: W = 4)
  }
  unsigned boff;
  {
    const int col = wj * 32 + (g & 1) * 16 + (li & 3) * 4;
    boff = XCH * 1024 + rl * 256 + (((col >> 3) ^ ((rl & 3) << 2)) << 4) + (col & 7) * 2;
  }

  f32x16_t acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  union Frag {
    s16x4_t h[2];
    bf16x8_t v;
  };

  Frag aq[CW_D + 1], bq[2];
  // fragment s (0 .. 35 + D: beyond 35 it belongs to the next step, stage base sn) / B block kk (4 = the next step's first)
  auto read_a = [&](auto S_, unsigned sb, unsigned sn) __attribute__((always_inline)) {
    constexpr int s = decltype(S_)::value, sm = s % 36, kk = sm / 9, tap = sm % 9;
    constexpr int b = Cfg::brow(kk, tap), v = b & 3;
    const unsigned base = s < 36 ? sb : sn;
    aq[s % (CW_D + 1)].h[0] = tr16_cw<b * 128>(base + lx[v]);
    if constexpr (Cfg::HALF_IMM) aq[s % (CW_D + 1)].h[1] = tr16_cw<b * 128 + 512>(base + lx[v]);
    else aq[s % (CW_D + 1)].h[1] = tr16_cw<b * 128>(base + lx1[v]);
  };
  auto read_b = [&](auto K_, unsigned sb, unsigned sn) __attribute__((always_inline)) {
    constexpr int kk = decltype(K_)::value;
    const unsigned base = (kk < 4 ? sb : sn) + boff;
    bq[kk & 1].h[0] = tr16_cw<(kk & 3) * 4096>(base);
    bq[kk & 1].h[1] = tr16_cw<(kk & 3) * 4096 + 1024>(base);
  };
  // one step: 36 MFMAs of this wave on stage `it`; FULL: the stage three steps ahead exists (straight-line body)
  auto step = [&](auto full_c, int it) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_c)::value;
    const unsigned sb = lds0 + (it & (CW_NST - 1)) * STAGE, sn = lds0 + ((it + 1) & (CW_NST - 1)) * STAGE;
    static_for<0, 36>([&](auto S_) __attribute__((always_inline)) {
      constexpr int s = decltype(S_)::value, kk = s / 9, tap = s % 9;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (s == CW_MB) {
        // stage it + 1 has landed (the one stage requested after it may stay in flight) for every wave, and every wave is past step
        // it - 1: its slot takes stage it + 3
        if (FULL) {
          wait_vmcnt_cw<CHMIN>();
          __builtin_amdgcn_s_barrier();
          issue(it + CW_NST - 1);
        } else if (it + 1 < n) {
          wait_vmcnt_cw<0>();
          __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (tap == CW_BT) read_b(std::integral_constant<int, kk + 1>{}, sb, sn);
      read_a(std::integral_constant<int, s + CW_D>{}, sb, sn);
      wait_lgkm_cw<cw_younger_reads(s)>();
      asm volatile("" : "+v"(aq[s % (CW_D + 1)].v), "+v"(bq[kk & 1].v));  // no MFMA on these registers above the wait
      acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[s % (CW_D + 1)].v, bq[kk & 1].v, acc[tap], 0, 0, 0);
    });
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- ring: stages 0 .. 2 requested, stage 0 awaited, the read stream primed with B(0), A(0 .. D - 1)
  {
    const int pre = n < CW_NST - 1 ? n : CW_NST - 1;
    for (int st = 0; st < pre; ++st) issue(st);
    if (n >= CW_NST - 1) wait_vmcnt_cw<(CW_NST - 2) * CHMIN>();
    else wait_vmcnt_cw<0>();
    __builtin_amdgcn_s_barrier();
    read_b(std::integral_constant<int, 0>{}, lds0, lds0);
    static_for<0, CW_D>([&](auto S_) __attribute__((always_inline)) { read_a(S_, lds0, lds0); });
  }
  int it = 0;
  for (; it + CW_NST - 1 < n; ++it) step(std::true_type{}, it);
  for (; it < n; ++it) step(std::false_type{}, it);
  // (the stream ran D fragments into a stage that does not exist: stale LDS, never used -- but they must land before their registers
  //  are given to the epilogue)
  wait_lgkm_cw<0>();
  asm volatile("" ::: "memory");

  // ---- acc[tap][r] = g[(tap, ci0 + wi * 32 + m), n0 + wj * 32 + (lane & 31)], m from r and the lane half
  float* gp = a.G + (a.part_stride > 0 ? (int64_t)split * a.part_stride : 0) + n0 + wj * 32 + (lane & 31);
  if (a.part_stride < 0) {  // LAB (dl_lab_set_wgrad_halo(2)): the loop alone -- one word per lane keeps the accumulators alive
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) v += acc[t][r];
    if (v == 12345.678f) *gp = v;
    return;
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int64_t row0 = (int64_t)t * a.Ci + ci0 + wi * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float* p = gp + (row0 + (r & 3) + 8 * (r >> 2)) * a.ldg;
      if (a.part_stride > 0) *p = acc[t][r];
      else unsafeAtomicAdd(p, acc[t][r]);
    }
  }
}

template <int LW, int NROW>
static int launch_cw(const CwArgs& a, int grid, hipStream_t stream) {
  using Cfg = CwCfg<LW, NROW>;
  static DevOnce once;
  (void)dev_cus(once, [] {
    (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_halo_k<LW, NROW>, hipFuncAttributeMaxDynamicSharedMemorySize, CW_NST * Cfg::STAGE);
  });
  hipLaunchKernelGGL((conv3x3_wgrad_halo_k<LW, NROW>), grid, CW_THREADS, CW_NST * Cfg::STAGE, stream, a);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

static int g_wgrad_halo = 1;  // LAB switch (not in the header): 0 = the implicit-GEMM kernels of gemm.hip everywhere
extern "C" __attribute__((visibility("default"))) void dl_lab_set_wgrad_halo(int on) { g_wgrad_halo = on; }

// conv_wgrad_halo_splits: the pixel ranges (= partial images of the store form) this kernel uses for a shape, 0 = shape not taken
// (the caller falls back to the implicit-GEMM kernels).  launch_conv_wgrad_halo: 1 = not taken; DL_OK / error otherwise; part_stride
// > 0 selects the store form (the caller has sized its images by conv_wgrad_halo_splits).
int conv_wgrad_halo_splits(int64_t H, int64_t W, int64_t Ci, int64_t Co, int64_t R, int max_workgroups, int n_cu, int* sps_out) {
  if (!g_wgrad_halo || Ci % CW_CK != 0 || Co % CW_CN != 0 || R % 64 != 0 || (H & (H - 1)) != 0) return 0;
  const bool geom = (W == 32 && H % 2 == 0) || (W == 16 && H % 4 == 0) || (W == 8 && H == 8) || (W == 4 && H == 4);
  if (!geom) return 0;
  const int nsteps = (int)(R / 64);
  const int64_t ntile = (Ci / CW_CK) * (Co / CW_CN);
  const int budget = (max_workgroups > 0 && max_workgroups < n_cu) ? max_workgroups : n_cu;
  int splits = (int)(budget / ntile);
  if (splits > nsteps / 8) splits = nsteps / 8;  // a range runs at least twice the ring depth
  if (splits < 1) splits = 1;
  if (splits >= 8) splits &= ~7;
  const int sps = (nsteps + splits - 1) / splits;
  splits = (nsteps + sps - 1) / sps;
  if (sps_out) *sps_out = sps;
  return splits;
}
int launch_conv_wgrad_halo(const void* x, int64_t ldx, int64_t Bn, int64_t H, int64_t W, int64_t Ci, const void* dY, int64_t ldy, int64_t R,
                           int64_t Co, float* g, int64_t ldg, int64_t part_stride, const void* zero, int max_workgroups, int n_cu,
                           hipStream_t stream) {
  int sps = 0;
  const int splits = conv_wgrad_halo_splits(H, W, Ci, Co, R, max_workgroups, n_cu, &sps);
  if (splits <= 0) return 1;
  DL_CHECK_ARG(ldx < (1 << 22) && ldy < (1 << 22), "dl_conv3x3_wgrad_tn: row strides beyond 4 M elements");  // per-lane source offsets are 32-bit
  CwArgs a{};
  a.X = (const bf16_t*)x, a.dY = (const bf16_t*)dY, a.G = g, a.zero = (const bf16_t*)zero;
  a.ldx = ldx, a.ldy = ldy, a.ldg = ldg, a.npix = Bn * H * W, a.part_stride = g_wgrad_halo == 2 ? -1 : part_stride;
  a.H = (int)H, a.Ci = (int)Ci, a.tiles_n = (int)(Co / CW_CN), a.ntile = (int)((Ci / CW_CK) * (Co / CW_CN));
  a.nsteps = (int)(R / 64), a.sps = sps;
  int grid;
  if (splits % 8 == 0) {
    a.xcd_major = splits / 8;
    grid = splits * a.ntile;
  } else {
    a.xcd_major = 0;
    grid = splits * a.ntile;
  }
  if (W == 32) return launch_cw<5, 2>(a, grid, stream);
  if (W == 16) return launch_cw<4, 4>(a, grid, stream);
  if (W == 8) return launch_cw<3, 8>(a, grid, stream);
  return launch_cw<2, 4>(a, grid, stream);
}
