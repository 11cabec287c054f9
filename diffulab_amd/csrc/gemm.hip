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
#include <stdlib.h>

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
  // fused SwiGLU forward (EPI 2): `aux` is h (written), F = hidden width
  bf16_t* aux;
  int64_t ld_aux;
  int F;
  // persistent big-tile kernel only: start-up skew (in units of ~0.5 us per k-step of a tile) between groups of workgroups,
  // so that the store bursts of their tile epilogues do not hit HBM in lockstep
  int stagger;
  // 128x128 kernel with a split contraction (blockIdx.y): > 0 = every split stores its partial tile with plain stores at
  // C + split * part_stride (floats; the caller folds the images in a fixed order: bit-reproducible); 0 = f32 atomics into C
  int64_t part_stride;
  // EPI 3 (384-wide persistent tiles): plain bf16 stores + per-row sums of squares of the ROUNDED outputs of the first `ssq_tiles`
  // column tiles, added into ssq[m * ssq_tiles + tile] (f32, zeroed by the caller): the statistics of the QK-RMSNorm (nn.py:427-431)
  // leave with the qkv GEMM.  Two addends per element (the tile's two column waves) on a zeroed word: order-independent.
  float* ssq;
  int ssq_tiles;
  // 128x128 kernel only (dl_gemm_nt_pair): a SECOND, independent product in the same launch -- workgroups from `tiles0` on compute
  // C2 = A2 B2^T (+ bias2, + resid2) with the same activation / output type; 0 = one product.  Two under-filled launches (the q and kv
  // projections of a UNet AttentionBlock at the low-resolution levels: 64 + 128 tiles on 256 CUs) become one that fills the chip.
  const bf16_t* A2;
  const bf16_t* B2;
  void* C2;
  const float* bias2;
  const bf16_t* resid2;
  int64_t lda2, ldb2, ldc2, ldr2;
  int M2, N2, K2, tiles0;
};

// implicit-GEMM view of a 3x3 / pad 1 convolution over NHWC rows: the A operand "cols[p, (tap, ci)]" is never materialised,
// the direct-to-LDS loads gather x[p + shift(tap), ci] and out-of-image taps read a zero line instead (unet.py:187,208)
struct ConvGeom {
  int H, W, Ci;        // Ci % 64 == 0 (NT: a 64-deep k-step lies inside one tap) / Ci % 128 == 0 (TN: a 128-wide column tile does)
  int64_t ldx;         // row stride of x (elements)
  int64_t npix;        // B*H*W real rows (rows beyond it are zero)
  const bf16_t* zero;  // >= 16 bytes of zeros, 16-byte aligned (caller-owned)
  int lw, lh;          // log2(W), log2(H) when both are powers of two (pixel coordinates by shift/mask), else -1

  __device__ __forceinline__ void pixel(int64_t p, int& py, int& px) const {
    if (lw >= 0) {
      px = (int)p & (W - 1);
      py = (int)(p >> lw) & (H - 1);
    } else {
      px = (int)(p % W);
      py = (int)((p / W) % H);
    }
  }
};
static ConvGeom make_conv_geom(int64_t H, int64_t W, int64_t Ci, int64_t ldx, int64_t npix, const void* zero) {
  auto lg = [](int64_t v) {
    int l = 0;
    while ((1ll << l) < v) ++l;
    return (1ll << l) == v ? l : -1;
  };
  ConvGeom cg{(int)H, (int)W, (int)Ci, ldx, npix, (const bf16_t*)zero, lg(W), lg(H)};
  if (cg.lw < 0 || cg.lh < 0) cg.lw = cg.lh = -1;
  return cg;
}

// NSLOT = 2: two 32 KiB operand stages, the next one requested while the current one is consumed (two workgroups per CU cover each
// other's waits).  NSLOT = 4 (round 6): launches of AT MOST ONE workgroup per CU -- the mid-size linears of the low-resolution UNet
// levels and of small-batch DiT steps: 6 GFLOP problems, 256 tiles or fewer -- have nobody to cover a wait, and one stage of
// prefetch is ~0.25 us of MFMAs against ~1.5 us of memory latency: every k-step then costs the latency.  Alone on its CU the
// workgroup can have the LDS: a four-slot ring (128 KiB), three stages in flight, counted vmcnt waits and raw barriers as in
// gemm_nt_big_k.
template <bool CONV, int NSLOT = 2>
__global__ __launch_bounds__(NT_THREADS, 2) void gemm_nt_k(const bf16_t* __restrict__ A_, int64_t lda_,
                                                             const bf16_t* __restrict__ Bm_, int64_t ldb_,
                                                             void* __restrict__ C_, int64_t ldc_, int M_, int N_, int K_,
                                                             NtEpilogue ep, int ksplit, ConvGeom cg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // (wave-uniform selection of the launch's first or second product: everything below sees one problem)
  const bool second = !CONV && ep.tiles0 > 0 && (int)blockIdx.x >= ep.tiles0;
  const bf16_t* __restrict__ A = second ? ep.A2 : A_;
  const bf16_t* __restrict__ Bm = second ? ep.B2 : Bm_;
  void* __restrict__ C = second ? ep.C2 : C_;
  const int64_t lda = second ? ep.lda2 : lda_, ldb = second ? ep.ldb2 : ldb_, ldc = second ? ep.ldc2 : ldc_;
  const int M = second ? ep.M2 : M_, N = second ? ep.N2 : N_, K = second ? ep.K2 : K_;
  if (second) ep.bias = ep.bias2, ep.resid = ep.resid2, ep.ldr = ep.ldr2;
  const int bid = second ? (int)blockIdx.x - ep.tiles0 : (int)blockIdx.x;
  const int tiles_n = (N + BN - 1) / BN;
  const int tiles_m = (M + BM - 1) / BM;
  const int lid = xcd_remap(bid, tiles_m * tiles_n);
  const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // ---- staging addresses: wave w DMA-copies chunks 4w..4w+3 (8 rows each) of the A tile and of the B tile
  const int srow = lane >> 3, sslot = lane & 7;
  const bf16_t* a_src[4];
  const bf16_t* b_src[4];
  int tap_ok[4];  // CONV: bit t set <=> tap t of this lane's pixel row lies inside the image
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int r = (wave * 4 + c) * 8 + srow;
    const int q = sslot ^ ((r >> 1) & 7);
    int gm = m0 + r;
    gm = gm < M ? gm : M - 1;
    int gn = n0 + r;
    gn = gn < N ? gn : N - 1;
    b_src[c] = Bm + (int64_t)gn * ldb + q * 8;
    if (CONV) {
      a_src[c] = A + (int64_t)gm * cg.ldx + q * 8;
      int px, py;
      cg.pixel(gm, py, px);
      int ok = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
        if (yy >= 0 && yy < cg.H && xx >= 0 && xx < cg.W) ok |= 1 << t;
      }
      tap_ok[c] = ok;
    } else {
      a_src[c] = A + (int64_t)gm * lda + q * 8;
      tap_ok[c] = 0;
    }
  }
  auto stage = [&](int kt, int buf) {
    char* ba = smem + buf * 32768 + wave * 4096;
    char* bb = ba + 16384;
    if (CONV) {
      const int k0 = kt * BK;
      const int tap = k0 / cg.Ci, ci0 = k0 - tap * cg.Ci;  // wave-uniform
      const int64_t shift = ((int64_t)(tap / 3 - 1) * cg.W + (tap % 3 - 1)) * cg.ldx + ci0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bf16_t* src = ((tap_ok[c] >> tap) & 1) ? a_src[c] + shift : cg.zero;
        glds16(src, ba + c * 1024);
        glds16(b_src[c] + kt * BK, bb + c * 1024);
      }
      return;
    }
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

  // split-K (skinny GEMMs with a deep contraction, f32 plain output only): blockIdx.y owns k-steps [k_lo, k_hi) and the
  // partial tiles meet in C through f32 atomics (C zeroed by the launcher)
  const int nk_all = K / BK;
  const int k_lo = (int)((int64_t)nk_all * blockIdx.y / ksplit), nk = (int)((int64_t)nk_all * (blockIdx.y + 1) / ksplit);
  // the bias of this lane's eight output columns, requested BEFORE the first operand stage (the oldest loads in flight: the counted
  // waits below only get stricter): in the epilogue the eight row passes used to fetch them again, one dependent round trip per
  // two passes -- 4.5 us of an 18 us launch (round 6, scripts/lab/unet_nt_shapes.py: every shape with a bias)
  const int erow = lane >> 3, ecol = (lane & 7) * 8;
  const int en = n0 + wc * 64 + ecol;
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = (ep.bias && en + e < N) ? ep.bias[en + e] : 0.f;
  if (NSLOT == 2) {
    if (k_lo < nk) stage(k_lo, k_lo & 1);
  } else {
#pragma unroll
    for (int s = 0; s < NSLOT - 1; ++s)
      if (k_lo + s < nk) stage(k_lo + s, (k_lo + s) % NSLOT);
  }
  for (int kt = k_lo; kt < nk; ++kt) {
    if (NSLOT == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    } else {
      // stage kt must have landed; the (up to NSLOT - 2) younger stages stay in flight: 8 DMA instructions per wave and stage
      const int younger = nk - 1 - kt < NSLOT - 2 ? nk - 1 - kt : NSLOT - 2;
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // raw: __syncthreads() would drain vmcnt (the DMA counts as an LDS write)
      if (kt + NSLOT - 1 < nk) stage(kt + NSLOT - 1, (kt + NSLOT - 1) % NSLOT);  // into the slot consumed in iteration kt - 1
    }
    const char* ta = smem + (NSLOT == 2 ? (kt & 1) : kt % NSLOT) * 32768;
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
  // the residual rows of all eight passes requested at once, under the accumulators' trip through the slab (one round trip
  // instead of four: cold operands cost 9 us of a 28 us launch)
  // (compiler-visible vmcnt(0) -- every operand stage has landed long ago -- so that the waitcnt pass does not put one in front of
  //  the slab's LDS traffic, behind the residual loads: it cannot tell LDS accesses from the DMA's targets otherwise)
  __builtin_amdgcn_s_waitcnt(0x0F70);
  const bool efull = en + 8 <= N;
  u32x4_t rv[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int m = m0 + wr * 64 + p * 8 + erow;
    rv[p] = u32x4_t{0u, 0u, 0u, 0u};
    if (ep.resid && efull && m < M) rv[p] = *(const u32x4_t*)(ep.resid + (int64_t)m * ep.ldr + en);
  }
  // every wave is done reading the operand buffers (its fragments were consumed by its MFMAs): a raw barrier -- __syncthreads()
  // would drain the residual loads just issued
  __builtin_amdgcn_s_barrier();
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
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the slab is wave-private: lanes read what other lanes of the wave wrote
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row = p * 8 + erow;
    const int m = m0 + wr * 64 + row;
    const int n = en;
    if (m >= M || n >= N) continue;
    float v[8];
    *(f32x4_t*)&v[0] = *(const f32x4_t*)&slab[row * STAGE_PITCH + ecol];
    *(f32x4_t*)&v[4] = *(const f32x4_t*)&slab[row * STAGE_PITCH + ecol + 4];
    const bool full = efull;
    if (ep.bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bv[e];
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
    } else if (ep.act == DL_ACT_GELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
    }
    if (ep.resid) {
      const bf16_t* rp = ep.resid + (int64_t)m * ep.ldr + n;
      float rr[8], gg[8];
      if (full) {
        unpack8(rv[p], rr);
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
    if (ksplit > 1 && ep.part_stride > 0) {
      float* cp = (float*)C + (int64_t)blockIdx.y * ep.part_stride + (int64_t)m * ldc + n;
      for (int e = 0; e < 8 && n + e < N; ++e) cp[e] = v[e];
    } else if (ksplit > 1) {
      float* cp = (float*)C + (int64_t)m * ldc + n;
      for (int e = 0; e < 8 && n + e < N; ++e) unsafeAtomicAdd(cp + e, v[e]);
    } else if (ep.out_f32) {
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

// =====================================================================================================
// gemm_nt_big_k: the token-GEMM workhorse (M = batch*tokens is huge, K = 384..3072, N = 384..3072).
//
// Why a second kernel: a 128x128 tile moves 64 FLOP per byte through the CU's vector-memory path
// (global_load_lds tops out at 64 B/clk/CU), which is EXACTLY the MFMA rate, and with one stage of prefetch
// the rocprof PMC run of the kernel above shows the waves parked on vmcnt/barrier 60-70 % of the time.
//   * 256 x TN tiles (TN = 128 | 192): 85 | 110 FLOP per staged byte;
//   * persistent: one 512-thread workgroup per CU (8 waves, 4(M) x 2(N), each 64 x TN/2) walks a list of
//     tiles; the (tile, k-step) space is flattened and fed through an NST-deep LDS ring with COUNTED
//     s_waitcnt vmcnt(N) (never 0 in steady state): NST-1 operand stages (up to ~96 KiB per CU) stay in
//     flight across barriers, tile boundaries and the epilogue stores;
//   * XCD-aware schedule: at every slot the 32 workgroups of one XCD own 32 consecutive tiles (n fastest),
//     i.e. they share A row-panels in that XCD's L2;
//   * operands are swapped in the MFMA (A-operand = weight rows, B-operand = activation rows) so that each
//     lane ends up with 4 consecutive output columns of one row; one v_permlane32_swap per register pair
//     widens that to 8 columns = one 16-byte store/load per lane (no LDS round trip in the epilogue).
// EPI = 0: plain bf16 store (its store count is known, so the ring keeps flowing through the epilogue);
// EPI = 1: every epilogue option (drains the ring once per tile).
// Requires M % 256 == 0, N % TN == 0, K % 64 == 0.
// =====================================================================================================
#define TBM 256
#define BIG_THREADS 512

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int JN, int TN_, int EPI>
__device__ __forceinline__ void nt_epilogue_regs(f32x16_t (&acc)[JN][2], int m_base, int n_base, int lane, void* C,
                                                 int64_t ldc, const NtEpilogue& ep);
template <int JN, int TN_>
__device__ __forceinline__ void nt_epilogue_ssq(f32x16_t (&acc)[JN][2], int m_base, int n_base, int lane, void* C, int64_t ldc,
                                                const NtEpilogue& ep);

template <int TN_, int NST, int EPI>
__global__ __launch_bounds__(BIG_THREADS, 2) void gemm_nt_big_k(const bf16_t* __restrict__ A, int64_t lda,
                                                                  const bf16_t* __restrict__ Bm, int64_t ldb,
                                                                  void* __restrict__ C, int64_t ldc, int M, int N,
                                                                  int K, NtEpilogue ep) {
  constexpr int STAGE = (TBM + TN_) * 128;   // bytes per ring slot
  constexpr int CH = (TBM + TN_) / 8 / 8;    // DMA chunks (1 KiB) per wave per stage: 6 | 7
  constexpr int JN = TN_ / 64;               // 32-wide MFMA column tiles per wave: 2 | 3
  constexpr int ESTORES = 4 * JN;            // 16-byte stores per wave in the EPI==0 epilogue
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = N / TN_, ntiles = (M / TBM) * tiles_n;
  const int nk = K / BK;
  const int G = gridDim.x;  // multiple of 8
  const int slot0 = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int cnt = (ntiles - slot0 + G - 1) / G;  // tiles owned by this workgroup (slot0 < ntiles by launch)
  const int total = cnt * nk;

  // ---- per-lane constant parts of the DMA chunks this wave issues per stage
  int64_t src_off[CH];
  int lds_off[CH];
  bool is_a[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = wave * CH + i;
    is_a[i] = c < TBM / 8;
    const int cc = is_a[i] ? c : c - TBM / 8;
    const int r = cc * 8 + (lane >> 3);
    const int q = (lane & 7) ^ ((r >> 1) & 7);
    src_off[i] = (int64_t)r * (is_a[i] ? lda : ldb) + q * 8;
    lds_off[i] = (is_a[i] ? 0 : TBM * 128) + cc * 1024;
  }
  // DMA cursor: runs NST-1 stages ahead of the compute cursor
  int s_tile = slot0, s_kt = 0, s_it = 0;
  const bf16_t* s_ta = A + (int64_t)((s_tile / tiles_n) * TBM) * lda;
  const bf16_t* s_tb = Bm + (int64_t)((s_tile % tiles_n) * TN_) * ldb;
  auto stage_next = [&]() {
    char* base = smem + (s_it % NST) * STAGE;
#pragma unroll
    for (int i = 0; i < CH; ++i) glds16((is_a[i] ? s_ta : s_tb) + src_off[i] + s_kt * BK, base + lds_off[i]);
    ++s_it;
    if (++s_kt == nk) {
      s_kt = 0;
      s_tile += G;
      s_ta = A + (int64_t)((s_tile / tiles_n) * TBM) * lda;
      s_tb = Bm + (int64_t)((s_tile % tiles_n) * TN_) * ldb;
    }
  };

  if (ep.stagger > 0) {
    // every tile costs the same time, so without a skew all 256 workgroups compute together and then store together (a
    // 50 MB write burst every ~11 us that each wave has to see acknowledged before its next-but-one k-step); four phase
    // groups spread the bursts over the tile period and let one group's stores drain while the others run MFMAs
    // (the column tiles of one A row-panel sit on one XCD in consecutive slots: they keep a common phase so that they still
    // read the panel together out of that XCD's L2)
    const int phase = ((blockIdx.x >> 3) / tiles_n) & ((ep.stagger >> 8) ? 1 : 3);
    for (int w = 0; w < phase * nk * (ep.stagger & 255); ++w) __builtin_amdgcn_s_sleep(16);
  }
  int xrow[2], wrow[JN];
#pragma unroll
  for (int i = 0; i < 2; ++i) xrow[i] = wm * 64 + i * 32 + (lane & 31);
#pragma unroll
  for (int j = 0; j < JN; ++j) wrow[j] = wn * (TN_ / 2) + j * 32 + (lane & 31);

  f32x16_t acc[JN][2];
#pragma unroll
  for (int j = 0; j < JN; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < total) stage_next();

  int tile = slot0, kt = 0;
  bool after_epi = false;
  for (int it = 0; it < total; ++it) {
    // stage `it` must have landed; the NST-2 younger stages (and the previous tile's stores) may stay in flight
    if (it + NST - 2 < total) {
      // (the epilogue's stores are the youngest entries of the in-order counter: a counted wait lets them drain under this k-step)
      // (EPI 3 issues up to two statistics atomics behind its ESTORES stores: waiting for "<= ESTORES outstanding" is then merely
      // stricter than needed -- the two oldest stores have to land as well -- never too weak)
      if ((EPI == 0 || EPI == 3) && after_epi) wait_vmcnt<(NST - 2) * CH + ESTORES>();
      else if (EPI == 2 && after_epi && C) wait_vmcnt<(NST - 2) * CH + 6 * JN>();
      else if (EPI == 2 && after_epi) wait_vmcnt<(NST - 2) * CH + 2 * JN>();
      else wait_vmcnt<(NST - 2) * CH>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();  // raw barrier: __syncthreads() would drain vmcnt (the DMA counts as an LDS write)
    // NST >= 3: the DMA for stage it+NST-1 is issued AFTER this step's MFMAs are queued (its issue is bound by the
    // 64 B/clk vector-memory path; up front it would hold every wave off the matrix pipe for ~900 cycles)
    if (NST < 3 && s_it < total) stage_next();  // into the slot whose stage was consumed in iteration it-1
    const char* sa = smem + (it % NST) * STAGE;
    const char* sb = sa + TBM * 128;
    {
      // Fragment software pipeline: the weight fragment of MFMA pair (kk, j) is read TWO pairs ahead into a three-entry register
      // ring and the two activation fragments of sub-step kk+1 during sub-step kk, each pair its own scheduling region -- the
      // LDS latency of a fragment is covered by the four MFMAs in front of its first use instead of being waited for before
      // every group of twelve (32 fragment registers).
      bf16x8_t xq[2][2], wq[3];
      auto rd_x = [&](int kk, int i) -> bf16x8_t {
        return *(const bf16x8_t*)(sa + xrow[i] * 128 + ((((kk << 1) | hi) ^ ((xrow[i] >> 1) & 7)) << 4));
      };
      auto rd_w = [&](int kk, int j) -> bf16x8_t {
        return *(const bf16x8_t*)(sb + wrow[j] * 128 + ((((kk << 1) | hi) ^ ((wrow[j] >> 1) & 7)) << 4));
      };
      xq[0][0] = rd_x(0, 0);
      xq[0][1] = rd_x(0, 1);
      wq[0] = rd_w(0, 0);
      wq[1] = rd_w(0, 1);
#pragma unroll
      for (int s = 0; s < 4 * JN; ++s) {
        const int kk = s / JN, j = s % JN;
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < 4 * JN) wq[(s + 2) % 3] = rd_w((s + 2) / JN, (s + 2) % JN);
        if (kk < 3 && j < 2) xq[(kk + 1) & 1][j] = rd_x(kk + 1, j);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[s % 3], xq[kk & 1][i], acc[j][i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (NST >= 3 && s_it < total) stage_next();
    after_epi = false;
    if (++kt == nk) {
      kt = 0;
      after_epi = true;
      // ---- epilogue of `tile` straight from registers: acc[j][i][r] = C[m][n] with
      //      m = m0 + wm*64 + i*32 + (lane&31),  n = n0 + wn*TN/2 + j*32 + 8*(r>>2) + 4*hi + (r&3)
      if constexpr (EPI == 3)
        nt_epilogue_ssq<JN, TN_>(acc, (tile / tiles_n) * TBM + wm * 64, (tile % tiles_n) * TN_ + wn * (TN_ / 2), lane, C, ldc, ep);
      else
        nt_epilogue_regs<JN, TN_, EPI>(acc, (tile / tiles_n) * TBM + wm * 64, (tile % tiles_n) * TN_ + wn * (TN_ / 2), lane, C, ldc,
                                       ep);
      if (EPI == 1) wait_vmcnt<0>();  // unknown number of epilogue memory ops: drain so the counted waits stay exact
      tile += G;
    }
  }
}

// register epilogue shared by the big-tile kernels (see gemm_nt_big_k header)
template <int JN, int TN_, int EPI>
__device__ __forceinline__ void nt_epilogue_regs(f32x16_t (&acc)[JN][2], int m_base, int n_base, int lane, void* C,
                                                 int64_t ldc, const NtEpilogue& ep) {
  const int hi = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m_base + i * 32 + (lane & 31);
    if (EPI == 2) {
      // fused PackedSwiGLU forward (nn.py:484-486).  The weight shadow is row-permuted so that every 32-column MFMA tile
      // holds x1 of 16 consecutive hidden units (columns 0..15) followed by x3 of the SAME units (columns 16..31): a lane
      // then owns x1 (acc regs 0..7) and x3 (acc regs 8..15) of the same 8 units, and after the half exchange each of the
      // three stores (x1, x3, h) is the usual 16 B per lane / 32 contiguous bytes per row of the plain epilogue.
#pragma unroll
      for (int j = 0; j < JN; ++j) {
        const int u0 = ((n_base + j * 32) >> 1) + 8 * hi;  // first hidden unit this lane stores after the exchange
        float v1[8], v3[8], hv[8], h8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) hv[e] = silu_f(acc[j][i][e]) * acc[j][i][8 + e];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j][i][e]), __float_as_uint(acc[j][i][4 + e]), false, false);
          auto s3 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j][i][8 + e]), __float_as_uint(acc[j][i][12 + e]), false, false);
          auto sh = __builtin_amdgcn_permlane32_swap(__float_as_uint(hv[e]), __float_as_uint(hv[4 + e]), false, false);
          v1[e] = __uint_as_float(s1[0]);
          v1[4 + e] = __uint_as_float(s1[1]);
          v3[e] = __uint_as_float(s3[0]);
          v3[4 + e] = __uint_as_float(s3[1]);
          h8[e] = __uint_as_float(sh[0]);
          h8[4 + e] = __uint_as_float(sh[1]);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
        // pre-activations in the reference layout u = [x1 | x3] (needed by the backward only: C == NULL in inference), then h
        if (C) {
          *(u32x4_t*)((bf16_t*)C + (int64_t)m * ldc + u0) = pack8(v1);
          *(u32x4_t*)((bf16_t*)C + (int64_t)m * ldc + ep.F + u0) = pack8(v3);
        }
        *(u32x4_t*)(ep.aux + (int64_t)m * ep.ld_aux + u0) = pack8(h8);
      }
      continue;
    }
    const bf16_t* gate_row = (EPI && ep.gate) ? ep.gate + (int64_t)(m / (int)ep.rows_per_gate) * ep.ldg : nullptr;
    // residual / gate rows of this row group are fetched up front (one exposed memory latency per group, not per chunk)
    u32x4_t lr[JN][2], lg[JN][2];
    if (EPI == 1 && ep.resid) {
#pragma unroll
      for (int j = 0; j < JN; ++j)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int n = n_base + j * 32 + 16 * gp + 8 * hi;
          lr[j][gp] = *(const u32x4_t*)(ep.resid + (int64_t)m * ep.ldr + n);
          if (gate_row) lg[j][gp] = *(const u32x4_t*)(gate_row + n);
        }
    }
#pragma unroll
    for (int j = 0; j < JN; ++j) {
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned x = __float_as_uint(acc[j][i][8 * gp + e]), y = __float_as_uint(acc[j][i][8 * gp + 4 + e]);
          auto sw = __builtin_amdgcn_permlane32_swap(x, y, false, false);
          v[e] = __uint_as_float(sw[0]);
          v[4 + e] = __uint_as_float(sw[1]);
          acc[j][i][8 * gp + e] = 0.f;
          acc[j][i][8 * gp + 4 + e] = 0.f;
        }
        const int n = n_base + j * 32 + 16 * gp + 8 * hi;
        if (EPI) {
          if (ep.bias) {
            const f32x4_t b0 = *(const f32x4_t*)(ep.bias + n), b1 = *(const f32x4_t*)(ep.bias + n + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] += b0[e];
              v[4 + e] += b1[e];
            }
          }
          if (ep.pre_out) *(u32x4_t*)(ep.pre_out + (int64_t)m * ldc + n) = pack8(v);
          if (ep.act == DL_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
          } else if (ep.act == DL_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
          }
          if (ep.resid) {
            float rr[8];
            unpack8(lr[j][gp], rr);
            if (gate_row) {
              float gg[8];
              unpack8(lg[j][gp], gg);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = rr[e] + gg[e] * v[e];
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += rr[e];
            }
          }
          if (ep.out_f32) {
            float* cp = (float*)C + (int64_t)m * ldc + n;
            *(f32x4_t*)cp = *(f32x4_t*)&v[0];
            *(f32x4_t*)(cp + 4) = *(f32x4_t*)&v[4];
          } else {
            *(u32x4_t*)((bf16_t*)C + (int64_t)m * ldc + n) = pack8(v);
          }
        } else {
          *(u32x4_t*)((bf16_t*)C + (int64_t)m * ldc + n) = pack8(v);
        }
      }
    }
  }
}

// EPI 3: the plain epilogue + row sums of squares (see NtEpilogue::ssq)
template <int JN, int TN_>
__device__ __forceinline__ void nt_epilogue_ssq(f32x16_t (&acc)[JN][2], int m_base, int n_base, int lane, void* C, int64_t ldc,
                                                const NtEpilogue& ep) {
  const int hi = lane >> 5;
  const int tile = n_base / TN_;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m_base + i * 32 + (lane & 31);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned x = __float_as_uint(acc[j][i][8 * gp + e]), y = __float_as_uint(acc[j][i][8 * gp + 4 + e]);
          auto sw = __builtin_amdgcn_permlane32_swap(x, y, false, false);
          v[e] = __uint_as_float(sw[0]);
          v[4 + e] = __uint_as_float(sw[1]);
          acc[j][i][8 * gp + e] = 0.f;
          acc[j][i][8 * gp + 4 + e] = 0.f;
        }
        const u32x4_t pk = pack8(v);
        *(u32x4_t*)((bf16_t*)C + (int64_t)m * ldc + n_base + j * 32 + 16 * gp + 8 * hi) = pk;
        float r[8];
        unpack8(pk, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += r[e] * r[e];
      }
    if (tile < ep.ssq_tiles) {
      s = xor32_sum(s);  // the row's other 96 columns of this wave live in the other lane half
      if (hi == 0) unsafeAtomicAdd(ep.ssq + (int64_t)m * ep.ssq_tiles + tile, s);
    }
  }
}

template <int TN_, int NST>
static int launch_big(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                      int64_t N, int64_t K, const NtEpilogue& ep_in, int epi, hipStream_t stream) {
  constexpr int LDS = NST * (TBM + TN_) * 128;
  static DevOnce once;
  const int n_cu = dev_cus(once, [] {
    (void)hipFuncSetAttribute((const void*)gemm_nt_big_k<TN_, NST, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)hipFuncSetAttribute((const void*)gemm_nt_big_k<TN_, NST, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    (void)hipFuncSetAttribute((const void*)gemm_nt_big_k<TN_, NST, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if constexpr (TN_ == 384)
      (void)hipFuncSetAttribute((const void*)gemm_nt_big_k<TN_, NST, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  });
  const int ntiles = (int)((M / TBM) * (N / TN_));
  const int budget = dl_wg_budget(n_cu);
  int grid = budget < ntiles ? budget : ntiles;
  grid &= ~7;
  NtEpilogue ep = ep_in;
  // phase skew of the tile starts (sleep units per k-step and phase group): pays only when every workgroup walks many tiles
  // (measured: MLP-up)
  ep.stagger = (ntiles >= 6 * grid) ? 2 : 0;
#define BIG_GO(E)                                                                                                     \
  hipLaunchKernelGGL((gemm_nt_big_k<TN_, NST, E>), grid, BIG_THREADS, LDS, stream, (const bf16_t*)A, lda, (const bf16_t*)B, \
                     ldb, C, ldc, (int)M, (int)N, (int)K, ep)
  if (epi == 0) BIG_GO(0);
  else if (epi == 1) BIG_GO(1);
  else if (epi == 3) {
    if constexpr (TN_ == 384) BIG_GO(3);
    else return 1;
  } else BIG_GO(2);
#undef BIG_GO
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// picks the big-tile variant; returns 1 if no big kernel applies (caller falls back), else the launch status (<= 0)
static int dispatch_big(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                        int64_t K, const NtEpilogue& ep, int epi, hipStream_t stream) {
  if (M % TBM) return 1;
  // One persistent workgroup per CU, so a launch runs in ceil(tiles / 256) rounds: score every tile width the shape allows by
  // (fraction of the CU-rounds that do work) x (relative efficiency of the tile: wider tiles stage fewer bytes per FLOP) and take
  // the best; widths that would leave more than 30 % of the rounds idle are not considered and the shape falls to the 128x128
  // kernel.  (A 256x384 tiling of an [8192, 768] output is 64 tiles -- a quarter of the chip -- and measured 2x slower than the
  // 128x128 kernel; [16384, 768] is best at 256x192 = 256 tiles; scripts/gemm_mid_bench.py.)
  const int64_t mt = M / TBM;
  auto score = [&](int tn, double eff, bool allowed) -> double {
    if (!allowed || N % tn) return 0.0;
    const int64_t tiles = mt * (N / tn), rounds = (tiles + 255) / 256;
    const double util = (double)tiles / (double)(rounds * 256);
    return util >= 0.7 ? util * eff : 0.0;
  };
  const double s384 = score(384, 1.00, epi == 0 || epi == 2), s256 = score(256, 0.95, true);
  const double s192 = score(192, 0.90, true), s128 = score(128, 0.80, true);
  const double best = fmax(fmax(s384, s256), fmax(s192, s128));
  if (best > 0.0) {
    if (best == s384) return launch_big<384, 2>(A, lda, B, ldb, C, ldc, M, N, K, ep, epi, stream);
    if (best == s256) return launch_big<256, 2>(A, lda, B, ldb, C, ldc, M, N, K, ep, epi, stream);
    if (best == s192) return launch_big<192, 2>(A, lda, B, ldb, C, ldc, M, N, K, ep, epi, stream);
    return launch_big<128, 3>(A, lda, B, ldb, C, ldc, M, N, K, ep, epi, stream);
  }
  if (epi == 2) {  // the fused SwiGLU epilogue only exists in the persistent kernels: take the tiling with the most tiles
    if (N % 192 == 0 && mt * (N / 192) >= 64)
      return launch_big<192, 2>(A, lda, B, ldb, C, ldc, M, N, K, ep, epi, stream);
    if (N % 384 == 0 && mt * (N / 384) >= 64)
      return launch_big<384, 2>(A, lda, B, ldb, C, ldc, M, N, K, ep, epi, stream);
    if (N % 128 == 0 && mt * (N / 128) >= 64)  // (2F = 4096: the 512-wide configurations)
      return launch_big<128, 3>(A, lda, B, ldb, C, ldc, M, N, K, ep, epi, stream);
  }
  return 1;
}

static int g_nt_deep = 1;  // LAB A/B switch (not in the header): 0 = the two-slot ring everywhere
extern "C" __attribute__((visibility("default"))) void dl_lab_set_nt_deep(int on) { g_nt_deep = on; }
// the 128 x 128 kernel: the four-slot ring when the launch is at most one workgroup per CU (and deep enough to have a steady state)
template <bool CONV>
static void launch_nt_small(dim3 grid, hipStream_t stream, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, void* C, int64_t ldc,
                            int M, int N, int K, const NtEpilogue& ep, int ksplit, const ConvGeom& cg) {
  static DevOnce once;
  const int n_cu = dev_cus(once, [] {
    (void)hipFuncSetAttribute((const void*)gemm_nt_k<CONV, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
  });
  if (g_nt_deep && (int64_t)grid.x * grid.y <= n_cu && K / BK / ksplit >= 4)
    hipLaunchKernelGGL((gemm_nt_k<CONV, 4>), grid, NT_THREADS, 4 * 32768, stream, A, lda, B, ldb, C, ldc, M, N, K, ep, ksplit, cg);
  else
    hipLaunchKernelGGL((gemm_nt_k<CONV, 2>), grid, NT_THREADS, NT_LDS_BYTES, stream, A, lda, B, ldb, C, ldc, M, N, K, ep, ksplit, cg);
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
                ldg, rows_per_gate > 0 ? rows_per_gate : 1, nullptr, 0, 0, 0, 0};
  {
    const bool plain = !bias && act == DL_ACT_NONE && out_dtype == DL_BF16 && !pre_out && !resid;
    if (!bias || ((uintptr_t)bias & 15) == 0) {
      const int rc = dispatch_big(A, lda, B, ldb, C, ldc, M, N, K, ep, plain ? 0 : 1, (hipStream_t)stream);
      if (rc <= 0) return rc;
    }
  }
  const int nwg = cdiv(M, BM) * cdiv(N, BN);
  int ksplit = 1;
  if (out_dtype == DL_F32 && !bias && act == DL_ACT_NONE && !pre_out && !resid && nwg < 64 && K >= 2048) {
    ksplit = 256 / nwg;
    if (ksplit > K / 512) ksplit = (int)(K / 512);
    if (ksplit > 1) {
      const hipError_t e = (ldc == N) ? hipMemsetAsync(C, 0, (size_t)M * N * 4, (hipStream_t)stream)
                                      : hipMemset2DAsync(C, (size_t)ldc * 4, 0, (size_t)N * 4, (size_t)M, (hipStream_t)stream);
      if (e != hipSuccess) ksplit = 1;
    } else {
      ksplit = 1;
    }
  }
  launch_nt_small<false>(dim3(nwg, ksplit), (hipStream_t)stream, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, (int)M, (int)N, (int)K, ep,
                         ksplit, ConvGeom{});
  DL_LAUNCH_CHECK();
  return DL_OK;
}

/* two independent products C_i = A_i B_i^T (+ bias_i, + resid_i), bf16 in / bf16 out, in ONE launch of the 128 x 128 kernel when both
 * are small (together at most 1.5 workgroups per CU: neither would take a persistent tiling); otherwise -- and for n == 1 -- the two
 * dl_gemm_nt calls it stands for.  Same products in the same order as those calls: bit-identical results. */
extern "C" int dl_gemm_nt_pair(const dl_nt_problem_t* p, int n, dl_stream_t stream) {
  DL_CHECK_ARG(p && (n == 1 || n == 2), "dl_gemm_nt_pair: one or two problems");
  static DevOnce once;
  const int n_cu = dev_cus(once, [] {});
  int64_t nwg[2] = {0, 0};
  bool small = n == 2;
  for (int i = 0; i < n; ++i) {
    const dl_nt_problem_t& q = p[i];
    DL_CHECK_ARG(q.A && q.B && q.C && q.M > 0 && q.N > 0 && q.K > 0 && q.K % BK == 0, "dl_gemm_nt_pair: operand %d", i);
    DL_CHECK_ARG(q.lda % 8 == 0 && q.ldb % 8 == 0 && q.ldc % 8 == 0 && q.lda >= q.K && q.ldb >= q.K && q.ldc >= q.N &&
                     (!q.resid || q.ldr % 8 == 0) && q.M < (1ll << 31) && q.N < (1ll << 31),
                 "dl_gemm_nt_pair: leading dims of problem %d", i);
    DL_CHECK_ARG((((uintptr_t)q.A | (uintptr_t)q.B | (uintptr_t)q.C | (uintptr_t)q.resid) & 15) == 0, "dl_gemm_nt_pair: 16-byte alignment");
    nwg[i] = (int64_t)cdiv(q.M, BM) * cdiv(q.N, BN);
  }
  if (small) small = nwg[0] + nwg[1] <= (3 * (int64_t)n_cu) / 2;
  if (!small) {
    for (int i = 0; i < n; ++i) {
      const int rc = dl_gemm_nt(p[i].A, p[i].lda, p[i].B, p[i].ldb, p[i].C, p[i].ldc, p[i].M, p[i].N, p[i].K, p[i].bias, DL_ACT_NONE, DL_BF16,
                                nullptr, p[i].resid, p[i].ldr, nullptr, 0, 1, stream);
      if (rc != DL_OK) return rc;
    }
    return DL_OK;
  }
  // the product with the shallower contraction goes first: launch_nt_small picks the ring depth from the K it is given (a four-slot
  // ring needs four k-steps to have a steady state)
  const int f = p[0].K <= p[1].K ? 0 : 1, g = 1 - f;
  NtEpilogue ep{p[f].bias, DL_ACT_NONE, 0, nullptr, (const bf16_t*)p[f].resid, p[f].ldr, nullptr, 0, 1, nullptr, 0, 0, 0, 0};
  ep.A2 = (const bf16_t*)p[g].A, ep.B2 = (const bf16_t*)p[g].B, ep.C2 = p[g].C, ep.bias2 = p[g].bias, ep.resid2 = (const bf16_t*)p[g].resid;
  ep.lda2 = p[g].lda, ep.ldb2 = p[g].ldb, ep.ldc2 = p[g].ldc, ep.ldr2 = p[g].ldr;
  ep.M2 = (int)p[g].M, ep.N2 = (int)p[g].N, ep.K2 = (int)p[g].K, ep.tiles0 = (int)nwg[f];
  launch_nt_small<false>(dim3((unsigned)(nwg[0] + nwg[1]), 1), (hipStream_t)stream, (const bf16_t*)p[f].A, p[f].lda, (const bf16_t*)p[f].B, p[f].ldb,
                         p[f].C, p[f].ldc, (int)p[f].M, (int)p[f].N, (int)p[f].K, ep, 1, ConvGeom{});
  DL_LAUNCH_CHECK();
  return DL_OK;
}

/* the plain bf16 product on the persistent 256 x 384 tiles + per-row sums of squares of the first `ssq_tiles` 384-wide column tiles
 * (the QK-RMSNorm statistics of the qkv GEMM, mmdit.py:81-88, nn.py:427-431); DL_ERR_UNSUPPORTED for other shapes */
extern "C" int dl_gemm_nt_ssq(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                              int64_t K, float* ssq, int64_t ssq_tiles, dl_stream_t stream) {
  DL_CHECK_ARG(A && B && C && ssq && M > 0 && N > 0 && K > 0, "dl_gemm_nt_ssq: null/empty operand");
  DL_CHECK_ARG(K % BK == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && lda >= K && ldb >= K && ldc >= N,
               "dl_gemm_nt_ssq: K %% 64, leading dims (lda=%lld ldb=%lld ldc=%lld)", (long long)lda, (long long)ldb, (long long)ldc);
  DL_CHECK_ARG((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0 && ssq_tiles >= 1 && ssq_tiles <= N / 384 + 1,
               "dl_gemm_nt_ssq: alignment / ssq_tiles");
  if (M % TBM || N % 384 || (M / TBM) * (N / 384) < 64) {
    dl_set_error("dl_gemm_nt_ssq: needs M %% 256 == 0, N %% 384 == 0 and >= 64 tiles (M=%lld N=%lld)", (long long)M, (long long)N);
    return DL_ERR_UNSUPPORTED;
  }
  NtEpilogue ep{};
  ep.rows_per_gate = 1;
  ep.ssq = ssq;
  ep.ssq_tiles = (int)ssq_tiles;
  return launch_big<384, 2>(A, lda, B, ldb, C, ldc, M, N, K, ep, 3, (hipStream_t)stream);
}

/* fused MLP-up + PackedSwiGLU forward: U = X W1^T (reference layout [x1 | x3]) and H = silu(x1) * x3 in one pass.
 * Wp is the row-permuted bf16 shadow produced by dl_cast_weight_swiglu.  Returns DL_ERR_UNSUPPORTED when the shape has
 * no big-tile kernel (caller then runs dl_gemm_nt + dl_swiglu_fwd). */
extern "C" int dl_gemm_nt_swiglu(const void* X, int64_t ldx, const void* Wp, int64_t ldw, void* U, int64_t ldu, void* H,
                                 int64_t ldh, int64_t M, int64_t F, int64_t K, dl_stream_t stream) {
  DL_CHECK_ARG(X && Wp && H && M > 0 && F > 0 && K > 0, "dl_gemm_nt_swiglu: null/empty operand");  // U may be NULL (inference)
  DL_CHECK_ARG(K % BK == 0 && F % 16 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && (!U || ldu % 8 == 0) && ldh % 8 == 0,
               "dl_gemm_nt_swiglu: K %% 64, F %% 16, ld %% 8");
  NtEpilogue ep{nullptr, 0, 0, nullptr, nullptr, 0, nullptr, 0, 1, (bf16_t*)H, ldh, (int)F};
  const int rc = dispatch_big(X, ldx, Wp, ldw, U, ldu, M, 2 * F, K, ep, 2, (hipStream_t)stream);
  if (rc == 1) {
    dl_set_error("dl_gemm_nt_swiglu: no fused kernel for M=%lld F=%lld", (long long)M, (long long)F);
    return DL_ERR_UNSUPPORTED;
  }
  return rc;
}

// =====================================================================================================
// TN (wgrad): C[m,n] += sum_r A[r,m] B[r,n].  Operand tiles are [64 r][128 cols] row-major in LDS (DMA'd
// straight from the row-major activations), MFMA fragments come out of ds_read_b64_tr_b16 transposing reads.
// =====================================================================================================
// Inline assembly ON PURPOSE (see gemm_w4.hip): behind a direct-to-LDS DMA the waitcnt pass puts s_waitcnt vmcnt(0) in front of a
// compiler-visible transposing read (it cannot tell the read from the DMA's LDS target), i.e. the prefetched stage is waited for
// before the current one is consumed and the DMA never overlaps the MFMAs.  The asm read has no memory operand; its lgkmcnt
// wait is explicit (tr_landed).
__device__ __forceinline__ s16x4_t lds_tr16(const char* p) {
  s16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((unsigned)(uintptr_t)(lds_void_t*)p));
  return v;
}
template <int NA, int NB>
__device__ __forceinline__ void tr_landed(bf16x8_t (&a)[NA], bf16x8_t (&b)[NB]) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < NA; ++i) asm volatile("" : "+v"(a[i]));  // no MFMA on these registers can be scheduled above the wait
#pragma unroll
  for (int j = 0; j < NB; ++j) asm volatile("" : "+v"(b[j]));
}

// (A four-slot ring for launches of at most one workgroup per CU -- what gemm_nt_k<CONV, 4> does -- was measured here in round 6 and
// changes nothing, alone or in the UNet step: profiles/r06_t_tn_ring_and_split_sweep.txt; a lone workgroup's k-step costs 0.64 us
// with either ring.)
template <bool CONV>
__global__ __launch_bounds__(NT_THREADS, 2) void gemm_tn_k(const bf16_t* __restrict__ A, int64_t lda,
                                                             const bf16_t* __restrict__ Bm, int64_t ldb,
                                                             float* __restrict__ C, int64_t ldc, int M, int N, int R,
                                                             int steps_per_split, ConvGeom cg, int64_t part_stride) {
  // part_stride > 0: split s stores its partial [M, N] image with plain stores at C + s * part_stride (the caller folds the images
  // in a fixed order: bit-reproducible; every split owns at least one step); 0: the splits meet in C through f32 atomics
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
    if (CONV) {
      // the 128 columns of this tile lie inside ONE tap (Ci % 128 == 0): column cm of cols is channel ci0 + (cm - m0)
      const int tap = m0 / cg.Ci, ci0 = m0 - tap * cg.Ci;
      a_src[c] = A + ((int64_t)r + (int64_t)(tap / 3 - 1) * cg.W + (tap % 3 - 1)) * cg.ldx + ci0 + (cm - m0);
    } else {
      a_src[c] = A + (int64_t)r * lda + cm;
    }
    b_src[c] = Bm + (int64_t)r * ldb + cn;
  }
  auto stage = [&](int st, int buf) {
    char* ba = smem + buf * 32768 + wave * 4096;
    char* bb = ba + 16384;
    const int64_t r0 = (int64_t)st * BK;
    if (CONV) {
      const int tap = m0 / cg.Ci;
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int64_t p = r0 + (wave * 4 + c) * 4 + srow;  // pixel row of this lane
        int px, py;
        cg.pixel(p, py, px);
        const bool ok = p < cg.npix && py + dy >= 0 && py + dy < cg.H && px + dx >= 0 && px + dx < cg.W;
        glds16(ok ? a_src[c] + r0 * cg.ldx : cg.zero, ba + c * 1024);
        glds16(b_src[c] + r0 * ldb, bb + c * 1024);
      }
      return;
    }
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
      tr_landed(af, bfg);
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
        if (m < M && n < N) {
          if (part_stride > 0) C[(int64_t)split * part_stride + (int64_t)m * ldc + n] = acc[i][j][r];
          else unsafeAtomicAdd(&C[(int64_t)m * ldc + n], acc[i][j][r]);
        }
      }
    }
}

// =====================================================================================================
// gemm_tn_big_k: wgrad of the token linears (reduction over R = batch*tokens, output [Mo, No] small).
// Same reasoning as gemm_nt_big_k: a 128x128 output tile stages 64 FLOP/B; here 384 x 128 tiles
// (96 FLOP/B), 8 waves as 4(M) x 2(N), each 96 x 64 = 3x2 MFMA 32x32x16 tiles, one workgroup per CU,
// the reduction split over workgroups (<= one per CU per launch) and combined with f32 atomics.
// Operand images in LDS are the row-major [64 r][384|128] slabs written by the DMA (1 KiB chunks run over
// row boundaries for the 768-byte A rows: the per-lane source address is derived from the linear image
// offset); fragments come out through ds_read_b64_tr_b16 with the 32-byte-chunk XOR swizzle of gemm_tn_k.
// Requires No % 128 == 0, R % 64 == 0, Mo % 8 == 0 (a ragged last 384-row m-tile clamps its loads and masks its atomics).
// =====================================================================================================
#define WBM 384
#define WBN 128
#define W_STAGE ((WBM + WBN) * 2 * BK)  // 65536 B per stage
template <bool CONV>
__global__ __launch_bounds__(BIG_THREADS, 2) void gemm_tn_big_k(const bf16_t* __restrict__ A, int64_t lda,
                                                                  const bf16_t* __restrict__ Bm, int64_t ldb,
                                                                  float* __restrict__ C, int64_t ldc, int M, int N,
                                                                  int R, int steps_per_split, ConvGeom cg, int64_t part_stride) {
  // part_stride > 0: split s stores its partial [M, N] image with plain stores at C + s * part_stride (every split owns at least
  // one step, so every image is written in full; the caller folds the images in a fixed order: bit-reproducible, and a plain
  // 128-byte row store costs a fraction of 32 f32 atomics); 0: the splits meet in C through f32 atomics
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // block -> (A-panel unit = (split, m-tile), n-tile): the tiles_n workgroups that read the same A panel get block
  // ids congruent mod 8, i.e. land on ONE XCD and share the panel in its L2 (one HBM read instead of tiles_n)
  const int tiles_n = N / WBN, tiles_m = (M + WBM - 1) / WBM;  // (the last m-tile may be ragged: M % 8 == 0)
  const int grp = blockIdx.x / (8 * tiles_n), rem = blockIdx.x - grp * 8 * tiles_n;
  const int unit = grp * 8 + (rem & 7), nt = rem >> 3;
  const int split = unit / tiles_m, mt = unit - split * tiles_m;
  const int m0 = mt * WBM, n0 = nt * WBN;
  const int nsteps_total = R / BK;
  const int s_begin = split * steps_per_split;
  int s_end = s_begin + steps_per_split;
  s_end = s_end < nsteps_total ? s_end : nsteps_total;
  if (s_begin >= s_end) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // DMA: 48 A chunks + 16 B chunks of 1 KiB per stage, 8 per wave (waves 0-5: A, waves 6-7: B)
  int64_t src_off[8];
  int lds_off[8];
  int crow[8], cdy[8], cdx[8];  // CONV: row inside the 64-row step and tap displacement of each chunk of this lane
  const bool is_a = wave < 6;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = is_a ? wave * 8 + i : (wave - 6) * 8 + i;
    const int pitch = is_a ? WBM * 2 : WBN * 2;
    const int o = c * 1024 + lane * 16;
    const int r = o / pitch, s = (o - r * pitch) >> 4;
    const int q = s ^ ((r & 3) << 2);
    // ragged last m-tile: lanes whose 8 columns lie beyond M re-read the tile's last valid chunk (their products land in
    // output rows >= M, which are never stored)
    const int qv = (!CONV && is_a && m0 + q * 8 >= M) ? (M - m0) / 8 - 1 : q;
    src_off[i] = (int64_t)r * (is_a ? lda : ldb) + qv * 8;
    lds_off[i] = (is_a ? 0 : WBM * 2 * BK) + c * 1024;
    crow[i] = r;
    cdy[i] = cdx[i] = 0;
    if (CONV && is_a) {
      // implicit im2col: column (m0 + 8q) of cols is (tap, ci); Ci % 8 == 0 keeps the 8 columns of a lane inside one tap
      const int col = m0 + q * 8;
      const int tap = col / cg.Ci, ci = col - tap * cg.Ci;
      cdy[i] = tap / 3 - 1;
      cdx[i] = tap % 3 - 1;
      src_off[i] = ((int64_t)r + (int64_t)cdy[i] * cg.W + cdx[i]) * cg.ldx + ci;
    }
  }
  const bf16_t* gsrc = is_a ? (CONV ? A : A + m0) : Bm + n0;
  const int64_t gld = is_a ? (CONV ? cg.ldx : lda) : ldb;
  auto stage = [&](int st, int buf) {
    char* base = smem + buf * W_STAGE;
    const bf16_t* p = gsrc + (int64_t)st * BK * gld;
    if (CONV && is_a) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t pix = (int64_t)st * BK + crow[i];
        int px, py;
        cg.pixel(pix, py, px);
        const bool ok = pix < cg.npix && py + cdy[i] >= 0 && py + cdy[i] < cg.H && px + cdx[i] >= 0 && px + cdx[i] < cg.W;
        glds16(ok ? p + src_off[i] : cg.zero, base + lds_off[i]);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) glds16(p + src_off[i], base + lds_off[i]);
  };

  const int li = lane & 15, g = lane >> 4;
  // byte offset of the transposing read for MFMA tile column base `cb`, sub-step kk, half, in an image of `pitch` bytes/row
  auto tr_off = [&](int cb, int kk, int half, int pitch) -> int {
    const int r = kk * 16 + (g >> 1) * 8 + half * 4 + (li >> 2);
    const int col = cb + (g & 1) * 16 + (li & 3) * 4;
    const int slot = (col >> 3) ^ ((r & 3) << 2);
    return r * pitch + slot * 16 + (col & 7) * 2;
  };

  f32x16_t acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  stage(s_begin, 0);
  for (int st = s_begin; st < s_end; ++st) {
    const int it = st - s_begin;
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (st + 1 < s_end) stage(st + 1, (it + 1) & 1);
    const char* ta = smem + (it & 1) * W_STAGE;
    const char* tb = ta + WBM * 2 * BK;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8_t af[3], bfg[2];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        union {
          s16x4_t h[2];
          bf16x8_t v;
        } u;
        u.h[0] = lds_tr16(ta + tr_off(wm * 96 + i * 32, kk, 0, WBM * 2));
        u.h[1] = lds_tr16(ta + tr_off(wm * 96 + i * 32, kk, 1, WBM * 2));
        af[i] = u.v;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        union {
          s16x4_t h[2];
          bf16x8_t v;
        } u;
        u.h[0] = lds_tr16(tb + tr_off(wn * 64 + j * 32, kk, 0, WBN * 2));
        u.h[1] = lds_tr16(tb + tr_off(wn * 64 + j * 32, kk, 1, WBN * 2));
        bfg[j] = u.v;
      }
      tr_landed(af, bfg);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfg[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 96 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M) {
          if (part_stride > 0) C[(int64_t)split * part_stride + (int64_t)m * ldc + n] = acc[i][j][r];
          else unsafeAtomicAdd(&C[(int64_t)m * ldc + n], acc[i][j][r]);
        }
      }
    }
}


int launch_tn_w4(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N, int64_t R,
                 int max_workgroups, hipStream_t stream);  // gemm_w4.hip
// How many ranges to split the contraction into when the ranges meet in C through f32 atomics.  Round 6 (profiles/r06_p_*): a
// workgroup's read-modify-writes drain at ~1.45 TB/s over the whole chip whatever the contention (256 x 295 KB: 52 us), a fifth of
// the plain-store rate, so "enough workgroups to fill the chip" is the wrong goal for small problems: M = N = 1024 over 2048 rows
// (the UNet's 4 x 4 attention linears) ran as 1024 workgroups of TWO k-steps and 67 MB of atomics, 54 us for 4 GFLOP.  Model:
// rounds(S) x ceil(nsteps / S) x step_us + S x image bytes / 1.45 TB/s; the smallest S within 3 % of the minimum.
static bool g_tn_split_model = true;  // LAB switch (not in the header): false = the workgroup-count rules of rounds 1-5
extern "C" __attribute__((visibility("default"))) void dl_lab_set_tn_split_model(int on) { g_tn_split_model = on != 0; }
static int g_tn_force_splits = 0;  // LAB: > 0 = this split count (clamped to the k-steps) instead of the model's
extern "C" __attribute__((visibility("default"))) void dl_lab_set_tn_force_splits(int n) { g_tn_force_splits = n; }
static int tn_pick_splits(int64_t ntile, int nsteps, int64_t out_elems, int wg_slots, double step_us, int max_splits) {
  if (g_tn_force_splits > 0) return g_tn_force_splits < nsteps ? g_tn_force_splits : nsteps;
  const double image_us = (double)out_elems * 4.0 / 1.45e6;
  int best = 1;
  double best_t = 1e30;
  if (max_splits > nsteps) max_splits = nsteps;
  for (int S = 1; S <= max_splits; ++S) {
    const int sps = (nsteps + S - 1) / S;
    if ((nsteps + sps - 1) / sps != S) continue;  // (not a split count the step rounding produces)
    const int64_t rounds = (ntile * S + wg_slots - 1) / wg_slots;
    const double t = (double)rounds * sps * step_us + S * image_us;
    if (t < best_t * 0.97) best_t = t, best = S;
  }
  return best;
}
extern "C" int dl_gemm_tn_ex(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                             int64_t N, int64_t R, int max_workgroups, dl_stream_t stream);
extern "C" int dl_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                          int64_t N, int64_t R, dl_stream_t stream) {
  return dl_gemm_tn_ex(A, lda, B, ldb, C, ldc, M, N, R, 0, stream);
}
extern "C" int dl_gemm_tn_ex(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                             int64_t N, int64_t R, int max_workgroups, dl_stream_t stream) {
  DL_CHECK_ARG(A && B && C && M > 0 && N > 0 && R > 0, "dl_gemm_tn: null/empty operand");
  DL_CHECK_ARG(R % BK == 0, "dl_gemm_tn: R=%lld must be a multiple of %d", (long long)R, BK);
  DL_CHECK_ARG(M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= M && ldb >= N && ldc >= N,
               "dl_gemm_tn: M,N,lda,ldb must be multiples of 8 (M=%lld N=%lld)", (long long)M, (long long)N);
  DL_CHECK_ARG((((uintptr_t)A | (uintptr_t)B) & 15) == 0, "dl_gemm_tn: 16-byte alignment");
  {
    static DevOnce once;
    const int n_cu = dev_cus(once, [] {
      (void)hipFuncSetAttribute((const void*)gemm_tn_big_k<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W_STAGE);
    });
    {  // 384 x 192 ring kernel (gemm_w4.hip) for the shapes its tile divides
      const int rc = launch_tn_w4(A, lda, B, ldb, C, ldc, M, N, R, max_workgroups, (hipStream_t)stream);
      if (rc <= 0) return rc;
    }
    const int nsteps = (int)(R / BK);
    const int64_t tiles_m64 = (M + WBM - 1) / WBM;
    if (N % WBN == 0 && nsteps >= 64 && 5 * M >= 3 * tiles_m64 * WBM) {  // (ragged last m-tile: >= 60 % useful)
      // one workgroup per CU at most (128 KiB of LDS each): units (= m-tiles x splits) are padded to a multiple of
      // 8 for the XCD mapping, so pick the split count from the padded budget
      const int tiles_m = (int)tiles_m64, tiles_n = (int)(N / WBN);
      // max_workgroups > 0: the caller runs this GEMM beside other work (the engines' side-stream wgrads) and wants some
      // CUs left unclaimed by the persistent workgroups, so the latency-bound kernels of the main chain keep full occupancy there
      const int budget = (max_workgroups > 0 && max_workgroups < n_cu) ? max_workgroups : n_cu;
      int padded_max = (budget / tiles_n) & ~7;
      if (padded_max < 8) padded_max = 8;
      int splits = padded_max / tiles_m;
      if (splits < 1) splits = 1;
      if (splits > nsteps / 8) splits = nsteps / 8;
      if (g_tn_split_model) splits = tn_pick_splits((int64_t)tiles_m * tiles_n, nsteps, M * N, budget, 0.9, splits);
      const int sps = (nsteps + splits - 1) / splits;
      splits = (nsteps + sps - 1) / sps;
      const int units = (tiles_m * splits + 7) & ~7;  // surplus units exit at once
      hipLaunchKernelGGL(gemm_tn_big_k<false>, units * (int)(N / WBN), BIG_THREADS, 2 * W_STAGE, (hipStream_t)stream,
                         (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, (int)M, (int)N, (int)R, sps, ConvGeom{}, (int64_t)0);
      DL_LAUNCH_CHECK();
      return DL_OK;
    }
  }
  const int ntile = cdiv(M, BM) * cdiv(N, BN);
  const int nsteps = (int)(R / BK);
  int splits = (1024 + ntile - 1) / ntile;  // aim at >= 4 workgroups per CU
  if (splits > nsteps) splits = nsteps;
  if (splits < 1) splits = 1;
  if (g_tn_split_model) splits = tn_pick_splits(ntile, nsteps, M * N, 512, 0.4, splits);  // (two workgroups per CU: 64 KiB of LDS each)
  const int sps = (nsteps + splits - 1) / splits;
  splits = (nsteps + sps - 1) / sps;
  hipLaunchKernelGGL(gemm_tn_k<false>, ntile * splits, NT_THREADS, 65536, (hipStream_t)stream, (const bf16_t*)A, lda,
                     (const bf16_t*)B, ldb, C, ldc, (int)M, (int)N, (int)R, sps, ConvGeom{}, (int64_t)0);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ----------------------------------------------------------------------------------------------------- deterministic small GEMMs
// The 128x128 kernels split their contraction over workgroups to fill the chip on skinny problems (the head's [16, 384] weight
// gradient over 65536 tokens, the conditioning path's [256, 384] product over K = 28416); by default the splits meet in f32 atomics,
// whose order varies from run to run.  The *_det entry points give every split its own partial image in a caller-owned scratch
// (plain stores) and fold the images in a fixed order -- two runs are bit-identical.
// S = split lanes per element (threads that share one output element and walk the partial images S apart; combined in LDS in a
// fixed tree): 1 for a handful of images of a large matrix, 16 for hundreds of images of a few dozen elements (a bias gradient
// over 512 row slabs: one thread per element would issue 512 dependent-latency loads)
template <int S>
__global__ __launch_bounds__(256) void fold_partials_k(const float* __restrict__ part, int64_t stride, int splits, int64_t M, int N,
                                                       int64_t ldp, float* __restrict__ C, int64_t ldc, int accumulate) {
  constexpr int E = 256 / S;
  __shared__ float red[S][E];
  const int el = threadIdx.x % E, sl = threadIdx.x / E;
  const int64_t total = M * N;
  for (int64_t base = (int64_t)blockIdx.x * E; base < total; base += (int64_t)gridDim.x * E) {
    const int64_t i = base + el;
    float s = 0.f;
    int64_t m = 0;
    int n = 0;
    if (i < total) {
      m = i / N;
      n = (int)(i - m * N);
      const float* p = part + m * ldp + n;
      for (int k = sl; k < splits; k += S) s += p[(int64_t)k * stride];
    }
    if (S > 1) {
      red[sl][el] = s;
      __syncthreads();
      if (sl == 0) {
#pragma unroll
        for (int w = S / 2; w >= 1; w >>= 1)  // fixed tree over the lanes: ((0+1)+(2+3))+...
#pragma unroll
          for (int q = 0; q < w; ++q) red[q][el] = red[2 * q][el] + red[2 * q + 1][el];
        s = red[0][el];
      }
    }
    if (sl == 0 && i < total) {
      float* c = C + m * ldc + n;
      *c = accumulate ? *c + s : s;
    }
    if (S > 1) __syncthreads();
  }
}
int launch_fold_partials(const float* part, int64_t stride, int splits, int64_t M, int64_t N, int64_t ldp, float* C, int64_t ldc,
                         int accumulate, hipStream_t stream) {
  const int64_t total = M * N;
#define FOLD_GO(S_)                                                                                                            \
  do {                                                                                                                         \
    int64_t g = (total + 256 / S_ - 1) / (256 / S_);                                                                           \
    if (g > 4096) g = 4096;                                                                                                    \
    hipLaunchKernelGGL(fold_partials_k<S_>, (int)g, 256, 0, stream, part, stride, splits, M, (int)N, ldp, C, ldc, accumulate); \
  } while (0)
  if (splits <= 4) FOLD_GO(1);
  else if (splits <= 32) FOLD_GO(4);
  else FOLD_GO(16);
#undef FOLD_GO
  return hipGetLastError() == hipSuccess ? DL_OK : DL_ERR_LAUNCH;
}

extern "C" int dl_gemm_tn_det(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N,
                              int64_t R, float* scratch, int64_t scratch_floats, dl_stream_t stream) {
  DL_CHECK_ARG(A && B && C && scratch && M > 0 && N > 0 && R > 0, "dl_gemm_tn_det: null/empty operand");
  DL_CHECK_ARG(R % BK == 0, "dl_gemm_tn_det: R=%lld must be a multiple of %d", (long long)R, BK);
  DL_CHECK_ARG(M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= M && ldb >= N && ldc >= N,
               "dl_gemm_tn_det: M,N,lda,ldb must be multiples of 8 (M=%lld N=%lld)", (long long)M, (long long)N);
  DL_CHECK_ARG((((uintptr_t)A | (uintptr_t)B) & 15) == 0, "dl_gemm_tn_det: 16-byte alignment");
  DL_CHECK_ARG(scratch_floats >= M * N, "dl_gemm_tn_det: scratch of %lld floats < one [M, N] image", (long long)scratch_floats);
  if (((M % 384 == 0 && N % 192 == 0) || (M % 256 == 0 && N % 256 == 0)) && R % 32 == 0 && R / 32 >= 64) {  // the ring kernel's atomics-free form
    const dl_wgrad_t one{A, lda, B, ldb, C, M, N};
    if (ldc == N) return dl_gemm_tn_group(&one, 1, R, scratch, scratch_floats, 0, stream);
  }
  const int ntile = cdiv(M, BM) * cdiv(N, BN);
  const int nsteps = (int)(R / BK);
  int splits = (1024 + ntile - 1) / ntile;  // aim at >= 4 workgroups per CU
  if (splits > nsteps) splits = nsteps;
  if ((int64_t)splits * M * N > scratch_floats) splits = (int)(scratch_floats / (M * N));
  if (splits < 1) splits = 1;
  const int sps = (nsteps + splits - 1) / splits;
  splits = (nsteps + sps - 1) / sps;  // every split owns at least one step
  hipLaunchKernelGGL(gemm_tn_k<false>, ntile * splits, NT_THREADS, 65536, (hipStream_t)stream, (const bf16_t*)A, lda,
                     (const bf16_t*)B, ldb, scratch, N, (int)M, (int)N, (int)R, sps, ConvGeom{}, (int64_t)(M * N));
  DL_LAUNCH_CHECK();
  return launch_fold_partials(scratch, M * N, splits, M, N, N, C, ldc, 1, (hipStream_t)stream);
}

extern "C" int dl_gemm_nt_f32_det(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N,
                                  int64_t K, float* scratch, int64_t scratch_floats, dl_stream_t stream) {
  DL_CHECK_ARG(A && B && C && scratch && M > 0 && N > 0 && K > 0, "dl_gemm_nt_f32_det: null/empty operand");
  DL_CHECK_ARG(K % BK == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K && ldc >= N,
               "dl_gemm_nt_f32_det: K %% 64, leading dims (lda=%lld ldb=%lld ldc=%lld)", (long long)lda, (long long)ldb, (long long)ldc);
  DL_CHECK_ARG((((uintptr_t)A | (uintptr_t)B) & 15) == 0 && M < (1ll << 31) && N < (1ll << 31), "dl_gemm_nt_f32_det: alignment / dims");
  const int nwg = cdiv(M, BM) * cdiv(N, BN);
  int ksplit = 1;
  if (nwg < 64 && K >= 2048) {
    ksplit = 256 / nwg;
    if (ksplit > K / 512) ksplit = (int)(K / 512);
    if ((int64_t)ksplit * M * N > scratch_floats) ksplit = (int)(scratch_floats / (M * N));
    if (ksplit < 1) ksplit = 1;
  }
  NtEpilogue ep{};
  ep.out_f32 = 1;
  ep.rows_per_gate = 1;
  if (ksplit == 1) {
    hipLaunchKernelGGL(gemm_nt_k<false>, dim3(nwg, 1), NT_THREADS, NT_LDS_BYTES, (hipStream_t)stream, (const bf16_t*)A, lda,
                       (const bf16_t*)B, ldb, C, ldc, (int)M, (int)N, (int)K, ep, 1, ConvGeom{});
    DL_LAUNCH_CHECK();
    return DL_OK;
  }
  ep.part_stride = M * N;
  hipLaunchKernelGGL(gemm_nt_k<false>, dim3(nwg, ksplit), NT_THREADS, NT_LDS_BYTES, (hipStream_t)stream, (const bf16_t*)A, lda,
                     (const bf16_t*)B, ldb, scratch, N, (int)M, (int)N, (int)K, ep, ksplit, ConvGeom{});
  DL_LAUNCH_CHECK();
  return launch_fold_partials(scratch, M * N, ksplit, M, N, N, C, ldc, 0, (hipStream_t)stream);
}

// ----------------------------------------------------------------------------------------------------- implicit-GEMM 3x3 conv
/* out[p, co] = bias[co] + sum_{tap, ci} x[p + shift(tap), ci] * Wf[co, tap*Ci + ci] (+ resid[p, co]); x NHWC rows [B*H*W, ldx].
 * Wf is the (tap, ci)-ordered shadow of dl_cast_conv3x3_weight (the rotated one gives the data gradient). */
// =====================================================================================================
// conv3x3_big_k: the 3x3 / pad-1 convolution (forward, and data gradient with the rotated shadow) as a PERSISTENT implicit GEMM on
// 256 x TN tiles (unet.py:187,208,594,745: nn.Conv2d(Ci, Co, 3, padding=1) over NHWC rows).  The 128x128 kernel above stages 64 FLOP
// per byte and, at the low-resolution levels of the UNet (8x8 / 4x4 pixels: 8192 / 2048 output rows against contractions of
// 4608 ... 18432), has a few hundred workgroups of 72 ... 288 k-steps each; here
//   * a work item is (row tile, column tile, k-split): out[256 x TN] over k-steps [k_lo, k_hi); the items of a launch are walked by
//     one 512-thread workgroup per CU (8 waves as 4 x 2, 64 x TN/2 each), items that share a row panel are neighbours on one XCD;
//   * the A operand is gathered by the DMA itself: a 64-deep k-step lies inside one tap (Ci % 64 == 0), every lane's 16 bytes come
//     from x[p + shift(tap), ci0 ..] or from the caller's zero line when the tap leaves the image -- the in-image bits of a lane's
//     pixel are recomputed per item, nothing of the geometry lives in memory;
//   * two-slot LDS ring, fragment software pipeline and register epilogue of gemm_nt_big_k (TN = 256: 128 FLOP per staged byte);
//   * k-split items store their f32 partial tile at C + split * part_stride (plain stores, conv_splitk_finalize_k adds the images in
//     a fixed order with bias / residual: no atomics), unsplit items run the full epilogue (bias, residual, bf16).
// Requires M % 256 == 0, N % TN == 0, Ci % 64 == 0.
// =====================================================================================================
// epilogue of conv3x3_big_k straight from the accumulator registers (layout of nt_epilogue_regs: one v_permlane32_swap per register
// pair gives every lane 8 consecutive columns of one row): F32 = partial image of a k-split (plain f32 stores), else
// bf16(acc + bias + resid).  Zeroes the accumulators.
template <int JN, bool F32>
__device__ __forceinline__ void conv_epilogue(f32x16_t (&acc)[JN][2], int m_base, int n_base, int lane, void* C, int64_t ldc,
                                              const float* __restrict__ bias, const bf16_t* __restrict__ resid, int64_t ldr) {
  const int hi = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t m = m_base + i * 32 + (lane & 31);
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j][i][8 * gp + e]), __float_as_uint(acc[j][i][8 * gp + 4 + e]), false, false);
          v[e] = __uint_as_float(sw[0]);
          v[4 + e] = __uint_as_float(sw[1]);
          acc[j][i][8 * gp + e] = 0.f;
          acc[j][i][8 * gp + 4 + e] = 0.f;
        }
        const int n = n_base + j * 32 + 16 * gp + 8 * hi;
        if (F32) {
          float* cp = (float*)C + m * ldc + n;
          *(f32x4_t*)cp = *(f32x4_t*)&v[0];
          *(f32x4_t*)(cp + 4) = *(f32x4_t*)&v[4];
        } else {
          if (bias) {
            const f32x4_t b0 = *(const f32x4_t*)(bias + n), b1 = *(const f32x4_t*)(bias + n + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] += b0[e];
              v[4 + e] += b1[e];
            }
          }
          if (resid) {
            float rr[8];
            unpack8(*(const u32x4_t*)(resid + m * ldr + n), rr);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rr[e];
          }
          *(u32x4_t*)((bf16_t*)C + m * ldc + n) = pack8(v);
        }
      }
  }
}
static int g_conv_big = 1;
extern "C" __attribute__((visibility("default"))) void dl_lab_set_conv_big(int on) { g_conv_big = on; }  // LAB A/B switch (not in the header)
static bool conv_big_enabled() { return g_conv_big != 0; }
template <int TN_>
__global__ __launch_bounds__(BIG_THREADS, 2) void conv3x3_big_k(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wf, int64_t ldw,
                                                                  void* __restrict__ C, int64_t ldc, int M, int N, int K, NtEpilogue ep,
                                                                  int ksplit, ConvGeom cg) {
  constexpr int STAGE = (TBM + TN_) * 128;
  constexpr int CHB = TN_ / 64;  // weight-row DMA chunks per wave and stage (the activation rows: 4)
  constexpr int JN = TN_ / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = N / TN_, ntiles = (M / TBM) * tiles_n;
  const int nk_all = K / BK, kc = cg.Ci / BK;  // k-steps per tap
  const int nitems = ntiles * ksplit;
  const int G = gridDim.x;
  const int slot0 = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);

  // item id -> (row tile tm, split s, column tile tn): n fastest, then the splits of a row panel (they share the panel's pixels)
  auto item_k = [&](int s, int& k_lo, int& k_hi) {
    k_lo = (int)((int64_t)nk_all * s / ksplit);
    k_hi = (int)((int64_t)nk_all * (s + 1) / ksplit);
  };
  // ---- DMA cursor (one stage ahead of the compute cursor), with the per-item gather state of this lane's four pixel rows
  int s_item = slot0, s_k = 0, s_khi = 0, s_it = 0;
  const bf16_t* a_src[4];
  int tap_ok[4];
  const bf16_t* b_src[CHB];
  auto open_item = [&](int item) {
    const int tn = item % tiles_n, rest = item / tiles_n, sp = rest % ksplit, tm = rest / ksplit;
    item_k(sp, s_k, s_khi);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (wave * 4 + i) * 8 + (lane >> 3);
      const int q = (lane & 7) ^ ((r >> 1) & 7);
      const int64_t p = (int64_t)tm * TBM + r;
      a_src[i] = X + p * cg.ldx + q * 8;
      int px, py;
      cg.pixel(p, py, px);
      int ok = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
        if (yy >= 0 && yy < cg.H && xx >= 0 && xx < cg.W && p < cg.npix) ok |= 1 << t;
      }
      tap_ok[i] = ok;
    }
#pragma unroll
    for (int i = 0; i < CHB; ++i) {
      const int r = (wave * CHB + i) * 8 + (lane >> 3);
      const int q = (lane & 7) ^ ((r >> 1) & 7);
      b_src[i] = Wf + (int64_t)(tn * TN_ + r) * ldw + q * 8;
    }
  };
  auto stage_next = [&]() {
    char* base = smem + (s_it & 1) * STAGE;
    const int tap = s_k / kc, ci0 = (s_k - tap * kc) * BK;  // wave-uniform
    const int64_t shift = ((int64_t)(tap / 3 - 1) * cg.W + (tap % 3 - 1)) * cg.ldx + ci0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      glds16(((tap_ok[i] >> tap) & 1) ? a_src[i] + shift : cg.zero, base + (wave * 4 + i) * 1024);
#pragma unroll
    for (int i = 0; i < CHB; ++i) glds16(b_src[i] + (int64_t)s_k * BK, base + TBM * 128 + (wave * CHB + i) * 1024);
    ++s_it;
    if (++s_k == s_khi) {
      s_item += G;
      if (s_item < nitems) open_item(s_item);
    }
  };

  int xrow[2], wrow[JN];
#pragma unroll
  for (int i = 0; i < 2; ++i) xrow[i] = wm * 64 + i * 32 + (lane & 31);
#pragma unroll
  for (int j = 0; j < JN; ++j) wrow[j] = wn * (TN_ / 2) + j * 32 + (lane & 31);
  f32x16_t acc[JN][2];
#pragma unroll
  for (int j = 0; j < JN; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  if (s_item < nitems) {
    open_item(s_item);
    stage_next();
  }
  int it = 0;
  for (int item = slot0; item < nitems; item += G) {
    const int tn = item % tiles_n, rest = item / tiles_n, sp = rest % ksplit, tm = rest / ksplit;
    int k_lo, k_hi;
    item_k(sp, k_lo, k_hi);
    for (int kt = k_lo; kt < k_hi; ++kt, ++it) {
      wait_vmcnt<0>();               // stage `it` has landed (two-slot ring: nothing younger is in flight)
      __builtin_amdgcn_s_barrier();  // every wave's share landed, the other slot's readers are done
      if (s_item < nitems) stage_next();
      const char* sa = smem + (it & 1) * STAGE;
      const char* sb = sa + TBM * 128;
      bf16x8_t xq[2][2], wq[3];
      auto rd_x = [&](int kk, int i) -> bf16x8_t {
        return *(const bf16x8_t*)(sa + xrow[i] * 128 + ((((kk << 1) | hi) ^ ((xrow[i] >> 1) & 7)) << 4));
      };
      auto rd_w = [&](int kk, int j) -> bf16x8_t {
        return *(const bf16x8_t*)(sb + wrow[j] * 128 + ((((kk << 1) | hi) ^ ((wrow[j] >> 1) & 7)) << 4));
      };
      xq[0][0] = rd_x(0, 0);
      xq[0][1] = rd_x(0, 1);
      wq[0] = rd_w(0, 0);
      wq[1] = rd_w(0, 1);
#pragma unroll
      for (int s = 0; s < 4 * JN; ++s) {
        const int kk = s / JN, j = s % JN;
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < 4 * JN) wq[(s + 2) % 3] = rd_w((s + 2) / JN, (s + 2) % JN);
        if (kk < 3 && j < 2) xq[(kk + 1) & 1][j] = rd_x(kk + 1, j);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[s % 3], xq[kk & 1][i], acc[j][i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue of the item straight from registers (zeroes the accumulators)
    if (ksplit > 1)
      conv_epilogue<JN, true>(acc, tm * TBM + wm * 64, tn * TN_ + wn * (TN_ / 2), lane, (float*)C + (int64_t)sp * ep.part_stride, N, nullptr,
                              nullptr, 0);
    else
      conv_epilogue<JN, false>(acc, tm * TBM + wm * 64, tn * TN_ + wn * (TN_ / 2), lane, C, ldc, ep.bias, ep.resid, ep.ldr);
    wait_vmcnt<0>();  // unknown number of epilogue memory operations: drain (the next stage's DMA is re-counted from zero)
  }
}
template <int TN_>
static int launch_conv_big(const void* x, const void* Wf, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                           const NtEpilogue& ep, int ksplit, const ConvGeom& cg, hipStream_t stream) {
  constexpr int LDS = 2 * (TBM + TN_) * 128;
  static DevOnce once;
  const int n_cu = dev_cus(once, [] { (void)hipFuncSetAttribute((const void*)conv3x3_big_k<TN_>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); });
  const int64_t items = (M / TBM) * (N / TN_) * ksplit;
  int grid = (int)(items < n_cu ? items : n_cu) & ~7;
  if (grid < 8) return 1;
  hipLaunchKernelGGL((conv3x3_big_k<TN_>), grid, BIG_THREADS, LDS, stream, (const bf16_t*)x, (const bf16_t*)Wf, ldw, C, ldc, (int)M, (int)N,
                     (int)K, ep, ksplit, cg);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// =====================================================================================================
// conv3x3_halo_k: conv3x3_big_k with the activation operand staged ONCE per 64-channel chunk instead of once per tap.  A 256-row
// tile is whole image rows (W | 256: `nrow` rows of one image, or 256 / (H W) whole images), so the nine taps of a channel chunk read
// nine SHIFTED windows of the same pixels: the tile's rows plus a one-pixel border -- the halo, (nrow + 2) x (W + 2) pixels per image
// segment, 324-400 rows of 128 bytes instead of 9 x 256 -- are staged once (zero line outside the image) and every tap's MFMA
// fragments are read from it at a per-lane row offset.  The k-steps run channel-chunk-major, tap-minor (the weight shadow is addressed
// accordingly: column tap * Ci + ci0); a k-step stages only its TN x 64 weight tile plus one 1 KiB chunk per wave of the NEXT channel
// chunk's halo, which lands in the second halo buffer while the nine taps of the current one compute: 21-37 KiB per k-step against
// the 48-64 KiB of conv3x3_big_k -- the loop is bound by what a CU can pull out of L2 while its matrix pipe runs.
// No k-split (the launch must have enough tiles); M % 256 == 0, 256 % W == 0, H W | 256 or 256 | H W, halo <= HALO_MAX rows.
// =====================================================================================================
#define HALO_MAX_CHUNKS 50  // 400 halo rows (8 x 8 images, four per tile): 72 chunk slots per channel chunk (9 taps x 8 waves)
template <int TN_, int NST>
__global__ __launch_bounds__(BIG_THREADS, 2) void conv3x3_halo_k(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wf, int64_t ldw,
                                                                   void* __restrict__ C, int64_t ldc, int M, int N, NtEpilogue ep,
                                                                   ConvGeom cg, int nrow, int nseg, int nchunks) {
  constexpr int CHB = TN_ / 64, JN = TN_ / 64;
  constexpr int BSTAGE = TN_ * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int HB = nchunks * 1024;  // bytes of one halo buffer
  char* const hbuf = smem;
  char* const bbuf = smem + 2 * HB;
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = N / TN_, nitems = (M / TBM) * tiles_n;
  const int kc = cg.Ci / BK;  // channel chunks; 9 k-steps each
  const int G = gridDim.x;
  const int slot0 = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
  const int W2 = cg.W + 2, seg_rows = (nrow + 2) * W2, HP = nseg * seg_rows;

  // ---- halo geometry of this lane's DMA rows: slot t of a channel chunk is chunk c = t * 8 + wave, LDS row hr = c * 8 + lane / 8
  int hsrc[9];  // element offset into X of slot t's source for the halo being fetched (channel offset added at issue); -1: zero line
  auto open_halo = [&](int tm) {
    const int64_t p0 = (int64_t)tm * TBM;  // first pixel of the tile
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int hr = (t * 8 + wave) * 8 + (lane >> 3);
      const int q = (lane & 7) ^ ((hr >> 1) & 7);
      int src = -1;
      if (hr < HP) {
        const int seg = hr / seg_rows, rem = hr - seg * seg_rows;
        const int hy = rem / W2, hx = rem - hy * W2;
        int64_t img0;  // first pixel of the image the segment belongs to
        int py;
        if (nseg > 1) {  // whole images: segment = image
          img0 = p0 + (int64_t)seg * cg.H * cg.W;
          py = hy - 1;
        } else {  // nrow rows of one image
          const int64_t hw = (int64_t)cg.H * cg.W;
          img0 = p0 / hw * hw;
          py = (int)((p0 - img0) / cg.W) + hy - 1;
        }
        const int px = hx - 1;
        if (py >= 0 && py < cg.H && px >= 0 && px < cg.W && img0 < cg.npix) src = (int)((img0 + (int64_t)py * cg.W + px) * cg.ldx + q * 8);
      }
      hsrc[t] = src;
    }
  };
  // ---- DMA cursor: stream step s = 9 * (halo number) + tap; stage s = weight tile of (item, cc, tap) + slot `tap` of the NEXT halo
  int s_item = slot0, s_cc = 0, s_tap = 0, s_it = 0;  // weights
  int h_item = slot0, h_cc = 0;                        // the halo being fetched (one channel chunk ahead of the weights)
  const bf16_t* b_src[CHB];
  auto open_b = [&](int item) {
    const int tn = item % tiles_n;
#pragma unroll
    for (int i = 0; i < CHB; ++i) {
      const int r = (wave * CHB + i) * 8 + (lane >> 3);
      const int q = (lane & 7) ^ ((r >> 1) & 7);
      b_src[i] = Wf + (int64_t)(tn * TN_ + r) * ldw + q * 8;
    }
  };
  auto issue_halo_slot = [&](int t, int buf) {  // (t is a compile-time constant at every call site: hsrc stays in registers)
    if (t * 8 + wave < nchunks) glds16(hsrc[t] >= 0 ? X + hsrc[t] + h_cc * BK : cg.zero, hbuf + buf * HB + (t * 8 + wave) * 1024);
  };
  auto advance_halo = [&]() {  // successor of (h_item, h_cc) in stream order
    if (++h_cc == kc) {
      h_cc = 0;
      const int prev_tm = h_item / tiles_n;
      h_item += G;
      if (h_item < nitems && h_item / tiles_n != prev_tm) open_halo(h_item / tiles_n);
    }
  };
  auto stage_next = [&]() {
    char* base = bbuf + (s_it % NST) * BSTAGE;
    const int64_t col = (int64_t)s_tap * cg.Ci + s_cc * BK;
#pragma unroll
    for (int i = 0; i < CHB; ++i) glds16(b_src[i] + col, base + (wave * CHB + i) * 1024);
    // the next halo, three chunk slots with the stages of taps 3, 5 and 7 (static slot indices: hsrc stays in registers).  NOT with
    // taps 0 .. NST-2: those stages are issued while the last taps of the previous channel chunk still read the buffer the next halo
    // goes into (a stage runs NST - 1 <= 3 steps ahead of the compute).
    if (h_item < nitems) {
      const int buf = ((s_it / 9) + 1) & 1;
      if (s_tap == 3) {
        issue_halo_slot(0, buf);
        issue_halo_slot(1, buf);
        issue_halo_slot(2, buf);
      } else if (s_tap == 5) {
        issue_halo_slot(3, buf);
        issue_halo_slot(4, buf);
        issue_halo_slot(5, buf);
      } else if (s_tap == 7) {
        issue_halo_slot(6, buf);
        issue_halo_slot(7, buf);
        issue_halo_slot(8, buf);
      }
    }
    ++s_it;
    if (++s_tap == 9) {
      s_tap = 0;
      advance_halo();
      if (++s_cc == kc) {
        s_cc = 0;
        s_item += G;
        if (s_item < nitems) open_b(s_item);
      }
    }
  };

  // ---- fragment rows: tile row m -> centre halo row (the same for every tile)
  int hc[2], wrow[JN];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = wm * 64 + i * 32 + (lane & 31);
    const int seg = m / (nrow * cg.W), rm = m - seg * nrow * cg.W;
    hc[i] = seg * seg_rows + (rm / cg.W + 1) * W2 + (rm % cg.W) + 1;
  }
#pragma unroll
  for (int j = 0; j < JN; ++j) wrow[j] = wn * (TN_ / 2) + j * 32 + (lane & 31);
  f32x16_t acc[JN][2];
#pragma unroll
  for (int j = 0; j < JN; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  if (s_item < nitems) {  // prologue: the whole first halo, then the first weight stage (which also starts the second halo)
    open_halo(h_item / tiles_n);
#pragma unroll
    for (int t = 0; t < 9; ++t) issue_halo_slot(t, 0);
    advance_halo();
    open_b(s_item);
#pragma unroll
    for (int r = 0; r < NST - 1; ++r)
      if (s_item < nitems) stage_next();
  }
  int it = 0;
  for (int item = slot0; item < nitems; item += G) {
    const int tn = item % tiles_n, tm = item / tiles_n;
    for (int cc = 0; cc < kc; ++cc) {
      const char* ha = hbuf + ((it / 9) & 1) * HB;
      for (int tap = 0; tap < 9; ++tap, ++it) {
        // stage `it` has landed: at most the NST - 2 younger weight stages are still in flight (halo chunks among them only make this
        // wait for more); at the end of the stream fewer stages are in flight than that: drain
        if (s_item < nitems) wait_vmcnt<(NST - 2) * CHB>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();  // every wave's share landed, the slot about to be refilled has no readers left
        if (s_item < nitems) stage_next();
        const char* sb = bbuf + (it % NST) * BSTAGE;
        const int delta = (tap / 3 - 1) * W2 + (tap % 3 - 1);
        const int hr0 = hc[0] + delta, hr1 = hc[1] + delta;
        const char* x0 = ha + hr0 * 128;
        const char* x1 = ha + hr1 * 128;
        const int z0 = (hr0 >> 1) & 7, z1 = (hr1 >> 1) & 7;
        bf16x8_t xq[2][2], wq[3];
        auto rd_x = [&](int kk, int i) -> bf16x8_t {
          return *(const bf16x8_t*)((i ? x1 : x0) + ((((kk << 1) | hi) ^ (i ? z1 : z0)) << 4));
        };
        auto rd_w = [&](int kk, int j) -> bf16x8_t {
          return *(const bf16x8_t*)(sb + wrow[j] * 128 + ((((kk << 1) | hi) ^ ((wrow[j] >> 1) & 7)) << 4));
        };
        xq[0][0] = rd_x(0, 0);
        xq[0][1] = rd_x(0, 1);
        wq[0] = rd_w(0, 0);
        wq[1] = rd_w(0, 1);
#pragma unroll
        for (int s = 0; s < 4 * JN; ++s) {
          const int kk = s / JN, j = s % JN;
          __builtin_amdgcn_sched_barrier(0);
          if (s + 2 < 4 * JN) wq[(s + 2) % 3] = rd_w((s + 2) / JN, (s + 2) % JN);
          if (kk < 3 && j < 2) xq[(kk + 1) & 1][j] = rd_x(kk + 1, j);
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[s % 3], xq[kk & 1][i], acc[j][i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    conv_epilogue<JN, false>(acc, tm * TBM + wm * 64, tn * TN_ + wn * (TN_ / 2), lane, C, ldc, ep.bias, ep.resid, ep.ldr);
    wait_vmcnt<0>();
  }
}
static int g_conv_split128 = 1;  // LAB switch: 0 = split launches always take 256-wide tiles where Co allows (round 5)
extern "C" __attribute__((visibility("default"))) void dl_lab_set_conv_split128(int on) { g_conv_split128 = on; }
static int g_conv_halo_nst = 4;  // LAB: ring depth of the 128-wide halo kernel's weight stages
extern "C" __attribute__((visibility("default"))) void dl_lab_set_conv_halo_nst(int n) { g_conv_halo_nst = n; }
static int g_conv_halo = 1;  // LAB switch (not in the header): 0 = never, 1 = conv3x3_halo_k where the shape fits (default), 2 = also for small launches (tests)
extern "C" __attribute__((visibility("default"))) void dl_lab_set_conv_halo(int on) { g_conv_halo = on; }
template <int TN_, int NST>
static int launch_conv_halo(const void* x, const void* Wf, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N, const NtEpilogue& ep,
                            const ConvGeom& cg, int nrow, int nseg, int nchunks, hipStream_t stream) {
  const int LDS = 2 * nchunks * 1024 + NST * TN_ * 128;
  if (LDS > 160 * 1024) return 1;
  static DevOnce once;
  const int n_cu = dev_cus(once, [] { (void)hipFuncSetAttribute((const void*)conv3x3_halo_k<TN_, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
  const int64_t items = (M / TBM) * (N / TN_);
  int grid = (int)(items < n_cu ? items : n_cu) & ~7;
  if (grid < 8) return 1;
  hipLaunchKernelGGL((conv3x3_halo_k<TN_, NST>), grid, BIG_THREADS, LDS, stream, (const bf16_t*)x, (const bf16_t*)Wf, ldw, C, ldc, (int)M, (int)N, ep,
                     cg, nrow, nseg, nchunks);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// second half of a split-K convolution: out = bf16(acc + bias + resid)
__global__ void conv_splitk_finalize_k(const float* __restrict__ acc, int splits, int64_t stride, const float* __restrict__ bias,
                                       const bf16_t* __restrict__ resid, int64_t ldr, bf16_t* __restrict__ out, int64_t ldc,
                                       int64_t M, int N) {
  const int N8 = N >> 3;
  const int64_t n8 = M * N8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / N8;
    const int n = (int)(i - m * N8) * 8;
    float v[8], r[8];
    *(f32x4_t*)&v[0] = *(const f32x4_t*)(acc + m * N + n);
    *(f32x4_t*)&v[4] = *(const f32x4_t*)(acc + m * N + n + 4);
    for (int p = 1; p < splits; ++p) {  // the partial images in a fixed order
      const f32x4_t a = *(const f32x4_t*)(acc + p * stride + m * N + n), b = *(const f32x4_t*)(acc + p * stride + m * N + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += a[e];
        v[4 + e] += b[e];
      }
    }
    if (resid) unpack8(*(const u32x4_t*)(resid + m * ldr + n), r);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += (bias ? bias[n + e] : 0.f) + (resid ? r[e] : 0.f);
    *(u32x4_t*)(out + m * ldc + n) = pack8(v);
  }
}
extern "C" int dl_conv3x3_nt(const void* x, int64_t ldx, int64_t Bn, int64_t H, int64_t W, int64_t Ci, const void* Wf,
                             int64_t ldw, void* out, int64_t ldc, int64_t Co, const float* bias, const void* resid,
                             int64_t ldr, const void* zero, float* splitk_scratch, int64_t scratch_floats, dl_stream_t stream) {
  DL_CHECK_ARG(x && Wf && out && zero && Bn > 0 && H > 0 && W > 0 && Co > 0, "dl_conv3x3_nt: bad args");
  if (Ci % 64 != 0) return DL_ERR_UNSUPPORTED;  // caller materialises cols with dl_im2col3x3 (e.g. the 1-channel stem)
  DL_CHECK_ARG(ldx % 8 == 0 && ldx >= Ci && ldw % 8 == 0 && ldw >= 9 * Ci && ldc % 8 == 0 && ldc >= Co,
               "dl_conv3x3_nt: leading dims");
  DL_CHECK_ARG((((uintptr_t)x | (uintptr_t)Wf | (uintptr_t)out | (uintptr_t)zero | (uintptr_t)resid) & 15) == 0,
               "dl_conv3x3_nt: 16-byte alignment");
  const int64_t M = Bn * H * W, K = 9 * Ci;
  const ConvGeom cg = make_conv_geom(H, W, Ci, ldx, M, zero);
  // persistent 256-row tiles (conv3x3_big_k) where the shape tiles: 256 columns wide when that still yields a full round of work
  // items, the contraction split when the output has few tiles (the low-resolution levels), every split at least 18 k-steps deep
  if (M % TBM == 0 && Co % 128 == 0 && (!bias || ((uintptr_t)bias & 15) == 0) && conv_big_enabled()) {
    static DevOnce once;
    const int n_cu = dev_cus(once, [] {});
    const int64_t mt = M / TBM;
    int tn = (Co % 256 == 0 && mt * (Co / 256) >= n_cu) ? 256 : 128;
    int64_t tiles = mt * (Co / tn);
    if (g_conv_halo && W <= 256 && TBM % W == 0 && ((H * W) % TBM == 0 || TBM % (H * W) == 0)) {
      // one staging of the activation rows per channel chunk (conv3x3_halo_k): launches with a full round of unsplit tiles
      const int nrow = (int)((H * W >= TBM) ? TBM / W : H), nseg = (int)((H * W >= TBM) ? 1 : TBM / (H * W));
      const int nchunks = (nseg * (nrow + 2) * ((int)W + 2) + 7) / 8;
      if (nchunks <= HALO_MAX_CHUNKS && M * ldx < (1ll << 31) && (tiles >= (n_cu * 3) / 4 || g_conv_halo == 2)) {
        NtEpilogue eph{bias, DL_ACT_NONE, 0, nullptr, (const bf16_t*)resid, ldr, nullptr, 0, 1, nullptr, 0, 0};
        int rc = 1;
        if (tn == 256) rc = launch_conv_halo<256, 2>(x, Wf, ldw, out, ldc, M, Co, eph, cg, nrow, nseg, nchunks, (hipStream_t)stream);
        else {
          if (g_conv_halo_nst == 4) rc = launch_conv_halo<128, 4>(x, Wf, ldw, out, ldc, M, Co, eph, cg, nrow, nseg, nchunks, (hipStream_t)stream);
          if (rc == 1 && g_conv_halo_nst >= 3) rc = launch_conv_halo<128, 3>(x, Wf, ldw, out, ldc, M, Co, eph, cg, nrow, nseg, nchunks, (hipStream_t)stream);
          if (rc == 1) rc = launch_conv_halo<128, 2>(x, Wf, ldw, out, ldc, M, Co, eph, cg, nrow, nseg, nchunks, (hipStream_t)stream);
        }
        if (rc < 0) return rc;
        if (rc == 0) return DL_OK;
      }
    }
    int ksplit = 1;
    if (tiles < (n_cu * 3) / 4) {
      auto splits_for = [&](int64_t t) {
        int ks = (int)((n_cu + t - 1) / t);
        if (ks > 8) ks = 8;
        if (ks > (int)(K / 1152)) ks = (int)(K / 1152);
        if (!splitk_scratch || ks > scratch_floats / (M * Co)) ks = splitk_scratch ? (int)(scratch_floats / (M * Co)) : 1;
        return ks < 2 ? 1 : ks;
      };
      // 256-wide tiles (128 FLOP per staged byte against 85) as long as their split still fills three quarters of the chip; the
      // smallest maps (8 x 8 and 4 x 4 at the configured batch of 64: 16-32 wide tiles, at most 8 splits of >= 18 k-steps) do not --
      // there the 128-wide tile's second half of the CUs is worth more than its bytes (round 6: 46-50 -> 35-40 us alone)
      tn = 128;
      tiles = mt * (Co / 128);
      ksplit = splits_for(tiles);
      if (Co % 256 == 0) {
        const int64_t t256 = mt * (Co / 256);
        const int k256 = splits_for(t256);
        // (an UNSPLIT launch of fewer than three quarters of the CUs stays with the 256-wide candidate and, failing the test below,
        //  with the 128 x 128 kernel: 16 x 16, 128 -> 256 channels at B = 64 is 24 us there and 28 on 128 persistent workgroups)
        if (t256 * k256 >= (n_cu * 3) / 4 || t256 * k256 >= tiles * ksplit || ksplit < 2 || !g_conv_split128) tn = 256, tiles = t256, ksplit = k256;
      }
    }
    if (tiles * ksplit >= n_cu / 2 || (g_conv_big == 2 && tiles * ksplit >= 8)) {  // (2 = LAB / tests: whenever the shape tiles)
      NtEpilogue ep{bias, DL_ACT_NONE, 0, nullptr, (const bf16_t*)resid, ldr, nullptr, 0, 1, nullptr, 0, 0};
      ep.part_stride = M * Co;
      const int rc = tn == 256 ? launch_conv_big<256>(x, Wf, ldw, ksplit > 1 ? (void*)splitk_scratch : out, ldc, M, Co, K, ep, ksplit, cg, (hipStream_t)stream)
                               : launch_conv_big<128>(x, Wf, ldw, ksplit > 1 ? (void*)splitk_scratch : out, ldc, M, Co, K, ep, ksplit, cg, (hipStream_t)stream);
      if (rc < 0) return rc;
      if (rc == 0) {
        if (ksplit > 1) {
          int64_t g = (M * (Co / 8) + 255) / 256;
          if (g > 4096) g = 4096;
          hipLaunchKernelGGL(conv_splitk_finalize_k, (int)g, 256, 0, (hipStream_t)stream, splitk_scratch, ksplit, M * Co, bias,
                             (const bf16_t*)resid, ldr, (bf16_t*)out, ldc, M, (int)Co);
          DL_LAUNCH_CHECK();
        }
        return DL_OK;
      }
    }
  }
  const int nwg = cdiv(M, BM) * cdiv(Co, BN);
  // low-resolution levels: few output tiles with a deep contraction (K = 9*Ci up to 18432) -> split K over blockIdx.y; every split
  // stores its partial [M, Co] image into the caller's f32 scratch (up to eight images), a second pass adds them in a fixed order with
  // bias / residual and rounds (no atomics: bit-reproducible).  The 256-thread workgroups run two to a CU: a launch wants ~512.
  int ksplit = 1;
  if (splitk_scratch && nwg <= 320 && K >= 2304 && Co % 8 == 0) {
    ksplit = (512 + nwg - 1) / nwg;
    if (ksplit > 8) ksplit = 8;
    if (ksplit > (int)(K / 1152)) ksplit = (int)(K / 1152);
    if (ksplit > scratch_floats / (M * Co)) ksplit = (int)(scratch_floats / (M * Co));  // never more images than the scratch holds
    if (ksplit < 2) ksplit = 1;
  }
  if (ksplit > 1) {
    NtEpilogue ep0{};
    ep0.rows_per_gate = 1;
    ep0.part_stride = M * Co;
    launch_nt_small<true>(dim3(nwg, ksplit), (hipStream_t)stream, (const bf16_t*)x, ldx, (const bf16_t*)Wf, ldw, splitk_scratch, Co, (int)M,
                          (int)Co, (int)K, ep0, ksplit, cg);
    int64_t g = (M * (Co / 8) + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(conv_splitk_finalize_k, (int)g, 256, 0, (hipStream_t)stream, splitk_scratch, ksplit, M * Co, bias,
                       (const bf16_t*)resid, ldr, (bf16_t*)out, ldc, M, (int)Co);
    DL_LAUNCH_CHECK();
    return DL_OK;
  }
  NtEpilogue ep{bias, DL_ACT_NONE, 0, nullptr, (const bf16_t*)resid, ldr, nullptr, 0, 1, nullptr, 0, 0};
  launch_nt_small<true>(dim3(nwg, 1), (hipStream_t)stream, (const bf16_t*)x, ldx, (const bf16_t*)Wf, ldw, out, ldc, (int)M, (int)Co, (int)K, ep, 1,
                        cg);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
/* weight gradient, transposed: g[(tap, ci), co] += sum_p x[p + shift(tap), ci] * dY[p, co]; dY rows [R, ldy] with R a multiple
 * of 64 and rows >= B*H*W zero; g f32 [9*Ci, ldg] accumulated into (dl_conv3x3_wgrad_fold finishes the job). */
// split plan of the implicit-GEMM weight gradient: which kernel, how many R-splits, steps per split
struct ConvWgradPlan {
  bool big;
  int splits, sps, tiles_m, tiles_n, ntile;
};
static ConvWgradPlan conv_wgrad_plan(int64_t M, int64_t N, int nsteps, int max_workgroups, int n_cu) {
  ConvWgradPlan p{};
#ifndef WGRAD_BIG_MIN_STEPS
#define WGRAD_BIG_MIN_STEPS 64
#endif
  if (M % WBM == 0 && N % WBN == 0 && nsteps >= WGRAD_BIG_MIN_STEPS) {  // same unit / split budget as dl_gemm_tn
    p.big = true;
    p.tiles_m = (int)(M / WBM), p.tiles_n = (int)(N / WBN);
    const int budget = (max_workgroups > 0 && max_workgroups < n_cu) ? max_workgroups : n_cu;
    int padded_max = (budget / p.tiles_n) & ~7;
    if (padded_max < 8) padded_max = 8;
    int splits = padded_max / p.tiles_m;
    if (splits < 1) splits = 1;
    if (splits > nsteps / 8) splits = nsteps / 8;
    p.sps = (nsteps + splits - 1) / splits;
    p.splits = (nsteps + p.sps - 1) / p.sps;
    return p;
  }
  p.ntile = cdiv(M, BM) * cdiv(N, BN);
  int splits = (1024 + p.ntile - 1) / p.ntile;
  if (splits > nsteps) splits = nsteps;
  if (splits < 1) splits = 1;
  p.sps = (nsteps + splits - 1) / splits;
  p.splits = (nsteps + p.sps - 1) / p.sps;
  return p;
}
static int conv_wgrad_cus() {
  static DevOnce once;
  return dev_cus(once, [] {
    (void)hipFuncSetAttribute((const void*)gemm_tn_big_k<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * W_STAGE);
  });
}
int launch_conv_wgrad_halo(const void* x, int64_t ldx, int64_t Bn, int64_t H, int64_t W, int64_t Ci, const void* dY, int64_t ldy, int64_t R,
                           int64_t Co, float* g, int64_t ldg, int64_t part_stride, const void* zero, int max_workgroups, int n_cu,
                           hipStream_t stream);  // conv_wgrad.hip
int conv_wgrad_halo_splits(int64_t H, int64_t W, int64_t Ci, int64_t Co, int64_t R, int max_workgroups, int n_cu, int* sps_out);
static int conv_wgrad_launch(const void* x, int64_t ldx, int64_t Bn, int64_t H, int64_t W, int64_t Ci, const void* dY, int64_t ldy,
                             int64_t R, int64_t Co, float* g, int64_t ldg, int64_t part_stride, int64_t max_parts, const void* zero,
                             int max_workgroups, dl_stream_t stream, const char* who) {
  DL_CHECK_ARG(x && dY && g && zero && Bn > 0 && H > 0 && W > 0 && Co > 0, "%s: bad args", who);
  if (Ci % 128 != 0) return DL_ERR_UNSUPPORTED;
  DL_CHECK_ARG(R % BK == 0 && R >= Bn * H * W && Co % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldy >= Co && ldg >= Co, "%s: dims", who);
  DL_CHECK_ARG((((uintptr_t)x | (uintptr_t)dY | (uintptr_t)zero) & 15) == 0, "%s: 16-byte alignment", who);
  const int64_t M = 9 * Ci, N = Co;
  {  // all nine taps from one staging of the activation rows (conv_wgrad.hip) where the geometry fits
    const int hs = conv_wgrad_halo_splits(H, W, Ci, Co, R, max_workgroups, conv_wgrad_cus(), nullptr);
    if (hs > 0) {
      if (part_stride > 0)
        DL_CHECK_ARG(hs <= max_parts && part_stride >= M * ldg, "%s: %d partial images of %lld floats needed, room for %lld of %lld", who, hs,
                     (long long)(M * ldg), (long long)max_parts, (long long)part_stride);
      const int rc = launch_conv_wgrad_halo(x, ldx, Bn, H, W, Ci, dY, ldy, R, Co, g, ldg, part_stride, zero, max_workgroups,
                                            conv_wgrad_cus(), (hipStream_t)stream);
      if (rc != 1) return rc;
    }
  }
  const ConvGeom cg = make_conv_geom(H, W, Ci, ldx, Bn * H * W, zero);
  const int nsteps = (int)(R / BK);
  const ConvWgradPlan p = conv_wgrad_plan(M, N, nsteps, max_workgroups, conv_wgrad_cus());
  if (part_stride > 0)
    DL_CHECK_ARG(p.splits <= max_parts && part_stride >= M * ldg, "%s: %d partial images of %lld floats needed, room for %lld of %lld", who,
                 p.splits, (long long)(M * ldg), (long long)max_parts, (long long)part_stride);
  if (p.big) {
    const int units = (p.tiles_m * p.splits + 7) & ~7;
    hipLaunchKernelGGL(gemm_tn_big_k<true>, units * p.tiles_n, BIG_THREADS, 2 * W_STAGE, (hipStream_t)stream, (const bf16_t*)x, ldx,
                       (const bf16_t*)dY, ldy, g, ldg, (int)M, (int)N, (int)R, p.sps, cg, part_stride);
  } else {
    hipLaunchKernelGGL(gemm_tn_k<true>, p.ntile * p.splits, NT_THREADS, 65536, (hipStream_t)stream, (const bf16_t*)x, ldx,
                       (const bf16_t*)dY, ldy, g, ldg, (int)M, (int)N, (int)R, p.sps, cg, part_stride);
  }
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_conv3x3_wgrad_tn(const void* x, int64_t ldx, int64_t Bn, int64_t H, int64_t W, int64_t Ci, const void* dY,
                                   int64_t ldy, int64_t R, int64_t Co, float* g, int64_t ldg, const void* zero,
                                   int max_workgroups, dl_stream_t stream) {
  return conv_wgrad_launch(x, ldx, Bn, H, W, Ci, dY, ldy, R, Co, g, ldg, 0, 0, zero, max_workgroups, stream, "dl_conv3x3_wgrad_tn");
}
/* the same product WITHOUT atomics: the R-splits store `n_parts` partial images g + s * part_stride (f32 [9*Ci, ldg] each, written in
 * full with plain stores; nothing is read, nothing needs zeroing) that dl_conv3x3_wgrad_fold_batched adds in a fixed order.
 * dl_conv3x3_wgrad_tn_nparts = the number of images this shape and map produce on this device (0: Ci % 128 != 0, unsupported). */
extern "C" int dl_conv3x3_wgrad_tn_nparts(int64_t H, int64_t W, int64_t Ci, int64_t Co, int64_t R, int max_workgroups) {
  if (Ci % 128 != 0 || R % BK != 0 || Co <= 0) return 0;
  const int hs = conv_wgrad_halo_splits(H, W, Ci, Co, R, max_workgroups, conv_wgrad_cus(), nullptr);
  if (hs > 0) return hs;
  return conv_wgrad_plan(9 * Ci, Co, (int)(R / BK), max_workgroups, conv_wgrad_cus()).splits;
}
extern "C" int dl_conv3x3_wgrad_tn_parts(const void* x, int64_t ldx, int64_t Bn, int64_t H, int64_t W, int64_t Ci, const void* dY,
                                         int64_t ldy, int64_t R, int64_t Co, float* g, int64_t ldg, int64_t part_stride,
                                         int64_t max_parts, const void* zero, int max_workgroups, dl_stream_t stream) {
  DL_CHECK_ARG(part_stride > 0 && max_parts > 0, "dl_conv3x3_wgrad_tn_parts: part_stride / max_parts");
  return conv_wgrad_launch(x, ldx, Bn, H, W, Ci, dY, ldy, R, Co, g, ldg, part_stride, max_parts, zero, max_workgroups, stream,
                           "dl_conv3x3_wgrad_tn_parts");
}

