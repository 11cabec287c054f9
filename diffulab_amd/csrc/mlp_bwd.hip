// PackedSwiGLU MLP backward without saved pre-activations (gfx950).
//
// The MLP-up forward (dl_gemm_nt_swiglu) has to write u = [x1 | x3] (2F wide: 403 MB per block at the headline shape, twice the
// size of its useful output h) only because the backward needs silu'(x1), x3 and silu(x1).  Here the backward RECOMPUTES the u
// tile it needs and never reads it from HBM:
//   per 256-row x 128-hidden-unit tile, one persistent workgroup (8 waves, 4 (m) x 2 (n)) runs TWO k-loops back to back
//     phase 1: u_tile[256, 256] = xm2_tile . Wp_tile^T   (Wp = the row-permuted weight shadow of the fused forward: every 32-row MFMA
//              tile holds x1 of 16 units then x3 of the same 16 units, so a lane ends up with x1 AND x3 of the same 8 units)
//              -> rounded to bf16 exactly like the forward's stored u and kept in 64 registers per lane
//     phase 2: dh_tile[256, 128] = dT_tile . W2t_tile^T  (the MLP-down data gradient, never written either)
//     epilogue: du = [dh * x3 * silu'(x1) | dh * silu(x1)], one 16-byte store per lane and 8 units, as in the fused forward
//   operands stream through a two-slot 64 KiB direct-to-LDS ring like gemm_nt_big_k's (phase-2 stages use 48 KiB of a slot).
// Per block and step this removes the u store of the forward (8 u), the dh round trip (8 u) and the u read of the SwiGLU backward
// (8 u) -- 24 of the 126 u of block traffic (u = 50 MB at B = 256) -- for 154.6 GFLOP of extra MFMA work.
// Requires M % 256 == 0, F % 128 == 0, K1 % 64 == 0, K2 % 64 == 0.
#include <stdlib.h>

#include "common.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__device__ __forceinline__ void rc_glds16(const void* gsrc, void* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void_t*)gsrc, (lds_void_t*)lds_dst_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void rc_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// one 64-deep k-step of a wave: acc[j][i] += W rows (32 x JN) . X rows (32 x 2)^T over the four 16-deep sub-steps (the fragment
// software pipeline of gemm_nt_big_k was measured here as well: 316 vs 317 us, not kept -- the kernel waits for its operand stream)
template <int JN>
__device__ __forceinline__ void rc_kstep(const char* sa, const char* sb, const int (&xrow)[2], const int (&wrow)[JN], int hi,
                                         f32x16_t (&acc)[JN][2]) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    bf16x8_t xf[2], wf[JN];
#pragma unroll
    for (int i = 0; i < 2; ++i) xf[i] = *(const bf16x8_t*)(sa + xrow[i] * 128 + ((((kk << 1) | hi) ^ ((xrow[i] >> 1) & 7)) << 4));
#pragma unroll
    for (int j = 0; j < JN; ++j) wf[j] = *(const bf16x8_t*)(sb + wrow[j] * 128 + ((((kk << 1) | hi) ^ ((wrow[j] >> 1) & 7)) << 4));
#pragma unroll
    for (int j = 0; j < JN; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], xf[i], acc[j][i], 0, 0, 0);
  }
}

#define RC_TBM 256
#define RC_THREADS 512
#ifndef RC_TU_DEFAULT
#define RC_TU_DEFAULT 128
#endif

// TU = hidden units per tile (2*TU columns of u): 128 -> 128 + 64 + 64 accumulator / saved-u registers per lane; 192 would need
// 192 + 96 + 96 and spills (157 registers) -- the bytes staged per unit are the same for both
template <int RC_TU>
__global__ __launch_bounds__(RC_THREADS, 2) void mlp_dswiglu_rc_k(const bf16_t* __restrict__ X, int64_t ldx,
                                                                    const bf16_t* __restrict__ Wp, int64_t ldwp,
                                                                    const bf16_t* __restrict__ dT, int64_t ldt,
                                                                    const bf16_t* __restrict__ W2t, int64_t ldw2,
                                                                    bf16_t* __restrict__ dU, int64_t lddu, int M, int F, int K1,
                                                                    int K2, int csplit) {
  constexpr int RC_STAGE = (RC_TBM + 2 * RC_TU) * 128;  // ring slot: phase 1 stages 256 + 2*TU rows of 64 k, phase 2 256 + TU
  constexpr int CH1 = (RC_TBM + 2 * RC_TU) / 64;  // 1 KiB DMA chunks per wave and phase-1 stage
  constexpr int CH2 = (RC_TBM + RC_TU) / 64;      // ... phase-2 stage
  constexpr int JN1 = RC_TU / 32, JN2 = RC_TU / 64;  // MFMA column tiles per wave in the two phases
  constexpr int ESTORES = JN2 * 8;                // 16-byte stores per wave in the epilogue (JN2 x 2 x 2 x 2)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = F / RC_TU, ntiles = (M / RC_TBM) * tiles_n;
  const int nk1 = K1 / 64, nk2 = K2 / 64, nks = nk1 + nk2;
  // Tile ownership by XCD (block b runs on XCD b % 8, each XCD has its own 4 MiB L2): the column tiles are cut into `csplit`
  // groups and the row panels into 8 / csplit groups; XCD x owns column group x % csplit of panel group x / csplit and walks it
  // panel-major, so at any time its 32 workgroups read a handful of row panels together (each panel is fetched once and hit
  // nc - 1 times) and re-read only 1 / csplit of the weights -- with csplit = 1 the 3.5 MB of weights (DiT-S: L2-sized) were evicted
  // by the streaming panels and the du stores and re-fetched for every slot (PMC: 477 MB fetched per launch for 128 MB of operands)
  const int G = gridDim.x;  // multiple of 8
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, GL = G >> 3;
  const int panels = M / RC_TBM, pgroups = 8 / csplit;
  const int nc = tiles_n / csplit, np_max = (panels + pgroups - 1) / pgroups;  // (the last panel group may be short or empty)
  const int tm0 = (xcd / csplit) * np_max, tn0 = (xcd % csplit) * nc;
  const int np = panels - tm0 < np_max ? (panels - tm0 > 0 ? panels - tm0 : 0) : np_max;
  const int cnt = np * nc > local ? (np * nc - local + GL - 1) / GL : 0;
  const int total = cnt * nks;

  // ---- per-lane constant parts of the DMA chunks (element offsets inside the operand tile; the swizzle sits in the source address)
  int off1[CH1], off2[CH2];
#pragma unroll
  for (int i = 0; i < CH1; ++i) {
    const int c = wave * CH1 + i;
    const bool a = c < RC_TBM / 8;
    const int cc = a ? c : c - RC_TBM / 8;
    const int r = cc * 8 + (lane >> 3);
    off1[i] = r * (int)(a ? ldx : ldwp) + (((lane & 7) ^ ((r >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int i = 0; i < CH2; ++i) {
    const int c = wave * CH2 + i;
    const bool a = c < RC_TBM / 8;
    const int cc = a ? c : c - RC_TBM / 8;
    const int r = cc * 8 + (lane >> 3);
    off2[i] = r * (int)(a ? ldt : ldw2) + (((lane & 7) ^ ((r >> 1) & 7)) << 3);
  }
  // DMA cursor: one stage ahead of the compute cursor
  int c_tile = local, c_s = 0, c_g = 0;  // (XCD-local tile index q: panel tm0 + q / nc, column tile tn0 + q % nc)
  auto issue = [&]() __attribute__((always_inline)) {
    char* base = smem + (c_g & 1) * RC_STAGE;
    const int pl = c_tile / nc;
    const int tm = tm0 + pl, tn = tn0 + (c_tile - pl * nc);
    if (c_s < nk1) {
      const bf16_t* pa = X + (int64_t)tm * RC_TBM * ldx + c_s * 64;
      const bf16_t* pb = Wp + (int64_t)tn * (2 * RC_TU) * ldwp + c_s * 64;
#pragma unroll
      for (int i = 0; i < CH1; ++i) {
        const int c = wave * CH1 + i;
        const bool a = c < RC_TBM / 8;
        rc_glds16((a ? pa : pb) + off1[i], base + (a ? c * 1024 : RC_TBM * 128 + (c - RC_TBM / 8) * 1024));
      }
    } else {
      const int k = c_s - nk1;
      const bf16_t* pa = dT + (int64_t)tm * RC_TBM * ldt + k * 64;
      const bf16_t* pb = W2t + (int64_t)tn * RC_TU * ldw2 + k * 64;
#pragma unroll
      for (int i = 0; i < CH2; ++i) {
        const int c = wave * CH2 + i;
        const bool a = c < RC_TBM / 8;
        rc_glds16((a ? pa : pb) + off2[i], base + (a ? c * 1024 : RC_TBM * 128 + (c - RC_TBM / 8) * 1024));
      }
    }
    ++c_g;
    if (++c_s == nks) {
      c_s = 0;
      c_tile += GL;
    }
  };

  int xrow[2], wrow1[JN1], wrow2[JN2];
#pragma unroll
  for (int i = 0; i < 2; ++i) xrow[i] = wm * 64 + i * 32 + (lane & 31);
#pragma unroll
  for (int j = 0; j < JN1; ++j) wrow1[j] = wn * RC_TU + j * 32 + (lane & 31);
#pragma unroll
  for (int j = 0; j < JN2; ++j) wrow2[j] = wn * (RC_TU / 2) + j * 32 + (lane & 31);

  if (total > 0) issue();
  int g = 0;
  bool after_epi = false;
  for (int t = 0; t < cnt; ++t) {
    const int tile = local + t * GL;
    const int tm = tm0 + tile / nc, tn = tn0 + tile % nc;
    // ------------------------------------------------------------------ phase 1: u tile (x1 | x3 of TU hidden units)
    f32x16_t acc1[JN1][2];
#pragma unroll
    for (int j = 0; j < JN1; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[j][i][r] = 0.f;
    for (int s = 0; s < nk1; ++s, ++g) {
      if (after_epi) rc_wait_vmcnt<ESTORES>();  // (the previous tile's stores are younger than this stage and may keep draining)
      else rc_wait_vmcnt<0>();
      after_epi = false;
      __builtin_amdgcn_s_barrier();
      const char* sa = smem + (g & 1) * RC_STAGE;
      const char* sb = sa + RC_TBM * 128;
      if (c_g < total) issue();
      rc_kstep<JN1>(sa, sb, xrow, wrow1, hi, acc1);
    }
    // acc1[j][i][0..7] = x1, [8..15] = x3 of the same 8 units (row = lane & 31 of row block i): round to bf16 like the forward's
    // stored u; pk[j][i][e] = (x1[2e], x1[2e+1]), pk[j][i][4+e] = (x3[2e], x3[2e+1])
    uint32_t pk[JN1][2][8];
#pragma unroll
    for (int j = 0; j < JN1; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pk[j][i][e] = pack2bf(acc1[j][i][2 * e], acc1[j][i][2 * e + 1]);
          pk[j][i][4 + e] = pack2bf(acc1[j][i][8 + 2 * e], acc1[j][i][8 + 2 * e + 1]);
        }
    // ------------------------------------------------------------------ phase 2: dh tile
    f32x16_t acc2[JN2][2];
#pragma unroll
    for (int j = 0; j < JN2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[j][i][r] = 0.f;
    for (int s = 0; s < nk2; ++s, ++g) {
      rc_wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      const char* sa = smem + (g & 1) * RC_STAGE;
      const char* sb = sa + RC_TBM * 128;
      if (c_g < total) issue();
      rc_kstep<JN2>(sa, sb, xrow, wrow2, hi, acc2);
    }
    // ------------------------------------------------------------------ epilogue: du1 = dh * x3 * silu'(x1), du3 = dh * silu(x1)
    // phase-2 register 8*gp + e of unit tile j2 <-> phase-1 tile 2*j2 + gp, register e; one v_permlane32_swap per register pair
    // gives every lane 8 consecutive units (the packed x1 / x3 pairs go through the same exchange)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t m = (int64_t)tm * RC_TBM + wm * 64 + i * 32 + (lane & 31);
#pragma unroll
      for (int j2 = 0; j2 < JN2; ++j2)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int j1 = 2 * j2 + gp;
          float v[8], a[8], b[8], d1[8], d3[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc2[j2][i][8 * gp + e]), __float_as_uint(acc2[j2][i][8 * gp + 4 + e]),
                                                       false, false);
            v[e] = __uint_as_float(sw[0]);
            v[4 + e] = __uint_as_float(sw[1]);
          }
          auto a0 = __builtin_amdgcn_permlane32_swap(pk[j1][i][0], pk[j1][i][2], false, false);
          auto a1 = __builtin_amdgcn_permlane32_swap(pk[j1][i][1], pk[j1][i][3], false, false);
          auto b0 = __builtin_amdgcn_permlane32_swap(pk[j1][i][4], pk[j1][i][6], false, false);
          auto b1 = __builtin_amdgcn_permlane32_swap(pk[j1][i][5], pk[j1][i][7], false, false);
          const u32x4_t ap = {a0[0], a1[0], a0[1], a1[1]}, bp = {b0[0], b1[0], b0[1], b1[1]};
          unpack8(ap, a);
          unpack8(bp, b);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            d1[e] = v[e] * b[e] * dsilu_f(a[e]);
            d3[e] = v[e] * silu_f(a[e]);
          }
          const int c = tn * RC_TU + wn * (RC_TU / 2) + j2 * 32 + 16 * gp + 8 * hi;
          *(u32x4_t*)(dU + m * lddu + c) = pack8(d1);
          *(u32x4_t*)(dU + m * lddu + F + c) = pack8(d3);
        }
    }
    after_epi = true;
  }
}

/* du = [dh * x3 * silu'(x1) | dh * silu(x1)] with u = [x1 | x3] = X Wp^T recomputed per tile and dh = dT W2t^T, neither of them
 * written (see the file header).  X [M, K1], Wp = the row-permuted [2F, K1] shadow of dl_cast_weight_swiglu, dT [M, K2],
 * W2t [F, K2] = transposed shadow of the MLP-down weight, dU [M, 2F]. */
extern "C" int dl_mlp_dswiglu_recompute(const void* X, int64_t ldx, const void* Wp, int64_t ldwp, const void* dT, int64_t ldt,
                                        const void* W2t, int64_t ldw2, void* dU, int64_t lddu, int64_t M, int64_t F, int64_t K1,
                                        int64_t K2, dl_stream_t stream) {
  DL_CHECK_ARG(X && Wp && dT && W2t && dU && M > 0 && F > 0 && K1 > 0 && K2 > 0, "dl_mlp_dswiglu_recompute: null/empty operand");
  DL_CHECK_ARG(ldx % 8 == 0 && ldwp % 8 == 0 && ldt % 8 == 0 && ldw2 % 8 == 0 && lddu % 8 == 0 && ldx >= K1 && ldwp >= K1 &&
                   ldt >= K2 && ldw2 >= K2 && lddu >= 2 * F,
               "dl_mlp_dswiglu_recompute: leading dims");
  DL_CHECK_ARG((((uintptr_t)X | (uintptr_t)Wp | (uintptr_t)dT | (uintptr_t)W2t | (uintptr_t)dU) & 15) == 0,
               "dl_mlp_dswiglu_recompute: 16-byte alignment");
  constexpr int TU = RC_TU_DEFAULT;
  if (M % RC_TBM || F % TU || K1 % 64 || K2 % 64 || (M / RC_TBM) * (F / TU) < 64) {
    dl_set_error("dl_mlp_dswiglu_recompute: no kernel for M=%lld F=%lld K1=%lld K2=%lld", (long long)M, (long long)F, (long long)K1,
                 (long long)K2);
    return DL_ERR_UNSUPPORTED;
  }
  static DevOnce once;
  const int n_cu = dev_cus(once, [] {
    (void)hipFuncSetAttribute((const void*)mlp_dswiglu_rc_k<TU>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (RC_TBM + 2 * TU) * 128);
  });
  const int ntiles = (int)((M / RC_TBM) * (F / TU));
  const int budget = dl_wg_budget(n_cu);
  int grid = budget < ntiles ? budget : ntiles;
  grid &= ~7;
  // column groups per launch (see the kernel): the widest split that keeps whole column tiles and whole row panels per XCD
  const int panels = (int)(M / RC_TBM), tiles_n = (int)(F / TU);
  // column groups per XCD (see the kernel).  Measured at the headline shape (M = 65536, F = 1536; PMC FETCH_SIZE per launch against
  // 128 MB of operands): csplit 1 / 2 / 4 = 610 / 389 / 526 MB fetched, 321 / 326 / 336 us alone, 21.65 / 21.58 / 21.73 ms per step.
  // Two groups fetch the least; the launch time does not follow the fetch (the kernel waits for the ISSUE of its operand stream,
  // not for L2 misses), so this only takes traffic off the fabric that the side-stream weight gradients share.
  int csplit = 1;
  for (int c = 2; c >= 1; c >>= 1)
    if (tiles_n % c == 0) {
      csplit = c;
      break;
    }
  (void)panels;
  hipLaunchKernelGGL(mlp_dswiglu_rc_k<TU>, grid, RC_THREADS, 2 * (RC_TBM + 2 * TU) * 128, (hipStream_t)stream, (const bf16_t*)X, ldx,
                     (const bf16_t*)Wp, ldwp, (const bf16_t*)dT, ldt, (const bf16_t*)W2t, ldw2, (bf16_t*)dU, lddu, (int)M, (int)F, (int)K1,
                     (int)K2, csplit);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
