// Microarchitecture probes (tests / tuning only): sustained MFMA rate with and without the LDS fragment traffic and
// the workgroup barrier of the GEMM main loop.  mode 0: MFMA only; 1: + ds_read_b128 fragments (256x192 tile pattern);
// 2: + one s_barrier per k-step; 3: + direct-to-LDS DMA of a 56 KiB stage per k-step from `src` (L2-resident).
//
// LAB CODE, NOT PART OF THE PRODUCT: this file is the whole of libdiffulab_probe.so (include/diffulab_probe.h); nothing in
// libdiffulab_hip.so or under diffulab_amd/*.py loads it.  Users: tests/ (the two instruction-semantics probes that pin the operand
// layouts the product kernels rely on: dl_probe_tr16, dl_probe_mfma_f8) and scripts/*_probe.py (tuning).
#include <stdarg.h>
#include <stdio.h>

#include "../common.h"
#include "../../../include/diffulab_probe.h"

static thread_local char g_probe_err[256];
void dl_set_error(const char* fmt, ...) {  // (the product library has its own; symbols are hidden, the two never meet)
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_probe_err, sizeof(g_probe_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* dl_probe_last_error(void) { return g_probe_err; }

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <int MODE>
__global__ __launch_bounds__(512, 2) void mfma_probe_k(const bf16_t* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  for (int i = threadIdx.x; i < 114688 / 4; i += 512) ((uint32_t*)smem)[i] = 0x3c003c00u + i;  // small finite bf16s
  __syncthreads();
  f32x16_t acc[3][2];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
  bf16x8_t xf[2], wf[3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) xf[i][e] = (__bf16)(0.001f * (lane + e + i));
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) wf[j][e] = (__bf16)(0.002f * (lane - e + j));
  const bf16_t* gsrc = src + (int64_t)blockIdx.x * 57344 / 2;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 3) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE >= 2) __builtin_amdgcn_s_barrier();
    u32x4_t stg[7];
    if (MODE == 4) {  // register staging: global -> VGPR now, VGPR -> LDS after this k-step's MFMAs
#pragma unroll
      for (int c = 0; c < 7; ++c) stg[c] = *(const u32x4_t*)(gsrc + ((wave * 7 + c) * 1024 + lane * 16) / 2);
    }
    if (MODE == 3) {
      char* base = smem + ((it + 1) & 1) * 57344;
#pragma unroll
      for (int c = 0; c < 7; ++c)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(gsrc + ((wave * 7 + c) * 1024 + lane * 16) / 2),
                                         (lds_void_t*)(base + (wave * 7 + c) * 1024), 16, 0, 0);
    }
    const char* sa = smem + (it & 1) * 57344;
    const char* sb = sa + 32768;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (MODE >= 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int r = wm * 64 + i * 32 + (lane & 31);
          xf[i] = *(const bf16x8_t*)(sa + r * 128 + ((((kk << 1) | hi) ^ ((r >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int r = wn * 96 + j * 32 + (lane & 31);
          wf[j] = *(const bf16x8_t*)(sb + r * 128 + ((((kk << 1) | hi) ^ ((r >> 1) & 7)) << 4));
        }
      }
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], xf[i], acc[j][i], 0, 0, 0);
    }
    if (MODE == 4) {
      char* base = smem + ((it + 1) & 1) * 57344;
#pragma unroll
      for (int c = 0; c < 7; ++c) *(u32x4_t*)(base + (wave * 7 + c) * 1024 + lane * 16) = stg[c];
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[j][i][r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void copy_probe_k(const bf16_t* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bf16_t* gsrc = src + (int64_t)blockIdx.x * 57344 / 2;
  u32x4_t acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    char* base = smem + (it & 1) * 57344;
    if (MODE == 5) {
#pragma unroll
      for (int c = 0; c < 7; ++c)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(gsrc + ((wave * 7 + c) * 1024 + lane * 16) / 2),
                                         (lds_void_t*)(base + (wave * 7 + c) * 1024), 16, 0, 0);
      if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    } else {
      u32x4_t stg[7];
#pragma unroll
      for (int c = 0; c < 7; ++c) stg[c] = __builtin_nontemporal_load((const u32x4_t*)(gsrc + ((wave * 7 + c) * 1024 + lane * 16) / 2));
#pragma unroll
      for (int c = 0; c < 7; ++c) {
        if (MODE == 6) *(u32x4_t*)(base + (wave * 7 + c) * 1024 + lane * 16) = stg[c];
        else acc ^= stg[c];
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[blockIdx.x * 512 + threadIdx.x] = (float)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) + ((float*)smem)[threadIdx.x];
}


// store-pattern probe: 256 persistent workgroups x 8 waves write a [65536, 1152] bf16 matrix in 256 x 384 tiles, every wave a
// 64 x 192 sub-tile (the qkv GEMM's output), `iters` times.  mode 8: the register epilogue's pattern -- one instruction = 32 rows
// x 32 contiguous bytes (lane & 31 = row, lane >> 5 = 16-byte half); mode 9: the same bytes with 8 lanes per row -- one
// instruction = 8 rows x 128 contiguous bytes.
template <int MODE>
__global__ __launch_bounds__(512, 2) void store_probe_k(bf16_t* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ld = 1152, tiles_n = 3, ntiles = 256 * 3;
  u32x4_t v = {0x3c003c00u + (unsigned)lane, 0x3c003c01u, 0x3c003c02u, 0x3c003c03u};
  for (int it = 0; it < iters; ++it)
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int m0 = (tile / tiles_n) * 256 + wm * 64, n0 = (tile % tiles_n) * 384 + wn * 192;
      if (MODE == 8) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp)
              *(u32x4_t*)(out + (int64_t)(m0 + i * 32 + (lane & 31)) * ld + n0 + j * 32 + 16 * gp + 8 * hi) = v;
      } else {
        // 192 columns = 384 B = 3 lines of 128 B per row: 8 lanes per line, 8 rows per instruction, 24 instructions per 64 rows
#pragma unroll
        for (int rg = 0; rg < 8; ++rg)
#pragma unroll
          for (int ln = 0; ln < 3; ++ln)
            *(u32x4_t*)(out + (int64_t)(m0 + rg * 8 + (lane >> 3)) * ld + n0 + ln * 64 + (lane & 7) * 8) = v;
      }
      v[1] += 1;
    }
}

// DMA-pattern probe for the NT GEMM operand stream: 256 persistent workgroups x 8 waves walk the (256-row tile, k-step) space of
// A[M, K] . B[384, K]^T exactly like gemm_nt_big_k<384,2,*> but only issue the direct-to-LDS loads (no LDS reads, no MFMAs):
//   KB = 128: 64-deep k-steps, 128-byte row segments, 80 KiB stages, 2 ring slots (1 stage in flight);
//   KB = 64 : 32-deep k-steps,  64-byte row segments (half a cache line per row and step), 40 KiB stages, 4 slots (3 in flight).
template <int KB>
__global__ __launch_bounds__(512, 2) void dma_probe_k(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Bw, int M, int K,
                                                        float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWS = 640, STAGE = ROWS * KB, NS = 163840 / STAGE, CH = STAGE / 1024 / 8, RPC = 1024 / KB, LPR = KB / 16;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nk = K * 2 / KB, ntiles = M / 256;
  int it = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (int kt = 0; kt < nk; ++kt, ++it) {
      char* base = smem + (it % NS) * STAGE;
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int c = wave * CH + i;
        const int row = c * RPC + lane / LPR;  // 0..639: rows < 256 belong to the A tile, the rest to the 384 weight rows
        const bf16_t* src = row < 256 ? A + (int64_t)(tile * 256 + row) * K : Bw + (int64_t)(row - 256) * K;
        __builtin_amdgcn_global_load_lds((glb_void_t*)(src + kt * (KB / 2) + (lane % LPR) * 8), (lds_void_t*)(base + c * 1024), 16, 0, 0);
      }
      if (KB == 128) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * CH) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * CH) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[blockIdx.x * 512 + threadIdx.x] = ((float*)smem)[threadIdx.x];
}
extern "C" int dl_probe_dma(int kb, const void* A, const void* Bw, int64_t M, int64_t K, float* out, dl_stream_t stream) {
  DL_CHECK_ARG(A && Bw && out && M % 256 == 0 && K % 64 == 0 && (kb == 64 || kb == 128), "dl_probe_dma: bad args");
  (void)hipFuncSetAttribute((const void*)dma_probe_k<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  (void)hipFuncSetAttribute((const void*)dma_probe_k<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  if (kb == 128) hipLaunchKernelGGL(dma_probe_k<128>, 256, 512, 163840, (hipStream_t)stream, (const bf16_t*)A, (const bf16_t*)Bw, (int)M, (int)K, out);
  else hipLaunchKernelGGL(dma_probe_k<64>, 256, 512, 163840, (hipStream_t)stream, (const bf16_t*)A, (const bf16_t*)Bw, (int)M, (int)K, out);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ---- round 4: does the operand stream scale with the number of RESIDENT waves / workgroups per CU?  A workgroup of `blockDim` threads
// walks row panels of `rows_a` activation rows (tile t: rows [t rows_a, ...)) and stages, per 64-deep k-step, rows_a + rows_b rows of
// 128 bytes (the weight rows are the same rows_b rows for every tile: L2-resident) into an `nslot`-deep ring; nothing is computed.
// grid = 256 * wgs_per_cu.  What the asymmetric-issue experiment suggested: the stream's rate follows the number of issuing waves.
__global__ void dma_probe2_k(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Bw, int M, int K, int rows_a, int rows_b,
                             int nslot, float* __restrict__ out, int rot) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
  const int rows = rows_a + rows_b, stage = rows * 128, nch = rows / 8;
  const int nk = K / 64, ntiles = M / rows_a;
  int it = 0;
  // rot > 0: workgroup w of an XCD walks the k-steps from k-step (rot * w) % nk on -- the 32 workgroups of an XCD then read
  // DIFFERENT lines of the shared weight panel at any moment instead of the same ones (L2 channel hot spot?)
  const int k0 = (rot * (int)(blockIdx.x >> 3)) % nk;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (int kq = 0; kq < nk; ++kq, ++it) {
      const int kt = (kq + k0) % nk;
      char* base = smem + (it % nslot) * stage;
      for (int c = wave; c < nch; c += nwaves) {
        const int row = c * 8 + (lane >> 3);
        const bf16_t* src = row < rows_a ? A + (int64_t)(tile * rows_a + row) * K : Bw + (int64_t)(row - rows_a) * K;
        __builtin_amdgcn_global_load_lds((glb_void_t*)(src + kt * 64 + (lane & 7) * 8), (lds_void_t*)(base + c * 1024), 16, 0, 0);
      }
      if (nslot == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      } else if ((it % nslot) == nslot - 1) {  // (every nslot stages: let all but the youngest stage land)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[(blockIdx.x * blockDim.x + threadIdx.x) & (256 * 512 - 1)] = ((float*)smem)[threadIdx.x];
}
static int probe_dma2(int wgs_per_cu, int threads, int rows_a, int rows_b, int nslot, int rot, const void* A, const void* Bw, int64_t M,
                      int64_t K, float* out, dl_stream_t stream) {
  DL_CHECK_ARG(A && Bw && out && wgs_per_cu >= 1 && threads % 64 == 0 && threads <= 1024 && rows_a % 8 == 0 && rows_b % 8 == 0 &&
               nslot >= 1 && M % rows_a == 0 && K % 64 == 0, "dl_probe_dma2: bad args");
  const int lds = (rows_a + rows_b) * 128 * nslot;
  DL_CHECK_ARG(lds <= 163840 / wgs_per_cu, "dl_probe_dma2: %d bytes of LDS per workgroup do not fit %d times", lds, wgs_per_cu);
  (void)hipFuncSetAttribute((const void*)dma_probe2_k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipLaunchKernelGGL(dma_probe2_k, 256 * wgs_per_cu, threads, lds, (hipStream_t)stream, (const bf16_t*)A, (const bf16_t*)Bw, (int)M, (int)K,
                     rows_a, rows_b, nslot, out, rot);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_probe_dma2(int wgs_per_cu, int threads, int rows_a, int rows_b, int nslot, const void* A, const void* Bw, int64_t M,
                             int64_t K, float* out, dl_stream_t stream) {
  return probe_dma2(wgs_per_cu, threads, rows_a, rows_b, nslot, 0, A, Bw, M, K, out, stream);
}
extern "C" int dl_probe_dma2_rot(int wgs_per_cu, int threads, int rows_a, int rows_b, int nslot, int rot, const void* A, const void* Bw,
                                 int64_t M, int64_t K, float* out, dl_stream_t stream) {
  DL_CHECK_ARG(rot >= 0, "dl_probe_dma2_rot: rot < 0");
  return probe_dma2(wgs_per_cu, threads, rows_a, rows_b, nslot, rot, A, Bw, M, K, out, stream);
}

// The same 256 + 384-row tile walk with PLAIN loads into registers (nothing goes to the LDS): is the L2 -> CU path itself faster than
// the direct-to-LDS form?  PAT 0: the DMA's address pattern (8 rows x 128 B per wave instruction); PAT 1: the MFMA fragment pattern
// (32 rows x 32 B per wave instruction: lane l reads row l & 31, 16 bytes at 16 (2 kk + (l >> 5))), i.e. what a kernel that feeds
// one operand straight from L1 / L2 would issue.  DEPTH stages' loads are issued before the oldest stage's registers are consumed.
template <int PAT, int DEPTH>
__global__ __launch_bounds__(512, 2) void ld_probe_k(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Bw, int M, int K,
                                                     float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nk = K / 64, ntiles = M / 256;
  u32x4_t acc = {0u, 0u, 0u, 0u};
  u32x4_t ring[DEPTH][10];
  int it = 0;
  const int total = ((ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * nk;
  auto issue = [&](int s, u32x4_t (&dst)[10]) {
    s = s < total ? s : s - total;  // (the run-ahead of the last stages wraps: every load stays in bounds, no branch in the loop)
    const int tile = blockIdx.x + (s / nk) * gridDim.x, kt = s % nk;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int c = wave * 10 + i;  // 80 chunks of 1 KiB: 32 of the activation tile, 48 of the weight panel
      int row, piece;
      if (PAT == 0) {
        row = c * 8 + (lane >> 3);
        piece = lane & 7;
      } else {
        row = (c >> 2) * 32 + (lane & 31);
        piece = 2 * (c & 3) + (lane >> 5);
      }
      const bf16_t* src = row < 256 ? A + (int64_t)(tile * 256 + row) * K : Bw + (int64_t)(row - 256) * K;
      dst[i] = *(const u32x4_t*)(src + kt * 64 + piece * 8);
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) issue(d, ring[d]);
  for (it = 0; it < total; it += DEPTH) {  // total % DEPTH == 0 (checked by the host)
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      issue(it + d + DEPTH - 1, ring[(d + DEPTH - 1) % DEPTH]);
      __builtin_amdgcn_sched_barrier(0);  // (the scheduler would otherwise sink the loads to their uses)
#pragma unroll
      for (int i = 0; i < 10; ++i) acc ^= ring[d][i];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  out[(blockIdx.x * blockDim.x + threadIdx.x) & (256 * 512 - 1)] = __uint_as_float(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
}
extern "C" int dl_probe_ld(int pattern, int depth, const void* A, const void* Bw, int64_t M, int64_t K, float* out, dl_stream_t stream) {
  DL_CHECK_ARG(A && Bw && out && M % 256 == 0 && K % 64 == 0 && (pattern == 0 || pattern == 1) && (depth == 2 || depth == 3) &&
               ((M / 256 + 255) / 256 * (K / 64)) % depth == 0 && (M / 256) % 256 == 0, "dl_probe_ld: bad args");
#define GO(P, D) hipLaunchKernelGGL((ld_probe_k<P, D>), 256, 512, 0, (hipStream_t)stream, (const bf16_t*)A, (const bf16_t*)Bw, (int)M, (int)K, out)
  if (pattern == 0 && depth == 2) GO(0, 2);
  else if (pattern == 0) GO(0, 3);
  else if (depth == 2) GO(1, 2);
  else GO(1, 3);
#undef GO
  DL_LAUNCH_CHECK();
  return DL_OK;
}

extern "C" int dl_probe_mfma(int mode, int iters, const void* src, float* out, dl_stream_t stream) {
  DL_CHECK_ARG(out && iters > 0 && mode >= 0 && mode <= 9 && (mode < 3 || mode > 7 || src), "dl_probe_mfma: bad args");
  if (mode >= 8) {  // store-pattern probe: out is a bf16 [65536, 1152] buffer
    if (mode == 8) hipLaunchKernelGGL(store_probe_k<8>, 256, 512, 0, (hipStream_t)stream, (bf16_t*)out, iters);
    else hipLaunchKernelGGL(store_probe_k<9>, 256, 512, 0, (hipStream_t)stream, (bf16_t*)out, iters);
    DL_LAUNCH_CHECK();
    return DL_OK;
  }
  const int lds = 114688;
#define GO(MODE)                                                                                               \
  do {                                                                                                         \
    (void)hipFuncSetAttribute((const void*)mfma_probe_k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    hipLaunchKernelGGL(mfma_probe_k<MODE>, 256, 512, lds, (hipStream_t)stream, (const bf16_t*)src, out, iters);   \
  } while (0)
  if (mode == 0) GO(0);
  else if (mode == 1) GO(1);
  else if (mode == 2) GO(2);
  else if (mode == 3) GO(3);
  else if (mode == 4) GO(4);
  else {
#define GOC(MODE)                                                                                              \
  do {                                                                                                         \
    (void)hipFuncSetAttribute((const void*)copy_probe_k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    hipLaunchKernelGGL(copy_probe_k<MODE>, 256, 512, lds, (hipStream_t)stream, (const bf16_t*)src, out, iters);   \
  } while (0)
    if (mode == 5) GOC(5);
    else if (mode == 6) GOC(6);
    else GOC(7);
#undef GOC
  }
#undef GO
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// =====================================================================================================
// raw lane map of ds_read_b64_tr_b16 (pins the semantics the weight-gradient GEMMs and the attention kernels rely on)
// =====================================================================================================
__global__ void probe_tr16_k(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t img[256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) img[i] = (uint16_t)i;
  __syncthreads();
  s16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"((unsigned)(uintptr_t)(lds_void_t*)((const char*)img + lane * 8)));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));
#pragma unroll
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}
extern "C" int dl_probe_tr16(uint16_t* out, dl_stream_t stream) {
  DL_CHECK_ARG(out, "dl_probe_tr16: null");
  hipLaunchKernelGGL(probe_tr16_k, 1, 64, 0, (hipStream_t)stream, out);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// =====================================================================================================
// D[32,32] = A[32,64] . B[32,64]^T through ONE v_mfma_scale_f32_32x32x64_f8f6f4 with the operand layout attention_fp8.hip relies
// on: lane l holds row (l & 31), bytes [32 (l >> 5), +32) of the row; D in the 32x32 accumulator layout (row (r&3) + 8 (r>>2) +
// 4 (l>>5), column l & 31).  cbsz = 0 / blgp = 0: both operands OCP e4m3; scale exponents 127 = 2^0 in every byte.
// =====================================================================================================
typedef int v8i_t __attribute__((ext_vector_type(8)));
__global__ void probe_mfma_f8_k(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, float* __restrict__ d) {
  const int lane = threadIdx.x, hi = lane >> 5;
  const v8i_t av = *(const v8i_t*)(a + (lane & 31) * 64 + hi * 32);
  const v8i_t bv = *(const v8i_t*)(b + (lane & 31) * 64 + hi * 32);
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#pragma unroll
  for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + (lane & 31)] = acc[r];
}
extern "C" int dl_probe_mfma_f8(const void* a, const void* b, float* d, dl_stream_t stream) {
  DL_CHECK_ARG(a && b && d, "dl_probe_mfma_f8: null");
  hipLaunchKernelGGL(probe_mfma_f8_k, 1, 64, 0, (hipStream_t)stream, (const uint8_t*)a, (const uint8_t*)b, d);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ---------------------------------------------------------------------------------------------- CU holder
__global__ void probe_spin_k(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
extern "C" int dl_probe_spin(int n_wgs, int threads, int usec, dl_stream_t stream) {
  if (n_wgs <= 0 || threads <= 0 || threads > 1024 || usec <= 0) return -1;
  hipLaunchKernelGGL(probe_spin_k, n_wgs, threads, 0, (hipStream_t)stream, (long long)usec * 100);  // wall_clock64: 100 MHz
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
