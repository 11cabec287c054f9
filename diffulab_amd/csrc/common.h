// Shared device/host helpers for libdiffulab_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/diffulab_hip.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits in HBM
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

#define DL_WAVE 64

// ---------------------------------------------------------------- host side error plumbing
void dl_set_error(const char* fmt, ...);
#define DL_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      dl_set_error(__VA_ARGS__);           \
      return DL_ERR_INVALID;               \
    }                                      \
  } while (0)
#define DL_LAUNCH_CHECK()                                                      \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      dl_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return DL_ERR_LAUNCH;                                                    \
    }                                                                          \
  } while (0)

// ---------------------------------------------------------------- bf16 helpers (device)
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float bflo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bfhi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }
// float -> bf16, round-to-nearest-even: the native __bf16 cast lowers to ONE v_cvt_pk_bf16_f32 on gfx950 (a
// hand-rolled integer rounding costs ~10 VALU per pair and made the GEMM epilogues VALU-bound: 7 VALU per MFMA
// in the first rocprof PMC pass)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ void unpack8(const u32x4_t& p, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = bflo(p[i]);
    f[2 * i + 1] = bfhi(p[i]);
  }
}
__device__ __forceinline__ u32x4_t pack8(const float (&f)[8]) {
  u32x4_t p;
#pragma unroll
  for (int i = 0; i < 4; ++i) p[i] = pack2bf(f[2 * i], f[2 * i + 1]);
  return p;
}

// Wavefront reductions without the LDS: `__shfl_xor` lowers to ds_bpermute_b32 on gfx950, i.e. six dependent round trips through
// the LDS pipeline per reduction -- slow on its own and, next to a GEMM workgroup that keeps the CU's LDS port busy (the side-stream
// weight-gradient GEMMs), the reason the LayerNorm kernels ran 2-3x slower inside the training step than alone.  DPP row operations
// stay in the VALU: two quad permutes, row_half_mirror and row_mirror leave every lane with the sum of its 16-lane row; the four
// row sums are read through v_readlane (SGPRs) and combined, so the result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);  // row_half_mirror
  v += dpp_f<0x140>(v);  // row_mirror
  const int i = __float_as_int(v);
  return (__int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48)));
}
// v + (the value 32 lanes away) / max of the two, for every lane: one v_permlane32_swap instead of a ds_bpermute round trip
__device__ __forceinline__ float xor32_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  const int i = __float_as_int(v);
  return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 0)), __int_as_float(__builtin_amdgcn_readlane(i, 16))),
               fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 32)), __int_as_float(__builtin_amdgcn_readlane(i, 48))));
}
// sigmoid through v_rcp_f32 (1 ulp): the IEEE division sequence (v_div_scale / v_div_fmas / v_div_fixup, ~10 VALU instructions)
// was most of the arithmetic of the GroupNorm / SwiGLU kernels and of the fused-SwiGLU GEMM epilogue; the result is rounded to
// bf16 right after
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
// d/dx [x * sigmoid(x)]
__device__ __forceinline__ float dsilu_f(float x) {
  const float s = sigmoid_f(x);
  return s * (1.0f + x * (1.0f - s));
}

// XCD-aware remap of a linear workgroup id: hardware deals block b to XCD b % 8, so give every XCD a
// contiguous chunk of the logical grid (bijective also when nwg % 8 != 0).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
// One-time launch setup per DEVICE (CU count + the dynamic-LDS attribute of kernels that need more than 64 KiB): keyed on the
// device that is current at the call, so a host that drives a second GPU from the same process gets the attribute there too; two
// threads racing the first call both run `setup` (idempotent) and agree on the count.  The only cached state of the library.
#include <atomic>
#define DL_MAX_DEVICES 16
struct DevOnce {
  std::atomic<int> cus[DL_MAX_DEVICES];
};
template <class F>
static inline int dev_cus(DevOnce& st, F&& setup) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const int slot = (dev >= 0 && dev < DL_MAX_DEVICES) ? dev : -1;
  int n = slot >= 0 ? st.cus[slot].load(std::memory_order_acquire) : 0;
  if (n == 0) {
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    if (n <= 0) n = 256;
    setup();
    if (slot >= 0) st.cus[slot].store(n, std::memory_order_release);
  }
  return n;
}
// Workgroup budget of the PERSISTENT main-chain kernels (one workgroup per CU: gemm_nt_big_k, gemm_nt_rows_k, mlp_dswiglu_rc_k) for
// the calls a block driver issues: dl_dit_block_{fwd,bwd} set it from dl_dit_block_t::max_workgroups for their own duration
// (thread-local, restored on return).  A data-parallel host leaves a few CUs to the communication library's workgroups: a grid
// sized for every CU that finds some of them taken runs a SECOND ROUND on the first CUs that free up (+19-22 % per step measured
// with 8-64 foreign workgroups), a grid of CUs - r costs r / CUs.
int dl_wg_budget(int n_cu);
void dl_set_wg_cap(int cap);  // 0 = no cap; returns nothing, callers save / restore through DlWgCapScope
int dl_get_wg_cap();
struct DlWgCapScope {
  int saved;
  explicit DlWgCapScope(int cap) : saved(dl_get_wg_cap()) { if (cap > 0) dl_set_wg_cap(cap); }
  ~DlWgCapScope() { dl_set_wg_cap(saved); }
};
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
