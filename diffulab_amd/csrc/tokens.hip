// Token routing of the SPRINT denoiser (reference networks/denoisers/sprint.py:317-387): gather of the kept tokens, scatter
// back into a mask-token canvas, and the gated residual materialised at a stage boundary.  HBM-bound row copies: one
// thread per (row, 8-channel chunk), 16-byte accesses.
#include "common.h"

// dst[b, j, :] = keep[b] ? src[b, idx[b, j], :] : 0     src rows [B*N] (ld_src), dst rows [B*k] (ld_dst)
__global__ void gather_tokens_k(const bf16_t* __restrict__ src, int64_t ld_src, const int* __restrict__ idx,
                                const int* __restrict__ keep, bf16_t* __restrict__ dst, int64_t ld_dst, int N, int k, int D8,
                                int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % D8);
    const int64_t row = i / D8;  // b * k + j
    const int64_t b = row / k;
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (!keep || keep[b]) v = *(const u32x4_t*)(src + (b * N + idx[row]) * ld_src + c * 8);
    *(u32x4_t*)(dst + row * ld_dst + c * 8) = v;
  }
}
// dst[b, idx[b, j], :] += src[b, j, :]   (adjoint of the gather; the indices of one sample are distinct)
__global__ void scatter_tokens_add_k(const bf16_t* __restrict__ src, int64_t ld_src, const int* __restrict__ idx,
                                     bf16_t* __restrict__ dst, int64_t ld_dst, int N, int k, int D8, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % D8);
    const int64_t row = i / D8;
    const int64_t b = row / k;
    float a[8], o[8];
    unpack8(*(const u32x4_t*)(src + row * ld_src + c * 8), a);
    bf16_t* p = dst + (b * N + idx[row]) * ld_dst + c * 8;
    unpack8(*(const u32x4_t*)p, o);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] += a[e];
    *(u32x4_t*)p = pack8(o);
  }
}
// out[b, n, :] = inv[b, n] >= 0 ? xd[b, inv[b, n], :] : bf16(mask[:])
__global__ void restore_tokens_k(const bf16_t* __restrict__ xd, int64_t ld_xd, const int* __restrict__ inv,
                                 const float* __restrict__ mask, bf16_t* __restrict__ out, int64_t ld_out, int N, int k, int D8,
                                 int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % D8);
    const int64_t row = i / D8;  // b * N + n
    const int64_t b = row / N;
    const int j = inv[row];
    u32x4_t v;
    if (j >= 0) {
      v = *(const u32x4_t*)(xd + (b * k + j) * ld_xd + c * 8);
    } else {
      float m[8];
      const f32x4_t m0 = *(const f32x4_t*)(mask + c * 8), m1 = *(const f32x4_t*)(mask + c * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = m0[e], m[4 + e] = m1[e];
      v = pack8(m);
    }
    *(u32x4_t*)(out + row * ld_out + c * 8) = v;
  }
}
// out[c] += sum over rows with sel[row] < 0 of x[row, c]   (gradient of the mask token)
__global__ void masked_colsum_k(const bf16_t* __restrict__ x, int64_t ld, const int* __restrict__ sel, float* __restrict__ out,
                                int64_t R, int C, int rows_per_slab) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
  const int64_t r1 = r0 + rows_per_slab < R ? r0 + rows_per_slab : R;
  float acc = 0.f;
  if (c < C)
    for (int64_t r = r0 + rl; r < r1; r += 4)
      if (sel[r] < 0) acc += bf2f(x[r * ld + c]);
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && c < C) unsafeAtomicAdd(&out[c], red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}
// out[m, :] = x[m, :] + gate[m / rows_per_mod, :] * t[m, :]
__global__ void gated_residual_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ t, const bf16_t* __restrict__ gate,
                                 int64_t ld_gate, int64_t rows_per_mod, bf16_t* __restrict__ out, int64_t ld_out, int D8,
                                 int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % D8);
    const int64_t row = i / D8;
    float a[8], b[8], g[8];
    unpack8(*(const u32x4_t*)(x + row * D8 * 8 + c * 8), a);
    unpack8(*(const u32x4_t*)(t + row * D8 * 8 + c * 8), b);
    unpack8(*(const u32x4_t*)(gate + (row / rows_per_mod) * ld_gate + c * 8), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += g[e] * b[e];
    *(u32x4_t*)(out + row * ld_out + c * 8) = pack8(a);
  }
}

__global__ void copy_rows3d_k(const bf16_t* __restrict__ src, int64_t sbs, int64_t srs, bf16_t* __restrict__ dst, int64_t dbs,
                              int64_t drs, int rows, int C8, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8);
    const int64_t br = i / C8;
    const int64_t b = br / rows, r = br - b * rows;
    *(u32x4_t*)(dst + b * dbs + r * drs + c * 8) = *(const u32x4_t*)(src + b * sbs + r * srs + c * 8);
  }
}

// DDT decoder conditioning (ddt.py:423-424 + the SiLU inside Modulation nn.py:530): z = silu(enc + temb[b]); out = silu(z)
__global__ void ddt_cond_fwd_k(const bf16_t* __restrict__ enc, int64_t ld, const float* __restrict__ temb, int64_t ld_t, int N,
                               bf16_t* __restrict__ out, int D8, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % D8);
    const int64_t row = i / D8;
    float v[8];
    unpack8(*(const u32x4_t*)(enc + row * ld + c * 8), v);
    const float* tp = temb + (row / N) * ld_t + c * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = silu_f(silu_f(v[e] + tp[e]));
    *(u32x4_t*)(out + row * (int64_t)D8 * 8 + c * 8) = pack8(v);
  }
}
// denc = dsz * silu'(z) * silu'(u), u = enc + temb[b], z = silu(u); dtemb[b, :] += sum_n denc   (one workgroup per (sample, slab))
__global__ __launch_bounds__(256) void ddt_cond_bwd_k(const bf16_t* __restrict__ dsz, const bf16_t* __restrict__ enc, int64_t ld,
                                                      const float* __restrict__ temb, int64_t ld_t, int N, bf16_t* __restrict__ denc,
                                                      float* __restrict__ dtemb, int D, int rows_per_wg) {
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_wg, r1 = min(r0 + rows_per_wg, N);
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    const float tb = temb[(int64_t)b * ld_t + c];
    float acc = 0.f;
    for (int n = r0; n < r1; ++n) {
      const int64_t row = (int64_t)b * N + n;
      const float u = bf2f(enc[row * ld + c]) + tb;
      const float su = 1.f / (1.f + __expf(-u));
      const float z = u * su;
      const float dz_du = su * (1.f + u * (1.f - su));
      const float sz = 1.f / (1.f + __expf(-z));
      const float d = bf2f(dsz[row * (int64_t)D + c]) * (sz * (1.f + z * (1.f - sz))) * dz_du;
      denc[row * (int64_t)D + c] = f2bf(d);
      acc += d;
    }
    unsafeAtomicAdd(dtemb + (int64_t)b * ld_t + c, acc);
  }
}

static inline int row_grid(int64_t total) {
  int64_t g = (total + 255) / 256;
  return (int)(g > 8192 ? 8192 : g);
}
#define ALIGNED16(p) ((((uintptr_t)(p)) & 15) == 0)

extern "C" int dl_gather_tokens(const void* src, int64_t ld_src, const int32_t* idx, const int32_t* keep, void* dst, int64_t ld_dst,
                                int64_t B, int64_t N, int64_t k, int64_t D, dl_stream_t stream) {
  DL_CHECK_ARG(src && idx && dst && B > 0 && N > 0 && k > 0 && D > 0 && D % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 &&
                   ld_src >= D && ld_dst >= D && ALIGNED16(src) && ALIGNED16(dst),
               "dl_gather_tokens: bad args");
  const int64_t total = B * k * (D / 8);
  hipLaunchKernelGGL(gather_tokens_k, row_grid(total), 256, 0, (hipStream_t)stream, (const bf16_t*)src, ld_src, idx, keep,
                     (bf16_t*)dst, ld_dst, (int)N, (int)k, (int)(D / 8), total);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_scatter_tokens_add(const void* src, int64_t ld_src, const int32_t* idx, void* dst, int64_t ld_dst, int64_t B,
                                     int64_t N, int64_t k, int64_t D, dl_stream_t stream) {
  DL_CHECK_ARG(src && idx && dst && B > 0 && N > 0 && k > 0 && D > 0 && D % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 &&
                   ld_src >= D && ld_dst >= D && ALIGNED16(src) && ALIGNED16(dst),
               "dl_scatter_tokens_add: bad args");
  const int64_t total = B * k * (D / 8);
  hipLaunchKernelGGL(scatter_tokens_add_k, row_grid(total), 256, 0, (hipStream_t)stream, (const bf16_t*)src, ld_src, idx,
                     (bf16_t*)dst, ld_dst, (int)N, (int)k, (int)(D / 8), total);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_restore_tokens(const void* xd, int64_t ld_xd, const int32_t* inv, const float* mask, void* out, int64_t ld_out,
                                 int64_t B, int64_t N, int64_t k, int64_t D, dl_stream_t stream) {
  DL_CHECK_ARG(xd && inv && mask && out && B > 0 && N > 0 && k > 0 && D > 0 && D % 8 == 0 && ld_xd % 8 == 0 && ld_out % 8 == 0 &&
                   ld_xd >= D && ld_out >= D && ALIGNED16(xd) && ALIGNED16(out) && ALIGNED16(mask),
               "dl_restore_tokens: bad args");
  const int64_t total = B * N * (D / 8);
  hipLaunchKernelGGL(restore_tokens_k, row_grid(total), 256, 0, (hipStream_t)stream, (const bf16_t*)xd, ld_xd, inv, mask,
                     (bf16_t*)out, ld_out, (int)N, (int)k, (int)(D / 8), total);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_masked_colsum(const void* x, int64_t ld, const int32_t* sel, float* out, int64_t R, int64_t C,
                                dl_stream_t stream) {
  DL_CHECK_ARG(x && sel && out && R > 0 && C > 0 && ld >= C, "dl_masked_colsum: bad args");
  int slabs = (int)((R + 255) / 256);
  if (slabs > 512) slabs = 512;
  const int rps = (int)((R + slabs - 1) / slabs);
  hipLaunchKernelGGL(masked_colsum_k, dim3(cdiv(C, 64), slabs), 256, 0, (hipStream_t)stream, (const bf16_t*)x, ld, sel, out, R,
                     (int)C, rps);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_gated_residual_fwd(const void* x, const void* t, const void* gate, int64_t ld_gate, int64_t rows_per_mod,
                                     void* out, int64_t ld_out, int64_t M, int64_t D, dl_stream_t stream) {
  DL_CHECK_ARG(x && t && gate && out && M > 0 && D > 0 && D % 8 == 0 && ld_gate % 8 == 0 && ld_out % 8 == 0 && ld_out >= D &&
                   rows_per_mod > 0 && ALIGNED16(x) && ALIGNED16(t) && ALIGNED16(gate) && ALIGNED16(out),
               "dl_gated_residual_fwd: bad args");
  const int64_t total = M * (D / 8);
  hipLaunchKernelGGL(gated_residual_k, row_grid(total), 256, 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)t,
                     (const bf16_t*)gate, ld_gate, rows_per_mod, (bf16_t*)out, ld_out, (int)(D / 8), total);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_copy_rows3d(const void* src, int64_t src_bs, int64_t src_rs, void* dst, int64_t dst_bs, int64_t dst_rs,
                              int64_t B, int64_t rows, int64_t cols, dl_stream_t stream) {
  DL_CHECK_ARG(src && dst && B > 0 && rows > 0 && cols > 0 && cols % 8 == 0 && src_bs % 8 == 0 && src_rs % 8 == 0 &&
                   dst_bs % 8 == 0 && dst_rs % 8 == 0 && src_rs >= cols && dst_rs >= cols && ALIGNED16(src) && ALIGNED16(dst),
               "dl_copy_rows3d: bad args");
  const int64_t total = B * rows * (cols / 8);
  hipLaunchKernelGGL(copy_rows3d_k, row_grid(total), 256, 0, (hipStream_t)stream, (const bf16_t*)src, src_bs, src_rs,
                     (bf16_t*)dst, dst_bs, dst_rs, (int)rows, (int)(cols / 8), total);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_ddt_cond_fwd(const void* enc, int64_t ld, const float* temb, int64_t ld_t, int64_t B, int64_t N, int64_t D,
                               void* out, dl_stream_t stream) {
  DL_CHECK_ARG(enc && temb && out && B > 0 && N > 0 && D > 0 && D % 8 == 0 && ld % 8 == 0 && ld >= D && ld_t >= D && ld_t % 4 == 0 &&
                   ALIGNED16(enc) && ALIGNED16(out) && ALIGNED16(temb),
               "dl_ddt_cond_fwd: bad args");
  const int64_t total = B * N * (D / 8);
  hipLaunchKernelGGL(ddt_cond_fwd_k, row_grid(total), 256, 0, (hipStream_t)stream, (const bf16_t*)enc, ld, temb, ld_t, (int)N,
                     (bf16_t*)out, (int)(D / 8), total);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_ddt_cond_bwd(const void* dsz, const void* enc, int64_t ld, const float* temb, int64_t ld_t, int64_t B, int64_t N,
                               int64_t D, void* denc, float* dtemb, dl_stream_t stream) {
  DL_CHECK_ARG(dsz && enc && temb && denc && dtemb && B > 0 && N > 0 && D > 0 && ld >= D && ld_t >= D, "dl_ddt_cond_bwd: bad args");
  const int rows_per_wg = 16;
  hipLaunchKernelGGL(ddt_cond_bwd_k, dim3(cdiv(N, rows_per_wg), (unsigned)B), 256, 0, (hipStream_t)stream, (const bf16_t*)dsz,
                     (const bf16_t*)enc, ld, temb, ld_t, (int)N, (bf16_t*)denc, dtemb, (int)D, rows_per_wg);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
