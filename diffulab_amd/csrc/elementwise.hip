// HBM-bound kernels: diffusion heads (noising, losses), sampler steps, optimizer, casts.
// All are float4 / 16-byte vectorised grid-stride loops (cdna_hip_programming.md Appendix B "Element-wise").
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"

// ---------------------------------------------------------------- error plumbing / library info
static thread_local char g_err[512] = "";
void dl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* dl_last_error(void) { return g_err; }
static thread_local int g_wg_cap = 0;
void dl_set_wg_cap(int cap) { g_wg_cap = cap > 0 ? cap : 0; }
int dl_get_wg_cap() { return g_wg_cap; }
int dl_wg_budget(int n_cu) {
  int n = (g_wg_cap > 0 && g_wg_cap < n_cu) ? g_wg_cap : n_cu;
  return n < 8 ? 8 : n;
}
extern "C" int dl_version(void) { return 100; }
extern "C" int dl_device_info(int device, int* cu, int* lds, int64_t* hbm, char* arch, int arch_len) {
  hipDeviceProp_t p;
  hipError_t e = hipGetDeviceProperties(&p, device);
  if (e != hipSuccess) {
    dl_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e));
    return DL_ERR_LAUNCH;
  }
  if (cu) *cu = p.multiProcessorCount;
  if (lds) *lds = (int)p.maxSharedMemoryPerMultiProcessor;
  if (hbm) *hbm = (int64_t)p.totalGlobalMem;
  if (arch && arch_len > 0) {
    strncpy(arch, p.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return DL_OK;
}

// HIP stream restricted to a subset of the compute units (hipExtStreamCreateWithCUMask): the engines put the weight-gradient
// GEMMs on such a stream so that they cannot crowd the latency-bound kernels of the main dependency chain off the CUs.
extern "C" int dl_stream_create_masked(const uint32_t* cu_mask, int words, void** stream_out) {
  DL_CHECK_ARG(cu_mask && words > 0 && stream_out, "dl_stream_create_masked: bad args");
  hipStream_t st = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, cu_mask);
  if (e != hipSuccess) {
    dl_set_error("hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
    return DL_ERR_LAUNCH;
  }
  *stream_out = (void*)st;
  return DL_OK;
}
// HIP stream of the LOWEST priority the device offers (the side stream of the weight gradients: when both queues have workgroups
// ready the dispatcher serves the main chain's first); *range_out = {least, greatest} as hipDeviceGetStreamPriorityRange reports it
extern "C" int dl_stream_create_low_priority(void** stream_out, int* range_out) {
  DL_CHECK_ARG(stream_out, "dl_stream_create_low_priority: bad args");
  int least = 0, greatest = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
  if (e != hipSuccess) {
    dl_set_error("hipDeviceGetStreamPriorityRange: %s", hipGetErrorString(e));
    return DL_ERR_LAUNCH;
  }
  if (range_out) {
    range_out[0] = least;
    range_out[1] = greatest;
  }
  hipStream_t st = nullptr;
  e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least);
  if (e != hipSuccess) {
    dl_set_error("hipStreamCreateWithPriority: %s", hipGetErrorString(e));
    return DL_ERR_LAUNCH;
  }
  *stream_out = (void*)st;
  return DL_OK;
}
extern "C" int dl_stream_destroy(void* stream) {
  if (stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess) return DL_ERR_LAUNCH;
  return DL_OK;
}

static inline int ew_grid(int64_t nvec, int threads = 256) {
  int64_t g = (nvec + threads - 1) / threads;
  if (g > 2048) g = 2048;  // 256 CUs x 8 blocks, grid-stride beyond that
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------- noising
// z = (1-t) x + t eps; chw % 4 == 0 fast path works on float4, else scalar
template <bool VEC>
__global__ void flow_add_noise_k(const float* __restrict__ x, const float* __restrict__ e,
                                 const float* __restrict__ t, float* __restrict__ z, int64_t n, int64_t chw) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (VEC) {
    const int64_t nv = n >> 2, cv = chw >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
      const float tt = t[i / cv], at = 1.0f - tt;
      const float4 a = ((const float4*)x)[i], b = ((const float4*)e)[i];
      float4 o;
      o.x = at * a.x + tt * b.x;
      o.y = at * a.y + tt * b.y;
      o.z = at * a.z + tt * b.z;
      o.w = at * a.w + tt * b.w;
      ((float4*)z)[i] = o;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
      const float tt = t[i / chw];
      z[i] = (1.0f - tt) * x[i] + tt * e[i];
    }
  }
}
extern "C" int dl_flow_add_noise(const float* x, const float* noise, const float* t, float* z, int64_t batch,
                                 int64_t chw, dl_stream_t stream) {
  DL_CHECK_ARG(x && noise && t && z && batch > 0 && chw > 0, "dl_flow_add_noise: bad args");
  const int64_t n = batch * chw;
  if (chw % 4 == 0)
    hipLaunchKernelGGL(flow_add_noise_k<true>, ew_grid(n / 4), 256, 0, (hipStream_t)stream, x, noise, t, z, n, chw);
  else
    hipLaunchKernelGGL(flow_add_noise_k<false>, ew_grid(n), 256, 0, (hipStream_t)stream, x, noise, t, z, n, chw);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

__global__ void ddpm_add_noise_k(const float* __restrict__ x, const float* __restrict__ e,
                                 const int32_t* __restrict__ t, const float* __restrict__ sqrt_ab,
                                 const float* __restrict__ ab, float* __restrict__ o, int64_t n, int64_t chw) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int32_t ti = t[i / chw];
    const float ca = sqrt_ab[ti];
    const float cb = sqrtf(1.0f - ab[ti]);  // gd.py:340: sqrt evaluated in fp32 after the .float() gather
    o[i] = ca * x[i] + cb * e[i];
  }
}
extern "C" int dl_ddpm_add_noise(const float* x, const float* noise, const int32_t* t, const float* sqrt_ab,
                                 const float* ab, float* xt, int64_t batch, int64_t chw, dl_stream_t stream) {
  DL_CHECK_ARG(x && noise && t && sqrt_ab && ab && xt && batch > 0 && chw > 0, "dl_ddpm_add_noise: bad args");
  const int64_t n = batch * chw;
  hipLaunchKernelGGL(ddpm_add_noise_k, ew_grid(n), 256, 0, (hipStream_t)stream, x, noise, t, sqrt_ab, ab, xt, n, chw);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ---------------------------------------------------------------- MSE heads
#define MSE_BLOCK 256
#define MSE_ELEMS_PER_BLOCK (MSE_BLOCK * 16)
extern "C" int64_t dl_mse_loss_partials(int64_t n) { return (n + MSE_ELEMS_PER_BLOCK - 1) / MSE_ELEMS_PER_BLOCK; }

__global__ void mse_partial_k(const float* __restrict__ pred, const float* __restrict__ a,
                              const float* __restrict__ b, float* __restrict__ partial, int64_t n, int mode) {
  __shared__ float red[MSE_BLOCK / DL_WAVE];
  const int64_t base = (int64_t)blockIdx.x * MSE_ELEMS_PER_BLOCK;
  float acc = 0.f;
#pragma unroll 4
  for (int j = 0; j < 16; ++j) {
    const int64_t i = base + (int64_t)j * MSE_BLOCK + threadIdx.x;
    if (i < n) {
      const float tgt = (mode == DL_LOSS_FLOW) ? (a[i] - b[i]) : a[i];
      const float d = tgt - pred[i];
      acc += d * d;
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < MSE_BLOCK / DL_WAVE; ++w) s += red[w];
    partial[blockIdx.x] = s;
  }
}
__global__ void mse_final_k(const float* __restrict__ partial, float* __restrict__ loss, int64_t np, float inv_n) {
  __shared__ float red[256 / DL_WAVE];
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < np; i += 256) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *loss = (red[0] + red[1] + red[2] + red[3]) * inv_n;
}
extern "C" int dl_mse_loss_fwd(const float* pred, const float* a, const float* b, float* partial, float* loss,
                               int64_t n, int mode, dl_stream_t stream) {
  DL_CHECK_ARG(pred && a && partial && loss && n > 0, "dl_mse_loss_fwd: bad args");
  DL_CHECK_ARG(mode == DL_LOSS_EPS || b, "dl_mse_loss_fwd: flow mode needs b (x0)");
  const int64_t np = dl_mse_loss_partials(n);
  hipLaunchKernelGGL(mse_partial_k, (int)np, MSE_BLOCK, 0, (hipStream_t)stream, pred, a, b, partial, n, mode);
  hipLaunchKernelGGL(mse_final_k, 1, 256, 0, (hipStream_t)stream, partial, loss, np, (float)(1.0 / (double)n));
  DL_LAUNCH_CHECK();
  return DL_OK;
}
__global__ void mse_bwd_k(const float* __restrict__ pred, const float* __restrict__ a, const float* __restrict__ b,
                          float coef, const float* __restrict__ gdev, float* __restrict__ dpred, int64_t n, int mode) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (gdev) coef *= *gdev;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float tgt = (mode == DL_LOSS_FLOW) ? (a[i] - b[i]) : a[i];
    dpred[i] = coef * (pred[i] - tgt);
  }
}
extern "C" int dl_mse_loss_bwd(const float* pred, const float* a, const float* b, float gscale,
                               const float* gscale_dev, float* dpred, int64_t n, int mode, dl_stream_t stream) {
  DL_CHECK_ARG(pred && a && dpred && n > 0, "dl_mse_loss_bwd: bad args");
  DL_CHECK_ARG(mode == DL_LOSS_EPS || b, "dl_mse_loss_bwd: flow mode needs b (x0)");
  const float coef = (float)(2.0 * (double)gscale / (double)n);
  hipLaunchKernelGGL(mse_bwd_k, ew_grid(n), 256, 0, (hipStream_t)stream, pred, a, b, coef, gscale_dev, dpred, n, mode);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

__global__ void x_to_v_k(const float* __restrict__ z, const float* __restrict__ xh, const float* __restrict__ t,
                         float* __restrict__ v, int64_t n, int64_t chw) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    v[i] = (z[i] - xh[i]) / t[i / chw];
}
__global__ void x_to_v_bwd_k(const float* __restrict__ dv, const float* __restrict__ t, float* __restrict__ dx,
                             int64_t n, int64_t chw) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dx[i] = -dv[i] / t[i / chw];
}
extern "C" int dl_flow_x_to_v(const float* z, const float* xhat, const float* t, float* v, int64_t batch,
                              int64_t chw, dl_stream_t stream) {
  DL_CHECK_ARG(z && xhat && t && v && batch > 0 && chw > 0, "dl_flow_x_to_v: bad args");
  hipLaunchKernelGGL(x_to_v_k, ew_grid(batch * chw), 256, 0, (hipStream_t)stream, z, xhat, t, v, batch * chw, chw);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_flow_x_to_v_bwd(const float* dv, const float* t, float* dxhat, int64_t batch, int64_t chw,
                                  dl_stream_t stream) {
  DL_CHECK_ARG(dv && t && dxhat && batch > 0 && chw > 0, "dl_flow_x_to_v_bwd: bad args");
  hipLaunchKernelGGL(x_to_v_bwd_k, ew_grid(batch * chw), 256, 0, (hipStream_t)stream, dv, t, dxhat, batch * chw, chw);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ---------------------------------------------------------------- sampler steps
__global__ void euler_step_k(const float* __restrict__ x, const float* __restrict__ v, const float* __restrict__ vu,
                             float g, float t_curr, float dt, float* __restrict__ xp, float* __restrict__ x0,
                             int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float vv = v[i];
    if (vu) {
      const float u = vu[i];
      vv = u + g * (vv - u);
    }
    const float xi = x[i];
    xp[i] = xi - vv * dt;
    if (x0) x0[i] = xi - vv * t_curr;
  }
}
extern "C" int dl_euler_step(const float* x, const float* v, const float* v_uncond, float guidance, float t_curr,
                             float dt, float* x_prev, float* x0_est, int64_t n, dl_stream_t stream) {
  DL_CHECK_ARG(x && v && x_prev && n > 0, "dl_euler_step: bad args");
  hipLaunchKernelGGL(euler_step_k, ew_grid(n), 256, 0, (hipStream_t)stream, x, v, v_uncond, guidance, t_curr, dt,
                     x_prev, x0_est, n);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

__global__ void em_step_k(const float* __restrict__ x, const float* __restrict__ v, const float* __restrict__ vu,
                          float g, const float* __restrict__ noise, const float* __restrict__ xpin, float t_curr,
                          float dt, float drift_c, float std, float log_std, float* __restrict__ xp,
                          float* __restrict__ mean, float* __restrict__ x0, float* __restrict__ lp, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float half_log_2pi = 0.91893853320467274178f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float vv = v[i];
    if (vu) {
      const float u = vu[i];
      vv = u + g * (vv - u);
    }
    const float xi = x[i];
    // x - (v + sigma^2/(2t) (x + (1-t) v)) dt
    const float m = xi - (vv + drift_c * (xi + (1.0f - t_curr) * vv)) * dt;
    const float xpv = xpin ? xpin[i] : (m + std * noise[i]);
    xp[i] = xpv;
    if (mean) mean[i] = m;
    if (x0) x0[i] = xi - vv * t_curr;
    if (lp) {
      const float d = xpv - m;
      lp[i] = -(d * d / (2.0f * std * std) + log_std + half_log_2pi);
    }
  }
}
extern "C" int dl_euler_maruyama_step(const float* x, const float* v, const float* v_uncond, float guidance,
                                      const float* noise, const float* x_prev_in, float t_curr, float dt,
                                      float sigma, float std, float* x_prev, float* mean, float* x0_est,
                                      float* logprob, int64_t n, dl_stream_t stream) {
  DL_CHECK_ARG(x && v && x_prev && n > 0, "dl_euler_maruyama_step: bad args");
  DL_CHECK_ARG((noise != nullptr) != (x_prev_in != nullptr), "dl_euler_maruyama_step: exactly one of noise/x_prev_in");
  const float drift_c = (float)((double)sigma * (double)sigma / (2.0 * (double)t_curr));
  hipLaunchKernelGGL(em_step_k, ew_grid(n), 256, 0, (hipStream_t)stream, x, v, v_uncond, guidance, noise, x_prev_in,
                     t_curr, dt, drift_c, std, logf(std), x_prev, mean, x0_est, logprob, n);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

__device__ __forceinline__ float x_start_from(int mean_type, float out, float xt, float sab, float ab, float c1,
                                              float c2) {
  if (mean_type == DL_MEAN_XSTART) return out;
  if (mean_type == DL_MEAN_EPSILON) return (1.0f / sab) * xt - (sqrtf(1.0f - ab) / sab) * out;  // ddpm.py:118-121
  return (1.0f / c1) * out - (c2 / c1) * xt;                                                    // ddpm.py:99-102
}

__global__ void ddpm_step_k(const float* __restrict__ pred, const float* __restrict__ pu, float g,
                            const float* __restrict__ xt, const float* __restrict__ noise,
                            const int32_t* __restrict__ t, const float* __restrict__ tab, int32_t T, int mean_type,
                            int clamp_x, float* __restrict__ xp, float* __restrict__ x0o, float* __restrict__ mo,
                            float* __restrict__ so, float* __restrict__ lp, int64_t n, int64_t chw) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int32_t ti = t[i / chw];
    const float sab = tab[ti], ab = tab[T + ti], c1 = tab[2 * T + ti], c2 = tab[3 * T + ti];
    const float var = tab[4 * T + ti], lv = tab[5 * T + ti];
    float o = pred[i];
    if (pu) {
      const float u = pu[i];
      o = u + g * (o - u);
    }
    const float x = xt[i];
    float x0 = x_start_from(mean_type, o, x, sab, ab, c1, c2);
    if (clamp_x) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    const float m = c1 * x0 + c2 * x;
    const float mask = ti > 0 ? 1.0f : 0.0f;
    const float xpv = m + mask * noise[i] * expf(0.5f * lv);
    xp[i] = xpv;
    if (x0o) x0o[i] = x0;
    if (mo) mo[i] = m;
    const float vs = fmaxf(var, 1e-20f);
    if (so) so[i] = sqrtf(vs);
    if (lp) {
      const float d = xpv - m;
      lp[i] = (-(d * d) / (2.0f * vs) - logf(6.283185307179586f * vs) * 0.5f) * mask;
    }
  }
}
extern "C" int dl_ddpm_step(const float* pred, const float* pred_uncond, float guidance, const float* xt,
                            const float* noise, const int32_t* t, const float* tables, int32_t T, int mean_type,
                            int clamp_x, float* x_prev, float* x0_est, float* mean, float* std, float* logprob,
                            int64_t batch, int64_t chw, dl_stream_t stream) {
  DL_CHECK_ARG(pred && xt && noise && t && tables && x_prev && batch > 0 && chw > 0 && T > 0, "dl_ddpm_step: bad args");
  DL_CHECK_ARG(mean_type >= 0 && mean_type <= 2, "dl_ddpm_step: mean_type %d", mean_type);
  const int64_t n = batch * chw;
  hipLaunchKernelGGL(ddpm_step_k, ew_grid(n), 256, 0, (hipStream_t)stream, pred, pred_uncond, guidance, xt, noise, t,
                     tables, T, mean_type, clamp_x, x_prev, x0_est, mean, std, logprob, n, chw);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

__global__ void ddim_step_k(const float* __restrict__ pred, const float* __restrict__ pu, float g,
                            const float* __restrict__ xt, const float* __restrict__ noise,
                            const int32_t* __restrict__ t, const float* __restrict__ tab, int32_t T, int mean_type,
                            const float* __restrict__ coefs, int clamp_x, float eta, float* __restrict__ xp,
                            float* __restrict__ x0o, float* __restrict__ mo, float* __restrict__ so,
                            float* __restrict__ lp, int64_t n, int64_t chw) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float half_log_2pi = 0.91893853320467274178f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int32_t ti = t[i / chw];
    const float sab = tab[ti], ab = tab[T + ti], abp = tab[2 * T + ti];
    const float c1 = coefs ? coefs[ti] : 1.0f, c2 = coefs ? coefs[T + ti] : 0.0f;
    float o = pred[i];
    if (pu) {
      const float u = pu[i];
      o = u + g * (o - u);
    }
    const float x = xt[i];
    float x0 = x_start_from(mean_type, o, x, sab, ab, c1, c2);
    if (clamp_x) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    const float eps = ((1.0f / sab) * x - x0) / sqrtf(1.0f / ab - 1.0f);  // ddpm.py:324-326
    const float sigma = eta * sqrtf((1.0f - abp) / (1.0f - ab)) * sqrtf(1.0f - ab / abp);
    const float m = x0 * sqrtf(abp) + sqrtf(1.0f - abp - sigma * sigma) * eps;
    const float mask = ti > 0 ? 1.0f : 0.0f;
    const float xpv = m + mask * sigma * noise[i];
    xp[i] = xpv;
    if (x0o) x0o[i] = x0;
    if (mo) mo[i] = m;
    if (so) so[i] = sigma;
    if (lp) {
      const float d = xpv - m;
      lp[i] = -(d * d / (2.0f * sigma * sigma) + logf(sigma) + half_log_2pi);
    }
  }
}
extern "C" int dl_ddim_step(const float* pred, const float* pred_uncond, float guidance, const float* xt,
                            const float* noise, const int32_t* t, const float* tables, int32_t T, int mean_type,
                            const float* ddpm_coefs, int clamp_x, float eta, float* x_prev, float* x0_est,
                            float* mean, float* std, float* logprob, int64_t batch, int64_t chw,
                            dl_stream_t stream) {
  DL_CHECK_ARG(pred && xt && noise && t && tables && x_prev && batch > 0 && chw > 0 && T > 0, "dl_ddim_step: bad args");
  DL_CHECK_ARG(mean_type != DL_MEAN_XPREV || ddpm_coefs, "dl_ddim_step: xprev needs ddpm_coefs");
  const int64_t n = batch * chw;
  hipLaunchKernelGGL(ddim_step_k, ew_grid(n), 256, 0, (hipStream_t)stream, pred, pred_uncond, guidance, xt, noise, t,
                     tables, T, mean_type, ddpm_coefs, clamp_x, eta, x_prev, x0_est, mean, std, logprob, n, chw);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ---------------------------------------------------------------- optimizer / casts
__global__ void adamw_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                        float* __restrict__ v, int64_t n4, int64_t n, float lr, float b1, float b2, float eps,
                        float wd, float inv_bc1_lr, float inv_sqrt_bc2, float gs) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = ((float4*)p)[i], gg = ((const float4*)g)[i], mm = ((float4*)m)[i], vv = ((float4*)v)[i];
    float* P = (float*)&pp;
    float* G = (float*)&gg;
    float* Mm = (float*)&mm;
    float* V = (float*)&vv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = G[j] * gs;
      float pj = P[j] * (1.0f - lr * wd);
      const float mj = b1 * Mm[j] + (1.0f - b1) * gr;
      const float vj = b2 * V[j] + (1.0f - b2) * gr * gr;
      const float denom = sqrtf(vj) * inv_sqrt_bc2 + eps;
      pj -= inv_bc1_lr * (mj / denom);
      P[j] = pj;
      Mm[j] = mj;
      V[j] = vj;
    }
    ((float4*)p)[i] = pp;
    ((float4*)m)[i] = mm;
    ((float4*)v)[i] = vv;
  }
  // tail (n % 4)
  const int64_t tail0 = n4 * 4;
  const int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi < n - tail0) {
    const int64_t i = tail0 + gi;
    const float gr = g[i] * gs;
    float pj = p[i] * (1.0f - lr * wd);
    const float mj = b1 * m[i] + (1.0f - b1) * gr;
    const float vj = b2 * v[i] + (1.0f - b2) * gr * gr;
    pj -= inv_bc1_lr * (mj / (sqrtf(vj) * inv_sqrt_bc2 + eps));
    p[i] = pj;
    m[i] = mj;
    v[i] = vj;
  }
}
extern "C" int dl_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2,
                             float grad_scale, dl_stream_t stream) {
  DL_CHECK_ARG(p && g && m && v && n > 0, "dl_adamw_step: bad args");
  DL_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "dl_adamw_step: 16B alignment");
  const float inv_bc1_lr = (float)((double)lr / (double)bias_corr1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt((double)bias_corr2));
  hipLaunchKernelGGL(adamw_k, ew_grid(n / 4 + 1), 256, 0, (hipStream_t)stream, p, g, m, v, n / 4, n, lr, beta1, beta2,
                     eps, weight_decay, inv_bc1_lr, inv_sqrt_bc2, grad_scale);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// same update with the scalars read from device memory (hyper = {lr, beta1, beta2, eps, weight_decay, lr / bias_corr1,
// 1 / sqrt(bias_corr2), grad_scale}): the launch carries no per-step value, so a captured hipGraph of the training step can be
// replayed while the host refreshes the 32 bytes before each replay (training/graph_step.py)
__global__ void adamw_dev_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            int64_t n, const float* __restrict__ hyper) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], inv_bc1_lr = hyper[5],
              inv_sqrt_bc2 = hyper[6], gs = hyper[7];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gr = g[i] * gs;
    float pj = p[i] * (1.0f - lr * wd);
    const float mj = b1 * m[i] + (1.0f - b1) * gr;
    const float vj = b2 * v[i] + (1.0f - b2) * gr * gr;
    pj -= inv_bc1_lr * (mj / (sqrtf(vj) * inv_sqrt_bc2 + eps));
    p[i] = pj;
    m[i] = mj;
    v[i] = vj;
  }
}
extern "C" int dl_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, dl_stream_t stream) {
  DL_CHECK_ARG(p && g && m && v && hyper && n > 0, "dl_adamw_step_dev: bad args");
  hipLaunchKernelGGL(adamw_dev_k, ew_grid(n), 256, 0, (hipStream_t)stream, p, g, m, v, n, hyper);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

__global__ void ema_k(float* __restrict__ ema, const float* __restrict__ p, float w, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float e = ema[i];
    ema[i] = e + w * (p[i] - e);
  }
}
extern "C" int dl_ema_update(float* ema, const float* p, float beta, int64_t n, dl_stream_t stream) {
  DL_CHECK_ARG(ema && p && n > 0, "dl_ema_update: bad args");
  hipLaunchKernelGGL(ema_k, ew_grid(n), 256, 0, (hipStream_t)stream, ema, p, 1.0f - beta, n);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

__global__ void cast_f2b_k(const float* __restrict__ s, bf16_t* __restrict__ d, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = f2bf(s[i]);
}
__global__ void cast_b2f_k(const bf16_t* __restrict__ s, float* __restrict__ d, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = bf2f(s[i]);
}
extern "C" int dl_cast_f32_to_bf16(const float* src, void* dst, int64_t n, dl_stream_t stream) {
  DL_CHECK_ARG(src && dst && n > 0, "dl_cast_f32_to_bf16: bad args");
  hipLaunchKernelGGL(cast_f2b_k, ew_grid(n), 256, 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
// strided rows: dst[r, c] = bf16(src[r, c]) for r < rows, c < cols (cols % 4 == 0; both row strides in elements)
__global__ void cast2d_f2b_k(const float* __restrict__ s, int64_t lds, bf16_t* __restrict__ d, int64_t ldd, int64_t rows, int cols4) {
  const int64_t n = rows * cols4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols4;
    const int c = (int)(i - r * cols4) * 4;
    const f32x4_t v = *(const f32x4_t*)(s + r * lds + c);
    uint2 o;
    o.x = pack2bf(v[0], v[1]);
    o.y = pack2bf(v[2], v[3]);
    *(uint2*)(d + r * ldd + c) = o;
  }
}
extern "C" int dl_cast2d_f32_to_bf16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int64_t cols,
                                     dl_stream_t stream) {
  DL_CHECK_ARG(src && dst && rows > 0 && cols > 0 && cols % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 &&
                   (((uintptr_t)src & 15) | ((uintptr_t)dst & 7)) == 0,
               "dl_cast2d_f32_to_bf16: cols, strides %% 4 and 16 / 8-byte alignment");
  hipLaunchKernelGGL(cast2d_f2b_k, ew_grid(rows * cols / 4), 256, 0, (hipStream_t)stream, src, ld_src, (bf16_t*)dst, ld_dst, rows,
                     (int)(cols / 4));
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_cast_bf16_to_f32(const void* src, float* dst, int64_t n, dl_stream_t stream) {
  DL_CHECK_ARG(src && dst && n > 0, "dl_cast_bf16_to_f32: bad args");
  hipLaunchKernelGGL(cast_b2f_k, ew_grid(n), 256, 0, (hipStream_t)stream, (const bf16_t*)src, dst, n);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// f32 [R,C] -> bf16 dst [R, ld_dst] and/or transposed dstT [C, ld_t]; 32x32 tiles through LDS so both the
// read and the two writes are coalesced.  Padding columns are zero-filled.
__global__ void cast_weight_k(const float* __restrict__ src, int64_t R, int64_t C, bf16_t* __restrict__ dst,
                              int64_t ld_dst, bf16_t* __restrict__ dstT, int64_t ld_t) {
  __shared__ float tile[32][33];
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t r = r0 + ty + j * 8, c = c0 + tx;
    const float v = (r < R && c < C) ? src[r * C + c] : 0.0f;
    tile[ty + j * 8][tx] = v;
    if (dst && r < R && c < ld_dst) dst[r * ld_dst + c] = f2bf(v);
  }
  __syncthreads();
  if (dstT) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t c = c0 + ty + j * 8, r = r0 + tx;  // dstT[c][r]
      if (c < C && r < ld_t) dstT[c * ld_t + r] = f2bf(tile[tx][ty + j * 8]);
    }
  }
}
extern "C" int dl_cast_weight(const float* src, int64_t R, int64_t C, void* dst, int64_t ld_dst, void* dstT,
                              int64_t ld_t, dl_stream_t stream) {
  DL_CHECK_ARG(src && R > 0 && C > 0 && (dst || dstT), "dl_cast_weight: bad args");
  DL_CHECK_ARG(!dst || ld_dst >= C, "dl_cast_weight: ld_dst < C");
  DL_CHECK_ARG(!dstT || ld_t >= R, "dl_cast_weight: ld_t < R");
  const int64_t cmax = dst ? (ld_dst > C ? ld_dst : C) : C;
  const int64_t rmax = dstT ? (ld_t > R ? ld_t : R) : R;
  dim3 grid(cdiv(cmax, 32), cdiv(rmax, 32));
  hipLaunchKernelGGL(cast_weight_k, grid, 256, 0, (hipStream_t)stream, src, R, C, (bf16_t*)dst, ld_dst, (bf16_t*)dstT,
                     ld_t);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// every bf16 weight shadow of a network in ONE launch: a device table of descriptors, one 32x32 tile per workgroup
__global__ void cast_weights_batched_k(const dl_cast_desc_t* __restrict__ desc, int n_desc) {
  __shared__ float tile[32][33];
  int lo = 0, hi = n_desc - 1;  // last descriptor whose tile_begin <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid].tile_begin <= (int64_t)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const dl_cast_desc_t d = desc[lo];
  const int64_t t = (int64_t)blockIdx.x - d.tile_begin;
  const int64_t r0 = (t / d.tiles_c) * 32, c0 = (t % d.tiles_c) * 32;
  const float* src = (const float*)d.src;
  bf16_t* dst = (bf16_t*)d.dst;
  bf16_t* dstT = (bf16_t*)d.dst_t;
  bf16_t* dstP = (bf16_t*)d.dst_swiglu;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t r = r0 + ty + j * 8, c = c0 + tx;
    const float v = (r < d.R && c < d.C) ? src[r * d.C + c] : 0.0f;
    tile[ty + j * 8][tx] = v;
    if (dst && r < d.R && c < d.ld_dst) dst[r * d.ld_dst + c] = f2bf(v);
    if (dstP && r < d.R && c < d.ld_swiglu) {  // row order of dl_cast_weight_swiglu: 32-row groups [16 x1 | 16 x3]
      const int64_t F = d.R >> 1;
      const int64_t rr = r < F ? r : r - F;
      dstP[((rr >> 4) * 32 + (r < F ? 0 : 16) + (rr & 15)) * d.ld_swiglu + c] = f2bf(v);
    }
  }
  __syncthreads();
  if (dstT) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t c = c0 + ty + j * 8, r = r0 + tx;
      if (c < d.C && r < d.ld_t) dstT[c * d.ld_t + r] = f2bf(tile[tx][ty + j * 8]);
    }
  }
}
extern "C" int dl_cast_weights_batched(const dl_cast_desc_t* desc_dev, int n_desc, int64_t total_tiles, dl_stream_t stream) {
  DL_CHECK_ARG(desc_dev && n_desc > 0 && total_tiles > 0 && total_tiles < (1ll << 31), "dl_cast_weights_batched: bad args");
  hipLaunchKernelGGL(cast_weights_batched_k, (int)total_tiles, 256, 0, (hipStream_t)stream, desc_dev, n_desc);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// row-permuted bf16 copy of the packed SwiGLU weight (see dl_gemm_nt_swiglu): dst row n' <- src row perm(n')
__global__ void cast_weight_swiglu_k(const float* __restrict__ src, int F, int C, bf16_t* __restrict__ dst, int64_t ld) {
  const int64_t total = (int64_t)2 * F * ld, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % ld);
    const int np = (int)(i / ld);
    const int q = np >> 5, s = np & 31;  // 32-row groups: 16 x1 rows then the 16 x3 rows of the same hidden units
    const int srow = (s < 16) ? (16 * q + s) : (F + 16 * q + (s - 16));
    dst[i] = c < C ? f2bf(src[(int64_t)srow * C + c]) : (bf16_t)0;
  }
}
extern "C" int dl_cast_weight_swiglu(const float* src, int64_t F, int64_t C, void* dst, int64_t ld_dst, dl_stream_t stream) {
  DL_CHECK_ARG(src && dst && F > 0 && F % 16 == 0 && C > 0 && ld_dst >= C, "dl_cast_weight_swiglu: bad args");
  hipLaunchKernelGGL(cast_weight_swiglu_k, ew_grid(2 * F * ld_dst), 256, 0, (hipStream_t)stream, src, (int)F, (int)C,
                     (bf16_t*)dst, ld_dst);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
