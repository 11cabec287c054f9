// UNet (guided-diffusion style) kernels that are not GEMMs: layout changes, GroupNorm32 (+FiLM +SiLU) fwd/bwd,
// im2col for the 3x3 convolutions (the contraction itself runs on the MFMA GEMMs of gemm.hip), 2x2 pool / nearest
// upsample, and the small-sequence attention of AttentionBlock (<= 64 tokens, head_dim up to 512).
// Activations are NHWC bf16 inside the library ([B*H*W, C] token rows: a 3x3 conv is then im2col + NT GEMM with
// K = 9*C contiguous per tap); the reference's NCHW f32 tensors are converted at the model boundary.
// Reference: networks/utils/nn.py:11-88, networks/denoisers/unet.py:215-237,296-322.
#include "common.h"

static inline int grid_for(int64_t n, int threads = 256, int cap = 4096) {
  int64_t g = (n + threads - 1) / threads;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------- layout: NCHW f32 <-> NHWC bf16
__global__ void nchw_to_nhwc_k(const float* __restrict__ x, bf16_t* __restrict__ o, int B, int C, int HW, int ld) {
  const int64_t n = (int64_t)B * C * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t r = i / C;  // b*HW + p
    const int p = (int)(r % HW), b = (int)(r / HW);
    o[r * ld + c] = f2bf(x[((int64_t)b * C + c) * HW + p]);
  }
}
__global__ void nhwc_to_nchw_k(const bf16_t* __restrict__ x, float* __restrict__ o, int B, int C, int HW, int ld) {
  const int64_t n = (int64_t)B * C * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const int64_t r = i / HW;
    const int c = (int)(r % C), b = (int)(r / C);
    o[i] = bf2f(x[((int64_t)b * HW + p) * ld + c]);
  }
}
extern "C" int dl_nchw_to_nhwc(const float* x, void* out, int64_t B, int64_t C, int64_t HW, int64_t ld, dl_stream_t stream) {
  DL_CHECK_ARG(x && out && B > 0 && C > 0 && HW > 0 && ld >= C, "dl_nchw_to_nhwc: bad args");
  hipLaunchKernelGGL(nchw_to_nhwc_k, grid_for(B * C * HW), 256, 0, (hipStream_t)stream, x, (bf16_t*)out, (int)B, (int)C, (int)HW,
                     (int)ld);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_nhwc_to_nchw(const void* x, float* out, int64_t B, int64_t C, int64_t HW, int64_t ld, dl_stream_t stream) {
  DL_CHECK_ARG(x && out && B > 0 && C > 0 && HW > 0 && ld >= C, "dl_nhwc_to_nchw: bad args");
  hipLaunchKernelGGL(nhwc_to_nchw_k, grid_for(B * C * HW), 256, 0, (hipStream_t)stream, (const bf16_t*)x, out, (int)B, (int)C,
                     (int)HW, (int)ld);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ---------------------------------------------------------------- GroupNorm32 (nn.py:11-13), NHWC rows, C % 8 == 0
// All four kernels move 16 bytes (8 channels) per lane.  A thread owns an 8-channel chunk and walks pixels; C/G may be
// smaller than 8 (C = 32..224), so group indices are taken per element.
#define GN_MAXG 64
static int g_gn_fused = 1;  // 3 (tests): as 1, also where a launch would not fill the chip; 1: fused (128-channel slabs on small maps where they fit, 64-channel slabs on the large maps); 2: fused, 512-channel slabs on
                             // small maps only (round 4); 0: the separate launches (forward: statistics + apply; backward: reduce + group sums + apply)
extern "C" __attribute__((visibility("default"))) void dl_lab_set_gn_fused(int mode) { g_gn_fused = mode; }  // LAB A/B switch (not in the header)
static int gn_bwd_fused() { return g_gn_fused; }  // (the three-launch form stays for the shapes the fused kernel does not take)

// statistics: one workgroup per sample; thread (pixel lane, chunk) accumulates sum / sum of squares of its 8 channels over its pixels
// and leaves them in an LDS row of its own; thread g then adds group g's channels over the pixel lanes in a fixed order -- no atomics:
// the statistics (and with them the inference forward) are bit-reproducible.  C <= 2048 (256 chunks); wider rows: LDS float atomics.
__global__ __launch_bounds__(256) void gn_stats_k(const bf16_t* __restrict__ x, float* __restrict__ st, int HW, int C, int G,
                                                  float eps) {
  __shared__ float part[2][2048];  // [sum | sum of squares][pixel lane * C + channel]  (npl * C <= 2048)
  __shared__ float rs[GN_MAXG][2];
  const int b = blockIdx.x, cg = C / G, C8 = C >> 3;
  if (C8 <= 256) {
    const int npl = 256 / C8;
    const int chunk = (int)threadIdx.x % C8, pl = (int)threadIdx.x / C8;
    if (pl < npl) {
      float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int p = pl; p < HW; p += npl) {
        float v[8];
        unpack8(*(const u32x4_t*)(x + ((int64_t)b * HW + p) * C + chunk * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s[e] += v[e];
          q[e] += v[e] * v[e];
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        part[0][pl * C + chunk * 8 + e] = s[e];
        part[1][pl * C + chunk * 8 + e] = q[e];
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < G) {
      float S = 0.f, Q = 0.f;
      for (int r = 0; r < npl; ++r)
        for (int j = 0; j < cg; ++j) {
          S += part[0][r * C + threadIdx.x * cg + j];
          Q += part[1][r * C + threadIdx.x * cg + j];
        }
      const float n = (float)HW * cg;
      const float mu = S / n;
      st[((int64_t)b * G + threadIdx.x) * 2] = mu;
      st[((int64_t)b * G + threadIdx.x) * 2 + 1] = rsqrtf(fmaxf(Q / n - mu * mu, 0.f) + eps);
    }
    return;
  }
  if (threadIdx.x < 2 * GN_MAXG) (&rs[0][0])[threadIdx.x] = 0.f;
  __syncthreads();
  for (int ch0 = 0; ch0 < C8; ch0 += 256) {
    const int chunk = ch0 + (int)threadIdx.x;
    if (chunk < C8) {
      float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int p = 0; p < HW; ++p) {
        float v[8];
        unpack8(*(const u32x4_t*)(x + ((int64_t)b * HW + p) * C + chunk * 8), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s[e] += v[e];
          q[e] += v[e] * v[e];
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {  // (a chunk of 8 channels may straddle two groups: booked per element)
        atomicAdd(&rs[(chunk * 8 + e) / cg][0], s[e]);
        atomicAdd(&rs[(chunk * 8 + e) / cg][1], q[e]);
      }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < G) {
    const float n = (float)HW * cg;
    const float mu = rs[threadIdx.x][0] / n;
    const float var = fmaxf(rs[threadIdx.x][1] / n - mu * mu, 0.f);
    st[((int64_t)b * G + threadIdx.x) * 2] = mu;
    st[((int64_t)b * G + threadIdx.x) * 2 + 1] = rsqrtf(var + eps);
  }
}
extern "C" int dl_gn_stats(const void* x, float* stats, int64_t B, int64_t HW, int64_t C, int64_t G, float eps,
                           dl_stream_t stream) {
  DL_CHECK_ARG(x && stats && B > 0 && HW > 0 && C > 0 && G > 0 && G <= GN_MAXG && C % G == 0 && C % 8 == 0 &&
                   ((uintptr_t)x & 15) == 0,
               "dl_gn_stats: bad args (C %% 8, C %% G, G <= 64, 16-byte alignment)");
  hipLaunchKernelGGL(gn_stats_k, (int)B, 256, 0, (hipStream_t)stream, (const bf16_t*)x, stats, (int)HW, (int)C, (int)G, eps);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// out = act( (xhat*w+b) * (1+scale) + shift ), elementwise over [B*HW, C], 8 channels per thread
__global__ void gn_apply_fwd_k(const bf16_t* __restrict__ x, const float* __restrict__ st, const float* __restrict__ w,
                               const float* __restrict__ bb, const bf16_t* __restrict__ fs, const bf16_t* __restrict__ fh,
                               int64_t ldf, int silu, bf16_t* __restrict__ out, int HW, int C, int G) {
  // grid: x walks the HW*C/8 16-byte chunks of ONE sample (blockIdx.y): every index fits 32 bits, and the per-element group
  // lookup is one division per chunk (64-bit % and / per chunk plus a division per element made these kernels VALU-bound)
  const unsigned cg = C / G, C8 = C >> 3, per = (unsigned)HW * C8;
  const int b = blockIdx.y;
  const int64_t base = (int64_t)b * per;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x) {
    const unsigned c0 = (i % C8) * 8;
    float v[8], wv[8], bv[8];
    unpack8(*(const u32x4_t*)(x + (base + i) * 8), v);
    *(f32x4_t*)&wv[0] = *(const f32x4_t*)(w + c0);
    *(f32x4_t*)&wv[4] = *(const f32x4_t*)(w + c0 + 4);
    *(f32x4_t*)&bv[0] = *(const f32x4_t*)(bb + c0);
    *(f32x4_t*)&bv[4] = *(const f32x4_t*)(bb + c0 + 4);
    float sc[8], sh[8];
    if (fs) {
      unpack8(*(const u32x4_t*)(fs + (int64_t)b * ldf + c0), sc);
      unpack8(*(const u32x4_t*)(fh + (int64_t)b * ldf + c0), sh);
    }
    unsigned g = c0 / cg, rem = c0 - g * cg;
    const float* sg = st + ((int64_t)b * G + g) * 2;
    float mu = sg[0], r = sg[1];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (rem == cg) {  // next group
        rem = 0;
        sg += 2;
        mu = sg[0];
        r = sg[1];
      }
      ++rem;
      float y = (v[e] - mu) * r * wv[e] + bv[e];
      if (fs) y = y * (1.0f + sc[e]) + sh[e];
      v[e] = silu ? silu_f(y) : y;
    }
    *(u32x4_t*)(out + (base + i) * 8) = pack8(v);
  }
}
// x-extent of the per-sample grids above: enough workgroups of 256 to cover a sample once, capped so that B * gx stays moderate
static inline dim3 gn_grid(int64_t B, int64_t per) {
  int64_t gx = (per + 255) / 256;
  const int64_t cap = B >= 4096 ? 1 : 4096 / B;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  return dim3((unsigned)gx, (unsigned)B);
}
extern "C" int dl_gn_apply_fwd(const void* x, const float* stats, const float* w, const float* b, const void* film_scale,
                               const void* film_shift, int64_t ld_film, int act_silu, void* out, int64_t B, int64_t HW,
                               int64_t C, int64_t G, dl_stream_t stream) {
  DL_CHECK_ARG(x && stats && w && b && out && B > 0 && C % G == 0 && C % 8 == 0, "dl_gn_apply_fwd: bad args");
  DL_CHECK_ARG((film_scale == nullptr) == (film_shift == nullptr), "dl_gn_apply_fwd: scale and shift go together");
  DL_CHECK_ARG((((uintptr_t)x | (uintptr_t)out | (uintptr_t)w | (uintptr_t)b | (uintptr_t)film_scale | (uintptr_t)film_shift) & 15) == 0 &&
                   ld_film % 8 == 0,
               "dl_gn_apply_fwd: 16-byte alignment");
  DL_CHECK_ARG(B < 65536 && HW * C / 8 < (1ll << 31), "dl_gn_apply_fwd: B < 65536, HW*C/8 < 2^31");
  hipLaunchKernelGGL(gn_apply_fwd_k, gn_grid(B, HW * C / 8), 256, 0, (hipStream_t)stream, (const bf16_t*)x, stats, w, b,
                     (const bf16_t*)film_scale, (const bf16_t*)film_shift, ld_film, act_silu, (bf16_t*)out, (int)HW, (int)C,
                     (int)G);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// statistics + apply in ONE launch (the training shapes): a workgroup owns a slab of NCH * 8 channels (a whole number of groups) of one
// sample; every x row of a lane is loaded once and stays in registers (KEEP rows per lane: feature maps of <= KEEP * 256 / NCH pixels)
// between the statistics pass and the normalise / FiLM / SiLU pass.  Same arithmetic as gn_stats_k + gn_apply_fwd_k (f32 sum and sum of
// squares per group); statistics are written for the backward.
template <int NCH, int KEEP>
__global__ __launch_bounds__(256) void gn_fwd_fused_k(const bf16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bb,
                                                      const bf16_t* __restrict__ fs, const bf16_t* __restrict__ fh, int64_t ldf, int silu,
                                                      bf16_t* __restrict__ out, float* __restrict__ st, int HW, int C, int G, float eps) {
  constexpr int SLAB = NCH * 8, PL = 256 / NCH;
  constexpr bool POW2 = (NCH & (NCH - 1)) == 0;  // (NCH = 12: 96-channel slabs, 21 pixel lanes + 4 idle threads -- see gn_bwd_fused_k)
  constexpr int NP = POW2 ? 4 : PL;
  __shared__ __attribute__((aligned(16))) float part[NP][2][SLAB];  // [wave or pixel lane][sum | sum of squares][channel of the slab]
  __shared__ float gst[32][2];
  const int slabs = C / SLAB;
  const int b = blockIdx.x / slabs, cbase = (blockIdx.x % slabs) * SLAB;
  const int chunk = threadIdx.x % NCH, pl = threadIdx.x / NCH, wave = threadIdx.x >> 6;
  const bool live = pl < PL;
  const int c0 = cbase + (live ? chunk * 8 : 0), cg = C / G;
  const int64_t row0 = ((int64_t)b * HW) * C + c0;
  u32x4_t xk[KEEP];
#pragma unroll
  for (int i = 0; i < KEEP; ++i) {
    const int p = pl + i * PL;
    if (p < HW && live) xk[i] = *(const u32x4_t*)(x + row0 + (int64_t)p * C);
  }
  float wv[8], bv[8], sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {  // (requested now, needed after the reduction)
    wv[e] = w[c0 + e];
    bv[e] = bb[c0 + e];
    sc[e] = fs ? bf2f(fs[(int64_t)b * ldf + c0 + e]) : 0.f;
    sh[e] = fs ? bf2f(fh[(int64_t)b * ldf + c0 + e]) : 0.f;
  }
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < KEEP; ++i) {
    if (pl + i * PL < HW && live) {
      float v[8];
      unpack8(xk[i], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s[e] += v[e];
        q[e] += v[e] * v[e];
      }
    }
  }
  if constexpr (POW2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int m = NCH; m < 64; m <<= 1) {
        s[e] += __shfl_xor(s[e], m);
        q[e] += __shfl_xor(q[e], m);
      }
    }
  }
  if (POW2 ? (int)(threadIdx.x & 63) < NCH : live) {  // (partial sums in LDS rows of their own: plain stores, no LDS atomics)
    const int row = POW2 ? wave : pl;
    *(f32x4_t*)&part[row][0][chunk * 8] = f32x4_t{s[0], s[1], s[2], s[3]};
    *(f32x4_t*)&part[row][0][chunk * 8 + 4] = f32x4_t{s[4], s[5], s[6], s[7]};
    *(f32x4_t*)&part[row][1][chunk * 8] = f32x4_t{q[0], q[1], q[2], q[3]};
    *(f32x4_t*)&part[row][1][chunk * 8 + 4] = f32x4_t{q[4], q[5], q[6], q[7]};
  }
  __syncthreads();
  {
    const int g = threadIdx.x >> 3, k = threadIdx.x & 7;  // 8 threads per group (<= 32 groups per slab), three shuffles
    float S = 0.f, Q = 0.f;
    if (g < SLAB / cg)
      for (int j = k; j < cg; j += 8) {
        const int i = g * cg + j;
#pragma unroll
        for (int r = 0; r < NP; ++r) {
          S += part[r][0][i];
          Q += part[r][1][i];
        }
      }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      S += __shfl_xor(S, m);
      Q += __shfl_xor(Q, m);
    }
    if (k == 0 && g < SLAB / cg) {
      const float n = (float)HW * cg;
      const float mu = S / n;
      const float r = rsqrtf(fmaxf(Q / n - mu * mu, 0.f) + eps);
      gst[g][0] = mu;
      gst[g][1] = r;
      float* sg = st + ((int64_t)b * G + cbase / cg + g) * 2;
      sg[0] = mu;
      sg[1] = r;
    }
  }
  __syncthreads();
  float mu[8], rs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    mu[e] = gst[(chunk * 8 + e) / cg][0];
    rs[e] = gst[(chunk * 8 + e) / cg][1];
  }
#pragma unroll
  for (int i = 0; i < KEEP; ++i) {
    const int p = pl + i * PL;
    if (p < HW && live) {
      float v[8];
      unpack8(xk[i], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float y = (v[e] - mu[e]) * rs[e] * wv[e] + bv[e];
        if (fs) y = y * (1.0f + sc[e]) + sh[e];
        v[e] = silu ? silu_f(y) : y;
      }
      *(u32x4_t*)(out + row0 + (int64_t)p * C) = pack8(v);
    }
  }
}
/* GroupNorm forward in one call: statistics (written to `stats` f32 [B, G, 2] for the backward) + normalise / FiLM / SiLU.  One launch
 * where a workgroup can own whole groups of a sample with its rows in registers (every shape of the training configurations); the
 * two-kernel form (dl_gn_stats + dl_gn_apply_fwd) otherwise. */
extern "C" int dl_gn_fwd(const void* x, const float* w, const float* b, const void* film_scale, const void* film_shift, int64_t ld_film,
                         int act_silu, void* out, float* stats, int64_t B, int64_t HW, int64_t C, int64_t G, float eps, dl_stream_t stream) {
  DL_CHECK_ARG(x && w && b && out && stats && B > 0 && HW > 0 && C > 0 && G > 0 && G <= GN_MAXG && C % G == 0 && C % 8 == 0,
               "dl_gn_fwd: bad args (C %% 8, C %% G, G <= 64)");
  DL_CHECK_ARG((film_scale == nullptr) == (film_shift == nullptr), "dl_gn_fwd: scale and shift go together");
  DL_CHECK_ARG((((uintptr_t)x | (uintptr_t)out | (uintptr_t)w | (uintptr_t)b | (uintptr_t)film_scale | (uintptr_t)film_shift) & 15) == 0 &&
                   ld_film % 8 == 0,
               "dl_gn_fwd: 16-byte alignment");
  DL_CHECK_ARG(B < 65536 && HW * C / 8 < (1ll << 31), "dl_gn_fwd: B < 65536, HW*C/8 < 2^31");
  const int64_t cg = C / G;
#define GN_FWD_LAUNCH(NCH_, KEEP_)                                                                                                    \
  hipLaunchKernelGGL((gn_fwd_fused_k<NCH_, KEEP_>), dim3((unsigned)(B * (C / (NCH_ * 8)))), 256, 0, (hipStream_t)stream,              \
                     (const bf16_t*)x, w, b, (const bf16_t*)film_scale, (const bf16_t*)film_shift, ld_film, act_silu, (bf16_t*)out,   \
                     stats, (int)HW, (int)C, (int)G, eps)
  // (4 x 4 maps below ~8 M elements: the launch is a latency chain, statistics + apply as two launches are faster -- 11 vs 16 us at
  //  4 x 4 x 1024; from 8 x 8 on the one launch wins at every width: 12.2 vs 13.0 us at 8 x 8 x 512, 13.0 vs 17.4 at 8 x 8 x 768)
  if (g_gn_fused && (B * HW * C >= (1ll << 23) || HW >= 64 || g_gn_fused == 3) && B * (C / 64) < (1ll << 31)) {
    if (HW <= 64 && C % 128 == 0 && 128 % cg == 0 && 128 / cg <= 32) {
      GN_FWD_LAUNCH(16, 4);
      DL_LAUNCH_CHECK();
      return DL_OK;
    }
    if (C % 96 == 0 && 96 % cg == 0 && 128 % cg != 0 && HW <= 21 * 12) {  // 12 / 24 / 48 channels per group: 96-channel slabs, 21 pixel lanes
      if (HW <= 84) GN_FWD_LAUNCH(12, 4);
      else GN_FWD_LAUNCH(12, 12);
      DL_LAUNCH_CHECK();
      return DL_OK;
    }
    if (C % 64 == 0 && 64 % cg == 0 && 64 / cg <= 32 && HW <= 1024) {
      if (HW <= 128) GN_FWD_LAUNCH(8, 4);
      else if (HW <= 256) GN_FWD_LAUNCH(8, 8);
      else GN_FWD_LAUNCH(8, 32);
      DL_LAUNCH_CHECK();
      return DL_OK;
    }
  }
#undef GN_FWD_LAUNCH
  const int rc = dl_gn_stats(x, stats, B, HW, C, G, eps, stream);
  if (rc != DL_OK) return rc;
  return dl_gn_apply_fwd(x, stats, w, b, film_scale, film_shift, ld_film, act_silu, out, B, HW, C, G, stream);
}

// backward pass 1: per (b, c) sums over pixels:  S[b][0][c] = sum dh*y, [1] = sum dh, [2] = sum dy*xhat, [3] = sum dy
// (dh = dout * act'(h), dy = dh * (1+scale)).  One workgroup per (b, 64-channel slab, pixel range): 8 chunks x 32 pixel lanes,
// two pixels per lane and iteration in flight; the pixel ranges (gridDim.y of them, so that small batches still fill the chip)
// write PARTIAL sums Sp[range][b][4][C], which gn_group_sums_k adds up (dw / db included: one global atomic per sample and channel).
// NCH = 16-byte channel chunks per workgroup (256 / NCH pixel lanes): 8 for the high-resolution levels (64-channel slabs, 32 pixel
// lanes), 64 for feature maps of <= 64 pixels (512-channel slabs, 4 pixel lanes: a wave reads 1 KiB of ONE pixel row and a
// thread's per-channel setup is spread over 4-16 pixels instead of half a pixel)
// (The shapes the training configurations run take gn_bwd_fused_k below instead: one launch for all three passes.)
struct GnFused {
  float* dw;
  float* db;
  bf16_t* dfs;
  bf16_t* dfh;
  int64_t lddf;
  const bf16_t* dres;
  bf16_t* dx;
};
template <int NCH>
__global__ __launch_bounds__(256) void gn_bwd_reduce_k(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ x,
                                                       const float* __restrict__ st, const float* __restrict__ w,
                                                       const float* __restrict__ bb, const bf16_t* __restrict__ fs,
                                                       const bf16_t* __restrict__ fh, int64_t ldf, int silu,
                                                       float* __restrict__ Sp, int B, int HW, int C, int G) {
  constexpr int SLAB = NCH * 8, PL = 256 / NCH;
  __shared__ float red[4][SLAB];
  const int slabs = (C + SLAB - 1) / SLAB;
  const int b = blockIdx.x / slabs, cbase = (blockIdx.x % slabs) * SLAB;
  const int chunk = threadIdx.x % NCH, pl = threadIdx.x / NCH;  // NCH chunks x PL pixel lanes
  const int c0_raw = cbase + chunk * 8;
  const int ppr = (HW + gridDim.y - 1) / gridDim.y;  // pixels per range
  const int p_lo = blockIdx.y * ppr, p_hi = (p_lo + ppr < HW) ? p_lo + ppr : HW;
  for (int i = threadIdx.x; i < 4 * SLAB; i += 256) (&red[0][0])[i] = 0.f;
  __syncthreads();
  {
    // (slabs beyond C: chunks with c0 >= C compute on channel 0 of the slab and are dropped at the end, so that every
    // lane of the wave takes part in the shuffles)
    const bool live = c0_raw < C;
    const int c0 = live ? c0_raw : cbase;
    const int cg = C / G;
    float mu[8], rs[8], wv[8], bv[8], sc[8], sh[8], a0[8], a1[8], a2[8], a3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float* sg = st + ((int64_t)b * G + (c0 + e) / cg) * 2;
      mu[e] = sg[0];
      rs[e] = sg[1];
      wv[e] = w[c0 + e];
      bv[e] = bb[c0 + e];
      sc[e] = fs ? bf2f(fs[(int64_t)b * ldf + c0 + e]) : 0.f;
      sh[e] = fs ? bf2f(fh[(int64_t)b * ldf + c0 + e]) : 0.f;
      a0[e] = a1[e] = a2[e] = a3[e] = 0.f;
    }
    auto accum = [&](const u32x4_t& xr, const u32x4_t& dr) {
      float xv[8], dv[8];
      unpack8(xr, xv);
      unpack8(dr, dv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (xv[e] - mu[e]) * rs[e];
        const float y = xh * wv[e] + bv[e];
        const float h = y * (1.0f + sc[e]) + sh[e];
        const float dh = dv[e] * (silu ? dsilu_f(h) : 1.0f);
        const float dy = dh * (1.0f + sc[e]);
        a0[e] += dh * y;
        a1[e] += dh;
        a2[e] += dy * xh;
        a3[e] += dy;
      }
    };
    int p = p_lo + pl;
    for (; p + PL < p_hi; p += 2 * PL) {  // two pixels of this lane in flight
      const int64_t i0 = ((int64_t)b * HW + p) * C + c0, i1 = i0 + (int64_t)PL * C;
      const u32x4_t x0 = *(const u32x4_t*)(x + i0), d0 = *(const u32x4_t*)(dout + i0);
      const u32x4_t x1 = *(const u32x4_t*)(x + i1), d1 = *(const u32x4_t*)(dout + i1);
      accum(x0, d0);
      accum(x1, d1);
    }
    if (p < p_hi) {
      const int64_t i0 = ((int64_t)b * HW + p) * C + c0;
      accum(*(const u32x4_t*)(x + i0), *(const u32x4_t*)(dout + i0));
    }
    // fold the pixel lanes of this wave (lane bits log2(NCH)..5; none when a wave is one pixel row) with shuffles, then one LDS
    // atomic per (wave, value)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int m = NCH; m < 64; m <<= 1) {
        a0[e] += __shfl_xor(a0[e], m);
        a1[e] += __shfl_xor(a1[e], m);
        a2[e] += __shfl_xor(a2[e], m);
        a3[e] += __shfl_xor(a3[e], m);
      }
    }
    if ((int)(threadIdx.x & 63) < NCH && live) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        atomicAdd(&red[0][chunk * 8 + e], a0[e]);
        atomicAdd(&red[1][chunk * 8 + e], a1[e]);
        atomicAdd(&red[2][chunk * 8 + e], a2[e]);
        atomicAdd(&red[3][chunk * 8 + e], a3[e]);
      }
    }
  }
  __syncthreads();
  float* S = Sp + (int64_t)blockIdx.y * B * 4 * C;
  for (int i = threadIdx.x; i < SLAB && cbase + i < C; i += 256) {
#pragma unroll
    for (int k = 0; k < 4; ++k) S[((int64_t)b * 4 + k) * C + cbase + i] = red[k][i];
  }
}
// FUSED form: the workgroup owns whole groups of one sample (a slab of NCH * 8 channels = a whole number of groups, every pixel), so the
// group sums and the dx pass follow in the same launch -- one launch instead of reduce + group sums + apply.  The launch is a chain of
// memory latencies, not a stream: at B = 128 a workgroup moves 16-130 KB, so what counts is how short the chain is.
//   * Only TWO sums per channel are accumulated over the pixels, P = sum dh * xhat and Q = sum dh (dh = dout * act'(h)): scale, w and b
//     are per-(sample, channel) constants, so S0 = sum dh*y = w P + b Q, S1 = Q, S2 = sum dy*xhat = (1 + scale) P, S3 = (1 + scale) Q.
//   * The four waves leave their partial sums in LDS rows of their own (plain stores, no LDS atomics, nothing to clear).
//   * KEEP > 0 (feature maps of <= KEEP * 256 / NCH pixels): every x / dout row of the lane is loaded ONCE, up front, and stays in
//     registers for the second pass; the residual-gradient rows are requested before the reduction and arrive under it.
//     KEEP = 0: both passes walk the pixels four rows per lane in flight; the second pass re-reads x / dout through L2.
//   * The group sums are taken by 8 threads per group (<= 32 groups per slab) and three shuffles.
template <int NCH, int KEEP>
__global__ __launch_bounds__(256) void gn_bwd_fused_k(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ x,
                                                      const float* __restrict__ st, const float* __restrict__ w,
                                                      const float* __restrict__ bb, const bf16_t* __restrict__ fs,
                                                      const bf16_t* __restrict__ fh, int64_t ldf, int silu, int B, int HW, int C, int G,
                                                      GnFused fz) {
  // NCH a power of two: a wave holds whole pixel rows of the slab, its pixel lanes meet in shuffles and the four waves leave one partial
  // each.  NCH = 12 (96-channel slabs: the 12 / 24 / 48-channel groups of C = 384 / 768 / 1536): 21 pixel lanes of 12 chunks (4 threads
  // idle), every pixel lane leaves its own partial.
  constexpr int SLAB = NCH * 8, PL = 256 / NCH;
  constexpr bool POW2 = (NCH & (NCH - 1)) == 0;
  constexpr int NP = POW2 ? 4 : PL;
  __shared__ __attribute__((aligned(16))) float part[NP][2][SLAB];  // [wave or pixel lane][P | Q][channel of the slab]
  __shared__ float prod[2][SLAB];
  __shared__ float ab[32][2];
  const int slabs = (C + SLAB - 1) / SLAB;
  const int b = blockIdx.x / slabs, cbase = (blockIdx.x % slabs) * SLAB;
  const int chunk = threadIdx.x % NCH, pl = threadIdx.x / NCH, wave = threadIdx.x >> 6;
  const int c0_raw = cbase + chunk * 8;
  // (slabs beyond C: chunks with c0 >= C compute on channel 0 of the slab and are dropped at the end, so that every lane of the wave
  // takes part in the shuffles)
  const bool live = c0_raw < C && pl < PL;
  const int c0 = live ? c0_raw : cbase;
  const int cg = C / G;
  float mu[8], rs[8], wv[8], bv[8], sc[8], sh[8], aP[8], aQ[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float* sg = st + ((int64_t)b * G + (c0 + e) / cg) * 2;
    mu[e] = sg[0];
    rs[e] = sg[1];
    wv[e] = w[c0 + e];
    bv[e] = bb[c0 + e];
    sc[e] = fs ? bf2f(fs[(int64_t)b * ldf + c0 + e]) : 0.f;
    sh[e] = fs ? bf2f(fh[(int64_t)b * ldf + c0 + e]) : 0.f;
    aP[e] = aQ[e] = 0.f;
  }
  auto accum = [&](const u32x4_t& xr, const u32x4_t& dr) {
    float xv[8], dv[8];
    unpack8(xr, xv);
    unpack8(dr, dv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (xv[e] - mu[e]) * rs[e];
      const float h = (xh * wv[e] + bv[e]) * (1.0f + sc[e]) + sh[e];
      const float dh = dv[e] * (silu ? dsilu_f(h) : 1.0f);
      aP[e] += dh * xh;
      aQ[e] += dh;
    }
  };
  const int64_t row0 = ((int64_t)b * HW) * C + c0;  // element offset of (pixel 0, chunk) of this sample
  constexpr int NK = KEEP > 0 ? KEEP : 1;
  u32x4_t xk[NK], dk[NK], rk[NK];
  if constexpr (KEEP > 0) {
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
      const int p = pl + i * PL;
      if (p < HW && pl < PL) {
        xk[i] = *(const u32x4_t*)(x + row0 + (int64_t)p * C);
        dk[i] = *(const u32x4_t*)(dout + row0 + (int64_t)p * C);
      }
    }
    if (fz.dres) {
#pragma unroll
      for (int i = 0; i < KEEP; ++i) {
        const int p = pl + i * PL;
        if (p < HW && pl < PL) rk[i] = *(const u32x4_t*)(fz.dres + row0 + (int64_t)p * C);
      }
    }
#pragma unroll
    for (int i = 0; i < KEEP; ++i)
      if (pl + i * PL < HW && pl < PL) accum(xk[i], dk[i]);
  } else {
    int p = pl < PL ? pl : HW;  // (the idle threads of the 12-chunk form walk nothing)
    for (; p + 3 * PL < HW; p += 4 * PL) {  // four rows of this lane in flight
      u32x4_t xr[4], dr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xr[i] = *(const u32x4_t*)(x + row0 + (int64_t)(p + i * PL) * C);
        dr[i] = *(const u32x4_t*)(dout + row0 + (int64_t)(p + i * PL) * C);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) accum(xr[i], dr[i]);
    }
    for (; p < HW; p += PL) accum(*(const u32x4_t*)(x + row0 + (int64_t)p * C), *(const u32x4_t*)(dout + row0 + (int64_t)p * C));
  }
  // fold the pixel lanes of this wave (lane bits log2(NCH)..5; none when a wave is one pixel row) with shuffles; lanes 0 .. NCH-1 of
  // every wave (all 64 at NCH = 64) store the wave's sums
  if constexpr (POW2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int m = NCH; m < 64; m <<= 1) {
        aP[e] += __shfl_xor(aP[e], m);
        aQ[e] += __shfl_xor(aQ[e], m);
      }
    }
  }
  if (POW2 ? (int)(threadIdx.x & 63) < NCH : pl < PL) {
    const int row = POW2 ? wave : pl, col = chunk * 8;
    const float z = live ? 1.f : 0.f;
    *(f32x4_t*)&part[row][0][col] = f32x4_t{aP[0] * z, aP[1] * z, aP[2] * z, aP[3] * z};
    *(f32x4_t*)&part[row][0][col + 4] = f32x4_t{aP[4] * z, aP[5] * z, aP[6] * z, aP[7] * z};
    *(f32x4_t*)&part[row][1][col] = f32x4_t{aQ[0] * z, aQ[1] * z, aQ[2] * z, aQ[3] * z};
    *(f32x4_t*)&part[row][1][col + 4] = f32x4_t{aQ[4] * z, aQ[5] * z, aQ[6] * z, aQ[7] * z};
  }
  __syncthreads();
  const int nch = (C - cbase < SLAB) ? C - cbase : SLAB;  // channels of this slab (a whole number of groups)
  for (int i = threadIdx.x; i < nch; i += 256) {
    const int c = cbase + i;
    // (NCH < 64: a wave covers 64 / NCH pixel lanes of every chunk, so the four waves hold four partials of the same channel;
    //  NCH = 64: the same -- four pixel lanes, one per wave)
    float P, Q;
    if constexpr (POW2) {
      P = (part[0][0][i] + part[1][0][i]) + (part[2][0][i] + part[3][0][i]);
      Q = (part[0][1][i] + part[1][1][i]) + (part[2][1][i] + part[3][1][i]);
    } else {
      P = Q = 0.f;
#pragma unroll
      for (int r = 0; r < NP; ++r) {
        P += part[r][0][i];
        Q += part[r][1][i];
      }
    }
    const float wc = w[c], bc = bb[c];
    const float s1 = fs ? 1.0f + bf2f(fs[(int64_t)b * ldf + c]) : 1.0f;
    if (fz.dfs) {
      fz.dfs[(int64_t)b * fz.lddf + c] = f2bf(wc * P + bc * Q);  // S0 = sum dh * y
      fz.dfh[(int64_t)b * fz.lddf + c] = f2bf(Q);                // S1 = sum dh
    }
    const float s2 = s1 * P, s3 = s1 * Q;  // sum dy * xhat, sum dy
    unsafeAtomicAdd(&fz.dw[c], s2);
    unsafeAtomicAdd(&fz.db[c], s3);
    prod[0][i] = wc * s3;  // (the group sums below: A = sum_c w[c] S3[c], Bv = sum_c w[c] S2[c])
    prod[1][i] = wc * s2;
  }
  __syncthreads();
  {
    const float inv_n = 1.0f / (float)(HW * cg);
    const int g = threadIdx.x >> 3, k = threadIdx.x & 7;  // (<= 32 groups per slab: 256 threads cover them)
    float A = 0.f, Bv = 0.f;
    if (g < nch / cg)
      for (int j = k; j < cg; j += 8) {
        A += prod[0][g * cg + j];
        Bv += prod[1][g * cg + j];
      }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      A += __shfl_xor(A, m);
      Bv += __shfl_xor(Bv, m);
    }
    if (k == 0 && g < nch / cg) {
      ab[g][0] = A * inv_n;
      ab[g][1] = Bv * inv_n;
    }
  }
  __syncthreads();
  if (!live) return;
  float gA[8], gB[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    gA[e] = ab[(c0 + e - cbase) / cg][0];
    gB[e] = ab[(c0 + e - cbase) / cg][1];
  }
  auto apply = [&](const u32x4_t& xr, const u32x4_t& dr, const u32x4_t& rr) -> u32x4_t {
    float xv[8], dv[8], rv[8];
    unpack8(xr, xv);
    unpack8(dr, dv);
    if (fz.dres) unpack8(rr, rv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (xv[e] - mu[e]) * rs[e];
      const float s1 = 1.0f + sc[e];
      const float h = (xh * wv[e] + bv[e]) * s1 + sh[e];
      const float dy = dv[e] * (silu ? dsilu_f(h) : 1.0f) * s1;
      xv[e] = rs[e] * (dy * wv[e] - gA[e] - xh * gB[e]) + (fz.dres ? rv[e] : 0.f);
    }
    return pack8(xv);
  };
  if constexpr (KEEP > 0) {
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {
      const int p = pl + i * PL;
      if (p < HW) *(u32x4_t*)(fz.dx + row0 + (int64_t)p * C) = apply(xk[i], dk[i], rk[i]);
    }
  } else {
    const u32x4_t zero = {0u, 0u, 0u, 0u};
    int p = pl;  // (live threads only: pl < PL)
    for (; p + 3 * PL < HW; p += 4 * PL) {
      u32x4_t xr[4], dr[4], rr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t o = row0 + (int64_t)(p + i * PL) * C;
        xr[i] = *(const u32x4_t*)(x + o);
        dr[i] = *(const u32x4_t*)(dout + o);
        rr[i] = fz.dres ? *(const u32x4_t*)(fz.dres + o) : zero;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) *(u32x4_t*)(fz.dx + row0 + (int64_t)(p + i * PL) * C) = apply(xr[i], dr[i], rr[i]);
    }
    for (; p < HW; p += PL) {
      const int64_t o = row0 + (int64_t)p * C;
      *(u32x4_t*)(fz.dx + o) = apply(*(const u32x4_t*)(x + o), *(const u32x4_t*)(dout + o), fz.dres ? *(const u32x4_t*)(fz.dres + o) : zero);
    }
  }
}
// per sample: add the pixel-range partials; FiLM gradients dfs = S[0], dfh = S[1] (bf16); per group g:
// A = sum_{c in g} w[c] S[3][c], Bv = sum_{c in g} w[c] S[2][c]
__global__ __launch_bounds__(256) void gn_group_sums_k(const float* __restrict__ Sp, int nsplit, int B, const float* __restrict__ w,
                                                       float* __restrict__ AB, float* __restrict__ dw, float* __restrict__ db,
                                                       bf16_t* __restrict__ dfs, bf16_t* __restrict__ dfh, int64_t lddf, int C,
                                                       int G) {
  // 8 threads per group (one aligned lane octet), each walking every 8th channel of the group; the group sums meet in three
  // shuffles -- no LDS atomics (64 channels of one group hitting one LDS word made this kernel as slow as the reduction itself)
  const int cg = C / G;
  const int b = blockIdx.x, k = threadIdx.x & 7;
  for (int g = threadIdx.x >> 3; g < G; g += 32) {
    float A = 0.f, Bv = 0.f;
    for (int c = g * cg + k; c < (g + 1) * cg; c += 8) {
      float s[4] = {0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < nsplit; ++r) {
        const float* S = Sp + ((int64_t)r * B + b) * 4 * C;
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] += S[(int64_t)q * C + c];
      }
      if (dfs) {
        dfs[(int64_t)b * lddf + c] = f2bf(s[0]);
        dfh[(int64_t)b * lddf + c] = f2bf(s[1]);
      }
      unsafeAtomicAdd(&dw[c], s[2]);
      unsafeAtomicAdd(&db[c], s[3]);
      A += w[c] * s[3];
      Bv += w[c] * s[2];
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      A += __shfl_xor(A, m);
      Bv += __shfl_xor(Bv, m);
    }
    if (k == 0) {
      AB[((int64_t)b * G + g) * 2] = A;
      AB[((int64_t)b * G + g) * 2 + 1] = Bv;
    }
  }
}
// backward pass 2: dx = r * (dy*w - A/n - xhat*Bv/n) (+ dres), 8 channels per thread
__global__ void gn_bwd_apply_k(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ x, const float* __restrict__ st,
                               const float* __restrict__ w, const float* __restrict__ bb, const bf16_t* __restrict__ fs,
                               const bf16_t* __restrict__ fh, int64_t ldf, int silu, const float* __restrict__ AB,
                               const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx, int HW, int C, int G) {
  // (grid and index arithmetic as in gn_apply_fwd_k: one sample per blockIdx.y, 32-bit indices, one group lookup per run)
  const unsigned cg = C / G, C8 = C >> 3, per = (unsigned)HW * C8;
  const float inv_n = 1.0f / (float)(HW * cg);
  const int b = blockIdx.y;
  const int64_t base = (int64_t)b * per;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x) {
    const unsigned c0 = (i % C8) * 8;
    const int64_t o = (base + i) * 8;
    float xv[8], dv[8], rv[8], sc[8], sh[8];
    unpack8(*(const u32x4_t*)(x + o), xv);
    unpack8(*(const u32x4_t*)(dout + o), dv);
    if (dres) unpack8(*(const u32x4_t*)(dres + o), rv);
    if (fs) {
      unpack8(*(const u32x4_t*)(fs + (int64_t)b * ldf + c0), sc);
      unpack8(*(const u32x4_t*)(fh + (int64_t)b * ldf + c0), sh);
    }
    float wv[8], bv[8];
    *(f32x4_t*)&wv[0] = *(const f32x4_t*)(w + c0);
    *(f32x4_t*)&wv[4] = *(const f32x4_t*)(w + c0 + 4);
    *(f32x4_t*)&bv[0] = *(const f32x4_t*)(bb + c0);
    *(f32x4_t*)&bv[4] = *(const f32x4_t*)(bb + c0 + 4);
    unsigned g = c0 / cg, rem = c0 - g * cg;
    const float* sg = st + ((int64_t)b * G + g) * 2;
    const float* ag = AB + ((int64_t)b * G + g) * 2;
    float mu = sg[0], r = sg[1], A = ag[0] * inv_n, Bv = ag[1] * inv_n;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (rem == cg) {  // next group
        rem = 0;
        sg += 2;
        ag += 2;
        mu = sg[0];
        r = sg[1];
        A = ag[0] * inv_n;
        Bv = ag[1] * inv_n;
      }
      ++rem;
      const float xh = (xv[e] - mu) * r;
      const float s1 = fs ? 1.0f + sc[e] : 1.0f;
      const float h = (xh * wv[e] + bv[e]) * s1 + (fs ? sh[e] : 0.f);
      const float dy = dv[e] * (silu ? dsilu_f(h) : 1.0f) * s1;
      xv[e] = r * (dy * wv[e] - A - xh * Bv) + (dres ? rv[e] : 0.f);
    }
    *(u32x4_t*)(dx + o) = pack8(xv);
  }
}
// pixel ranges of gn_bwd_reduce_k: enough workgroups to fill the chip at small batch x channel counts, at least 64 pixels each
static inline int gn_bwd_ranges(int64_t B, int64_t HW, int64_t C) {
  const int64_t slabs = (C + 63) / 64;
  int64_t ns = 1024 / (B * slabs);
  if (ns > HW / 64) ns = HW / 64;
  if (ns > DL_GN_BWD_MAX_RANGES) ns = DL_GN_BWD_MAX_RANGES;
  return ns < 1 ? 1 : (int)ns;
}
/* scratch: f32 [DL_GN_BWD_MAX_RANGES * B*4*C + B*G*2] */
extern "C" int dl_gn_bwd(const void* dout, const void* x, const float* stats, const float* w, const float* b,
                         const void* film_scale, const void* film_shift, int64_t ld_film, int act_silu, const void* dres,
                         void* dx, float* dw, float* db, void* dfilm_scale, void* dfilm_shift, int64_t ld_dfilm, float* scratch, int64_t B,
                         int64_t HW, int64_t C, int64_t G, dl_stream_t stream) {
  DL_CHECK_ARG(dout && x && stats && w && b && dx && dw && db && scratch && B > 0 && C % G == 0 && C % 8 == 0 && G <= GN_MAXG,
               "dl_gn_bwd: bad args");
  DL_CHECK_ARG((film_scale == nullptr) == (dfilm_scale == nullptr), "dl_gn_bwd: film grads iff film inputs");
  DL_CHECK_ARG((((uintptr_t)dout | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dres | (uintptr_t)film_scale | (uintptr_t)film_shift) & 15) == 0 &&
                   ld_film % 8 == 0,
               "dl_gn_bwd: 16-byte alignment");
  DL_CHECK_ARG(B < 65536 && HW * C / 8 < (1ll << 31), "dl_gn_bwd: B < 65536, HW*C/8 < 2^31");
  float* Sp = scratch;
  float* AB = scratch + (int64_t)DL_GN_BWD_MAX_RANGES * B * 4 * C;
  int ns = 1;
  if (HW <= 64 && C >= 512 && 128 % (C / G) == 0 && 128 / (C / G) <= 32 && C % 128 == 0 && (gn_bwd_fused() & 1)) {
    // 128-channel slabs (whole groups), 16 pixel lanes: four times the workgroups of the 512-channel form below -- at B = 128 that
    // one launches 128-384 workgroups of 4-16 serial pixel iterations on a 256-CU chip (54 us for 40 MB of traffic)
    const GnFused fz{dw, db, (bf16_t*)dfilm_scale, (bf16_t*)dfilm_shift, ld_dfilm, (const bf16_t*)dres, (bf16_t*)dx};
    hipLaunchKernelGGL((gn_bwd_fused_k<16, 4>), dim3((unsigned)(B * (C / 128)), 1u), 256, 0, (hipStream_t)stream,
                       (const bf16_t*)dout, (const bf16_t*)x, stats, w, b, (const bf16_t*)film_scale, (const bf16_t*)film_shift,
                       ld_film, act_silu, (int)B, (int)HW, (int)C, (int)G, fz);
    DL_LAUNCH_CHECK();
    return DL_OK;
  }
  if (HW > 64 && HW <= 1024 && 64 % (C / G) == 0 && 64 / (C / G) <= 32 && C % 64 == 0 && (B * (C / 64) >= 128 || gn_bwd_fused() == 3) && (gn_bwd_fused() & 1)) {
    // high-resolution maps (16 x 16, 32 x 32): the same fusion on 64-channel slabs with 32 pixel lanes -- one launch instead of
    // reduce (pixel ranges) + group sums + apply; the second pass re-reads the slab's x / dout (up to 2 x 128 KB) through L2 / MALL
    const GnFused fz{dw, db, (bf16_t*)dfilm_scale, (bf16_t*)dfilm_shift, ld_dfilm, (const bf16_t*)dres, (bf16_t*)dx};
    // (rows kept in registers -- KEEP = 8 at 16 x 16 -- cost the second wave per SIMD: 42 vs 25 us; both passes stream)
    hipLaunchKernelGGL((gn_bwd_fused_k<8, 0>), dim3((unsigned)(B * (C / 64)), 1u), 256, 0, (hipStream_t)stream,
                       (const bf16_t*)dout, (const bf16_t*)x, stats, w, b, (const bf16_t*)film_scale, (const bf16_t*)film_shift,
                       ld_film, act_silu, (int)B, (int)HW, (int)C, (int)G, fz);
    DL_LAUNCH_CHECK();
    return DL_OK;
  }
  if (C % 96 == 0 && 96 % (C / G) == 0 && 96 / (C / G) <= 32 && HW <= 1024 && (B * (C / 96) >= 128 || gn_bwd_fused() == 3) &&
      (gn_bwd_fused() & 1)) {
    // 12 / 24 / 48 channels per group (C = 384 / 768 / 1536, the concatenated inputs of the output blocks): 96-channel slabs
    const GnFused fz{dw, db, (bf16_t*)dfilm_scale, (bf16_t*)dfilm_shift, ld_dfilm, (const bf16_t*)dres, (bf16_t*)dx};
    if (HW <= 84)
      hipLaunchKernelGGL((gn_bwd_fused_k<12, 4>), dim3((unsigned)(B * (C / 96)), 1u), 256, 0, (hipStream_t)stream,
                         (const bf16_t*)dout, (const bf16_t*)x, stats, w, b, (const bf16_t*)film_scale, (const bf16_t*)film_shift,
                         ld_film, act_silu, (int)B, (int)HW, (int)C, (int)G, fz);
    else
      hipLaunchKernelGGL((gn_bwd_fused_k<12, 0>), dim3((unsigned)(B * (C / 96)), 1u), 256, 0, (hipStream_t)stream,
                         (const bf16_t*)dout, (const bf16_t*)x, stats, w, b, (const bf16_t*)film_scale, (const bf16_t*)film_shift,
                         ld_film, act_silu, (int)B, (int)HW, (int)C, (int)G, fz);
    DL_LAUNCH_CHECK();
    return DL_OK;
  }
  if (HW <= 64 && C >= 512 && 512 % (C / G) == 0 && 512 / (C / G) <= 32 && gn_bwd_fused()) {
    const GnFused fz{dw, db, (bf16_t*)dfilm_scale, (bf16_t*)dfilm_shift, ld_dfilm, (const bf16_t*)dres, (bf16_t*)dx};
    hipLaunchKernelGGL((gn_bwd_fused_k<64, 0>), dim3((unsigned)(B * ((C + 511) / 512)), 1u), 256, 0, (hipStream_t)stream,
                       (const bf16_t*)dout, (const bf16_t*)x, stats, w, b, (const bf16_t*)film_scale, (const bf16_t*)film_shift,
                       ld_film, act_silu, (int)B, (int)HW, (int)C, (int)G, fz);
    DL_LAUNCH_CHECK();
    return DL_OK;
  }
  if (HW <= 64 && C >= 512) {
    hipLaunchKernelGGL((gn_bwd_reduce_k<64>), dim3((unsigned)(B * ((C + 511) / 512)), 1u), 256, 0, (hipStream_t)stream,
                       (const bf16_t*)dout, (const bf16_t*)x, stats, w, b, (const bf16_t*)film_scale, (const bf16_t*)film_shift,
                       ld_film, act_silu, Sp, (int)B, (int)HW, (int)C, (int)G);
  } else {
    ns = gn_bwd_ranges(B, HW, C);
    hipLaunchKernelGGL((gn_bwd_reduce_k<8>), dim3((unsigned)(B * ((C + 63) / 64)), (unsigned)ns), 256, 0, (hipStream_t)stream,
                       (const bf16_t*)dout, (const bf16_t*)x, stats, w, b, (const bf16_t*)film_scale, (const bf16_t*)film_shift,
                       ld_film, act_silu, Sp, (int)B, (int)HW, (int)C, (int)G);
  }
  hipLaunchKernelGGL(gn_group_sums_k, (int)B, 256, 0, (hipStream_t)stream, Sp, ns, (int)B, w, AB, dw, db, (bf16_t*)dfilm_scale,
                     (bf16_t*)dfilm_shift, ld_dfilm, (int)C, (int)G);
  hipLaunchKernelGGL(gn_bwd_apply_k, gn_grid(B, HW * C / 8), 256, 0, (hipStream_t)stream, (const bf16_t*)dout, (const bf16_t*)x, stats,
                     w, b, (const bf16_t*)film_scale, (const bf16_t*)film_shift, ld_film, act_silu, AB, (const bf16_t*)dres,
                     (bf16_t*)dx, (int)HW, (int)C, (int)G);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ---------------------------------------------------------------- im2col for 3x3 / pad 1: cols[row, (ky*3+kx)*C + c]
// one thread per (row, tap, 8-channel chunk); rows >= B*H*W and columns >= 9*C are zero-filled (GEMM padding)
__global__ void im2col3x3_k(const bf16_t* __restrict__ x, int64_t ldx, bf16_t* __restrict__ cols, int B, int H, int W, int C,
                            int64_t rows, int64_t ld) {
  const int ld8 = (int)(ld >> 3), C8 = C >> 3;
  const int64_t total = rows * ld8;
  const int64_t M = (int64_t)B * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k8 = (int)(i % ld8);
    const int64_t row = i / ld8;
    u32x4_t v = {0, 0, 0, 0};
    if (row < M && k8 < 9 * C8) {
      const int tap = k8 / C8, c8 = k8 - tap * C8;
      const int ky = tap / 3 - 1, kx = tap % 3 - 1;
      const int xw = (int)(row % W), yh = (int)((row / W) % H);
      const int yy = yh + ky, xx = xw + kx;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = *(const u32x4_t*)(x + (row + (int64_t)ky * W + kx) * ldx + c8 * 8);
    }
    *(u32x4_t*)(cols + row * ld + (int64_t)k8 * 8) = v;
  }
}
// generic (C not a multiple of 8, e.g. the 1-channel stem): scalar version
__global__ void im2col3x3_scalar_k(const bf16_t* __restrict__ x, int64_t ldx, bf16_t* __restrict__ cols, int B, int H, int W,
                                   int C, int64_t rows, int64_t ld) {
  const int64_t total = rows * ld;
  const int64_t M = (int64_t)B * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % ld);
    const int64_t row = i / ld;
    bf16_t v = 0;
    if (row < M && k < 9 * C) {
      const int tap = k / C, c = k - tap * C;
      const int ky = tap / 3 - 1, kx = tap % 3 - 1;
      const int xw = (int)(row % W), yh = (int)((row / W) % H);
      if (yh + ky >= 0 && yh + ky < H && xw + kx >= 0 && xw + kx < W) v = x[(row + (int64_t)ky * W + kx) * ldx + c];
    }
    cols[i] = v;
  }
}
extern "C" int dl_im2col3x3(const void* x, int64_t ldx, void* cols, int64_t B, int64_t H, int64_t W, int64_t C, int64_t rows,
                            int64_t ld, dl_stream_t stream) {
  DL_CHECK_ARG(x && cols && B > 0 && H > 0 && W > 0 && C > 0 && ldx >= C && rows >= B * H * W && ld >= 9 * C && ld % 8 == 0,
               "dl_im2col3x3: bad args");
  if (C % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0)
    hipLaunchKernelGGL(im2col3x3_k, grid_for(rows * ld / 8), 256, 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)cols,
                       (int)B, (int)H, (int)W, (int)C, rows, ld);
  else
    hipLaunchKernelGGL(im2col3x3_scalar_k, grid_for(rows * ld), 256, 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)cols,
                       (int)B, (int)H, (int)W, (int)C, rows, ld);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// conv weight [Co, Ci, 3, 3] f32 -> forward shadow [Co, ldf] (k = tap*Ci + ci) and dgrad shadow [Ci, ldd]
// (k = tap'*Co + co with the kernel rotated by 180 degrees: tap' = 8 - tap); padding columns zeroed
__global__ void cast_conv3x3_k(const float* __restrict__ w, int Co, int Ci, bf16_t* __restrict__ wf, int64_t ldf,
                               bf16_t* __restrict__ wd, int64_t ldd) {
  const int64_t nf = (int64_t)Co * ldf, nd = (int64_t)Ci * ldd;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nd; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nf) {
      const int k = (int)(i % ldf), co = (int)(i / ldf);
      float v = 0.f;
      if (k < 9 * Ci) v = w[((int64_t)co * Ci + (k % Ci)) * 9 + k / Ci];
      wf[i] = f2bf(v);
    } else {
      const int64_t j = i - nf;
      const int k = (int)(j % ldd), ci = (int)(j / ldd);
      float v = 0.f;
      if (k < 9 * Co) v = w[((int64_t)(k % Co) * Ci + ci) * 9 + (8 - k / Co)];
      wd[j] = f2bf(v);
    }
  }
}
// The same two shadows through an LDS tile of 32 output x 32 input channels x 9 taps (whole channel tiles, no padding columns): the
// weight is read in 1152-byte runs and both shadows are written in 64-byte runs -- the direct form above reads the f32 weight with a
// stride of 9 (forward shadow) / 9 Ci (data-gradient shadow) floats per lane and cost the UNet step 1.7 ms for 2.2 GB of traffic.
#define CW_T 32
#define CW_P (CW_T * 9 + 1)
__global__ __launch_bounds__(256) void cast_conv3x3_tiled_k(const float* __restrict__ w, int Co, int Ci, bf16_t* __restrict__ wf,
                                                            int64_t ldf, bf16_t* __restrict__ wd, int64_t ldd) {
  __shared__ float tile[CW_T * CW_P];
  const int co0 = blockIdx.y * CW_T, ci0 = blockIdx.x * CW_T;
  for (int i = threadIdx.x; i < CW_T * CW_T * 9; i += 256) {
    const int col = i / (CW_T * 9), j = i - col * (CW_T * 9);
    tile[col * CW_P + j] = w[((int64_t)(co0 + col) * Ci + ci0) * 9 + j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CW_T * CW_T * 9; i += 256) {
    const int l = i & (CW_T - 1), rest = i / CW_T, tap = rest % 9, o = rest / 9;
    wf[(int64_t)(co0 + o) * ldf + tap * Ci + ci0 + l] = f2bf(tile[o * CW_P + l * 9 + tap]);          // l = input channel
    wd[(int64_t)(ci0 + o) * ldd + (8 - tap) * Co + co0 + l] = f2bf(tile[l * CW_P + o * 9 + tap]);    // l = output channel
  }
}
extern "C" int dl_cast_conv3x3_weight(const float* w, int64_t Co, int64_t Ci, void* wf, int64_t ldf, void* wd, int64_t ldd,
                                      dl_stream_t stream) {
  DL_CHECK_ARG(w && wf && wd && Co > 0 && Ci > 0 && ldf >= 9 * Ci && ldd >= 9 * Co, "dl_cast_conv3x3_weight: bad args");
  if (Co % CW_T == 0 && Ci % CW_T == 0 && ldf == 9 * Ci && ldd == 9 * Co && Co / CW_T <= 65535) {
    hipLaunchKernelGGL(cast_conv3x3_tiled_k, dim3((unsigned)(Ci / CW_T), (unsigned)(Co / CW_T)), 256, 0, (hipStream_t)stream, w, (int)Co,
                       (int)Ci, (bf16_t*)wf, ldf, (bf16_t*)wd, ldd);
    DL_LAUNCH_CHECK();
    return DL_OK;
  }
  hipLaunchKernelGGL(cast_conv3x3_k, grid_for(Co * ldf + Ci * ldd), 256, 0, (hipStream_t)stream, w, (int)Co, (int)Ci, (bf16_t*)wf,
                     ldf, (bf16_t*)wd, ldd);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
// every 3x3 convolution weight of a network in ONE launch (the UNet has 56 of them: 56 launches of 3-130 us per step otherwise, each
// with its launch gap on the main stream): a device table of descriptors, one 32 x 32 channel tile (x 9 taps) per workgroup, the
// tile arithmetic of cast_conv3x3_tiled_k
__global__ __launch_bounds__(256) void cast_conv3x3_batched_k(const dl_cast_conv_desc_t* __restrict__ desc, int n_desc) {
  __shared__ float tile[CW_T * CW_P];
  int lo = 0, hi = n_desc - 1;  // last descriptor whose tile_begin <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid].tile_begin <= (int64_t)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const dl_cast_conv_desc_t d = desc[lo];
  const int64_t t = (int64_t)blockIdx.x - d.tile_begin;
  const int Ci = (int)d.Ci, Co = (int)d.Co, tci = Ci / CW_T;
  const int co0 = (int)(t / tci) * CW_T, ci0 = (int)(t % tci) * CW_T;
  const float* w = (const float*)d.w;
  bf16_t* wf = (bf16_t*)d.wf;
  bf16_t* wd = (bf16_t*)d.wd;
  for (int i = threadIdx.x; i < CW_T * CW_T * 9; i += 256) {
    const int col = i / (CW_T * 9), j = i - col * (CW_T * 9);
    tile[col * CW_P + j] = w[((int64_t)(co0 + col) * Ci + ci0) * 9 + j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CW_T * CW_T * 9; i += 256) {
    const int l = i & (CW_T - 1), rest = i / CW_T, tap = rest % 9, o = rest / 9;
    if (wf) wf[(int64_t)(co0 + o) * d.ldf + tap * Ci + ci0 + l] = f2bf(tile[o * CW_P + l * 9 + tap]);
    if (wd) wd[(int64_t)(ci0 + o) * d.ldd + (8 - tap) * Co + co0 + l] = f2bf(tile[l * CW_P + o * 9 + tap]);
  }
}
extern "C" int dl_cast_conv3x3_weights_batched(const dl_cast_conv_desc_t* desc_dev, int n_desc, int64_t total_tiles, dl_stream_t stream) {
  DL_CHECK_ARG(desc_dev && n_desc > 0 && total_tiles > 0 && total_tiles < (1ll << 31), "dl_cast_conv3x3_weights_batched: bad args");
  hipLaunchKernelGGL(cast_conv3x3_batched_k, (int)total_tiles, 256, 0, (hipStream_t)stream, desc_dev, n_desc);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
// wgrad lands TRANSPOSED as [(tap, ci), Co] f32 (cols^T dY: its row count 9*Ci is a multiple of the 384-row tile of the big TN
// GEMM for every 128-multiple channel count); fold it into the reference layout [Co, Ci, 3, 3] (+=)
__global__ void conv3x3_wgrad_fold_k(const float* __restrict__ g, int64_t ldg, float* __restrict__ dw, int Co, int Ci) {
  const int64_t n = (int64_t)Co * Ci * 9;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % 9);
    const int64_t r = i / 9;
    const int ci = (int)(r % Ci), co = (int)(r / Ci);
    dw[i] += g[(int64_t)(tap * Ci + ci) * ldg + co];
  }
}
// the same fold through the LDS tile of cast_conv3x3_tiled_k: g is read in 128-byte runs along co, dw updated in 1152-byte runs
__global__ __launch_bounds__(256) void conv3x3_wgrad_fold_tiled_k(const float* __restrict__ g, int64_t ldg, float* __restrict__ dw, int Co,
                                                                  int Ci) {
  __shared__ float tile[CW_T * CW_P];
  const int co0 = blockIdx.y * CW_T, ci0 = blockIdx.x * CW_T;
  for (int i = threadIdx.x; i < CW_T * CW_T * 9; i += 256) {
    const int col = i & (CW_T - 1), rest = i / CW_T, cil = rest % CW_T, tap = rest / CW_T;
    tile[col * CW_P + cil * 9 + tap] = g[(int64_t)(tap * Ci + ci0 + cil) * ldg + co0 + col];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CW_T * CW_T * 9; i += 256) {
    const int col = i / (CW_T * 9), j = i - col * (CW_T * 9);
    dw[((int64_t)(co0 + col) * Ci + ci0) * 9 + j] += tile[col * CW_P + j];
  }
}
// every convolution's staged weight gradient folded by ONE launch at the end of the backward (the UNet has 56: 56 launches of
// 20-80 us on the side stream otherwise, 1.6 ms of the step): a device table of descriptors, one 32 x 32 channel tile (x 9 taps) per
// workgroup; clear != 0 writes zeros back into the staging tile, so the next backward accumulates into a clean stage without a memset
__global__ __launch_bounds__(256) void conv3x3_wgrad_fold_batched_k(const dl_fold_conv_desc_t* __restrict__ desc, int n_desc, int clear) {
  __shared__ float tile[CW_T * CW_P];
  int lo = 0, hi = n_desc - 1;  // last descriptor whose tile_begin <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid].tile_begin <= (int64_t)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const dl_fold_conv_desc_t d = desc[lo];
  int64_t t = (int64_t)blockIdx.x - d.tile_begin;
  const int Ci = (int)d.Ci, tci = Ci / CW_T;
  float* g = (float*)d.g;
  float* dw = (float*)d.dw;
  if (d.n_img >= DL_FOLD_TAP_SPLIT_MIN_IMAGES) {
    // many partial images (the high-resolution layers: 16 - 32 channel tiles, 64 - 128 images of the whole chip's accumulators): one
    // TAP of a channel tile per workgroup -- nine times the workgroups, 16 bytes per lane and image, the images still added in
    // image order by one lane per element (bit-reproducible)
    const int tap = (int)(t % 9);
    t /= 9;
    const int co0 = (int)(t / tci) * CW_T, ci0 = (int)(t % tci) * CW_T;
    const int cil = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
    const float* src = g + (int64_t)(tap * Ci + ci0 + cil) * d.ldg + co0 + c4;
    f32x4_t v = *(const f32x4_t*)src;
    int s = 1;
    for (; s + 8 <= (int)d.n_img; s += 8) {
      f32x4_t u[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) u[e] = *(const f32x4_t*)(src + (int64_t)(s + e) * d.img_stride);
#pragma unroll
      for (int e = 0; e < 8; ++e) v += u[e];
    }
    for (; s < (int)d.n_img; ++s) v += *(const f32x4_t*)(src + (int64_t)s * d.img_stride);
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[(c4 + e) * (CW_T + 1) + cil] = v[e];
    __syncthreads();
    for (int i = threadIdx.x; i < CW_T * CW_T; i += 256) {
      const int col = i >> 5, ci = i & (CW_T - 1);
      dw[((int64_t)(co0 + col) * Ci + ci0 + ci) * 9 + tap] += tile[col * (CW_T + 1) + ci];
    }
    return;
  }
  const int co0 = (int)(t / tci) * CW_T, ci0 = (int)(t % tci) * CW_T;
  const int n_img = d.n_img > 1 ? (int)d.n_img : 1;
  for (int i = threadIdx.x; i < CW_T * CW_T * 9; i += 256) {
    const int col = i & (CW_T - 1), rest = i / CW_T, cil = rest % CW_T, tap = rest / CW_T;
    float* src = g + (int64_t)(tap * Ci + ci0 + cil) * d.ldg + co0 + col;
    if (d.n_img >= 1) {  // partial images of dl_conv3x3_wgrad_tn_parts: added in image order, nothing to clear
      float v = *src;
      for (int s = 1; s < n_img; ++s) v += src[(int64_t)s * d.img_stride];
      tile[col * CW_P + cil * 9 + tap] = v;
    } else {
      tile[col * CW_P + cil * 9 + tap] = *src;
      if (clear) *src = 0.f;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < CW_T * CW_T * 9; i += 256) {
    const int col = i / (CW_T * 9), j = i - col * (CW_T * 9);
    dw[((int64_t)(co0 + col) * Ci + ci0) * 9 + j] += tile[col * CW_P + j];
  }
}
extern "C" int dl_conv3x3_wgrad_fold_batched(const dl_fold_conv_desc_t* desc_dev, int n_desc, int64_t total_tiles, int clear_stage,
                                             dl_stream_t stream) {
  DL_CHECK_ARG(desc_dev && n_desc > 0 && total_tiles > 0 && total_tiles < (1ll << 31), "dl_conv3x3_wgrad_fold_batched: bad args");
  hipLaunchKernelGGL(conv3x3_wgrad_fold_batched_k, (int)total_tiles, 256, 0, (hipStream_t)stream, desc_dev, n_desc, clear_stage);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_conv3x3_wgrad_fold(const float* g, int64_t ldg, float* dw, int64_t Co, int64_t Ci, dl_stream_t stream) {
  DL_CHECK_ARG(g && dw && Co > 0 && Ci > 0 && ldg >= Co, "dl_conv3x3_wgrad_fold: bad args");
  if (Co % CW_T == 0 && Ci % CW_T == 0 && Co / CW_T <= 65535) {
    hipLaunchKernelGGL(conv3x3_wgrad_fold_tiled_k, dim3((unsigned)(Ci / CW_T), (unsigned)(Co / CW_T)), 256, 0, (hipStream_t)stream, g, ldg, dw,
                       (int)Co, (int)Ci);
    DL_LAUNCH_CHECK();
    return DL_OK;
  }
  hipLaunchKernelGGL(conv3x3_wgrad_fold_k, grid_for(Co * Ci * 9), 256, 0, (hipStream_t)stream, g, ldg, dw, (int)Co, (int)Ci);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ---------------------------------------------------------------- 2x2 reduce / expand (avg-pool & nearest-upsample, fwd & bwd)
__global__ void reduce2x2_k(const bf16_t* __restrict__ x, bf16_t* __restrict__ o, int B, int Ho, int Wo, int C, float scale) {
  const int64_t n = (int64_t)B * Ho * Wo * C;
  const int W = 2 * Wo;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t r = i / C;
    const int xo = (int)(r % Wo), yo = (int)((r / Wo) % Ho), b = (int)(r / ((int64_t)Wo * Ho));
    const int64_t base = (((int64_t)b * 2 * Ho + 2 * yo) * W + 2 * xo) * C + c;
    o[i] = f2bf(scale * (bf2f(x[base]) + bf2f(x[base + C]) + bf2f(x[base + (int64_t)W * C]) + bf2f(x[base + (int64_t)W * C + C])));
  }
}
__global__ void expand2x2_k(const bf16_t* __restrict__ x, bf16_t* __restrict__ o, int B, int Hi, int Wi, int C, float scale) {
  const int64_t n = (int64_t)B * 4 * Hi * Wi * C;
  const int W = 2 * Wi, H = 2 * Hi;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t r = i / C;
    const int xx = (int)(r % W), yy = (int)((r / W) % H), b = (int)(r / ((int64_t)W * H));
    o[i] = f2bf(scale * bf2f(x[(((int64_t)b * Hi + yy / 2) * Wi + xx / 2) * C + c]));
  }
}
extern "C" int dl_reduce2x2(const void* x, void* out, int64_t B, int64_t Ho, int64_t Wo, int64_t C, float scale, dl_stream_t stream) {
  DL_CHECK_ARG(x && out && B > 0 && Ho > 0 && Wo > 0 && C > 0, "dl_reduce2x2: bad args");
  hipLaunchKernelGGL(reduce2x2_k, grid_for(B * Ho * Wo * C), 256, 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, (int)B,
                     (int)Ho, (int)Wo, (int)C, scale);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_expand2x2(const void* x, void* out, int64_t B, int64_t Hi, int64_t Wi, int64_t C, float scale, dl_stream_t stream) {
  DL_CHECK_ARG(x && out && B > 0 && Hi > 0 && Wi > 0 && C > 0, "dl_expand2x2: bad args");
  hipLaunchKernelGGL(expand2x2_k, grid_for(B * 4 * Hi * Wi * C), 256, 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, (int)B,
                     (int)Hi, (int)Wi, (int)C, scale);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
// 16 bytes (8 channels) per lane; blockIdx.y picks one of up to two (x, out) pairs of the same geometry (a resampling ResBlock moves
// its h and its x through the same 2 x 2 window: one launch instead of two)
struct Resample2 {
  const bf16_t* x[2];
  bf16_t* o[2];
};
template <bool EXPAND>
__global__ void resample2x2_vec_k(Resample2 a, int B, int Hs, int Ws, int C8, float scale) {
  // Hs x Ws = the SMALL map (reduce: output; expand: input); the large map is 2Hs x 2Ws
  const bf16_t* x = a.x[blockIdx.y];
  bf16_t* o = a.o[blockIdx.y];
  const int64_t n = (int64_t)B * Hs * Ws * C8 * (EXPAND ? 4 : 1);
  const int W = 2 * Ws, H = 2 * Hs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8);
    const int64_t r = i / C8;
    if (EXPAND) {
      const int xx = (int)(r % W), yy = (int)((r / W) % H), b = (int)(r / ((int64_t)W * H));
      float v[8];
      unpack8(*(const u32x4_t*)(x + ((((int64_t)b * Hs + yy / 2) * Ws + xx / 2) * C8 + c) * 8), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= scale;
      *(u32x4_t*)(o + i * 8) = pack8(v);
    } else {
      const int xo = (int)(r % Ws), yo = (int)((r / Ws) % Hs), b = (int)(r / ((int64_t)Ws * Hs));
      const int64_t base = ((((int64_t)b * H + 2 * yo) * W + 2 * xo) * C8 + c) * 8;
      float p0[8], p1[8], p2[8], p3[8];
      unpack8(*(const u32x4_t*)(x + base), p0);
      unpack8(*(const u32x4_t*)(x + base + (int64_t)C8 * 8), p1);
      unpack8(*(const u32x4_t*)(x + base + (int64_t)W * C8 * 8), p2);
      unpack8(*(const u32x4_t*)(x + base + (int64_t)(W + 1) * C8 * 8), p3);
#pragma unroll
      for (int e = 0; e < 8; ++e) p0[e] = scale * (p0[e] + p1[e] + p2[e] + p3[e]);
      *(u32x4_t*)(o + i * 8) = pack8(p0);
    }
  }
}
/* dl_reduce2x2 / dl_expand2x2 on two tensors of the same geometry in one launch (x1 / out1 may be NULL: one tensor) */
extern "C" int dl_resample2x2_pair(const void* x0, void* out0, const void* x1, void* out1, int64_t B, int64_t Hs, int64_t Ws, int64_t C,
                                   float scale, int expand, dl_stream_t stream) {
  DL_CHECK_ARG(x0 && out0 && (x1 == nullptr) == (out1 == nullptr) && B > 0 && Hs > 0 && Ws > 0 && C > 0, "dl_resample2x2_pair: bad args");
  const bool vec = C % 8 == 0 && (((uintptr_t)x0 | (uintptr_t)out0 | (uintptr_t)x1 | (uintptr_t)out1) & 15) == 0;
  if (!vec) {
    for (int k = 0; k < (x1 ? 2 : 1); ++k) {
      const int rc = expand ? dl_expand2x2(k ? x1 : x0, k ? out1 : out0, B, Hs, Ws, C, scale, stream)
                            : dl_reduce2x2(k ? x1 : x0, k ? out1 : out0, B, Hs, Ws, C, scale, stream);
      if (rc != DL_OK) return rc;
    }
    return DL_OK;
  }
  const Resample2 a{{(const bf16_t*)x0, (const bf16_t*)x1}, {(bf16_t*)out0, (bf16_t*)out1}};
  const int64_t n = B * Hs * Ws * (C / 8) * (expand ? 4 : 1);
  dim3 grid((unsigned)grid_for(n), x1 ? 2u : 1u);
  if (expand)
    hipLaunchKernelGGL(resample2x2_vec_k<true>, grid, 256, 0, (hipStream_t)stream, a, (int)B, (int)Hs, (int)Ws, (int)(C / 8), scale);
  else
    hipLaunchKernelGGL(resample2x2_vec_k<false>, grid, 256, 0, (hipStream_t)stream, a, (int)B, (int)Hs, (int)Ws, (int)(C / 8), scale);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ---------------------------------------------------------------- stride-2 pick / zero-stuff (Downsample's 3x3 stride-2 conv, nn.py:79)
// A stride-2 pad-1 3x3 convolution is the stride-1 one sampled at the even positions: forward = conv3x3 at full resolution + pick,
// backward = zero-stuff dY to full resolution + the stride-1 data / weight gradients.  8 channels (16 B) per thread.
__global__ void pick2x2_k(const bf16_t* __restrict__ x, bf16_t* __restrict__ o, int B, int Ho, int Wo, int C8) {
  const int64_t n = (int64_t)B * Ho * Wo * C8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8);
    const int64_t r = i / C8;
    const int xo = (int)(r % Wo), yo = (int)((r / Wo) % Ho), b = (int)(r / ((int64_t)Wo * Ho));
    ((u32x4_t*)o)[i] = ((const u32x4_t*)x)[(((int64_t)b * 2 * Ho + 2 * yo) * 2 * Wo + 2 * xo) * C8 + c];
  }
}
__global__ void stuff2x2_k(const bf16_t* __restrict__ dy, bf16_t* __restrict__ o, int B, int Hi, int Wi, int C8) {
  const int64_t n = (int64_t)B * 4 * Hi * Wi * C8;
  const int W = 2 * Wi, H = 2 * Hi;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8);
    const int64_t r = i / C8;
    const int xx = (int)(r % W), yy = (int)((r / W) % H), b = (int)(r / ((int64_t)W * H));
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (!((xx | yy) & 1)) v = ((const u32x4_t*)dy)[(((int64_t)b * Hi + yy / 2) * Wi + xx / 2) * C8 + c];
    ((u32x4_t*)o)[i] = v;
  }
}
extern "C" int dl_pick2x2(const void* x, void* out, int64_t B, int64_t Ho, int64_t Wo, int64_t C, dl_stream_t stream) {
  DL_CHECK_ARG(x && out && B > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0, "dl_pick2x2: bad args (C % 8)");
  hipLaunchKernelGGL(pick2x2_k, grid_for(B * Ho * Wo * C / 8), 256, 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)out, (int)B,
                     (int)Ho, (int)Wo, (int)(C / 8));
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_stuff2x2(const void* dy, void* out, int64_t B, int64_t Hi, int64_t Wi, int64_t C, dl_stream_t stream) {
  DL_CHECK_ARG(dy && out && B > 0 && Hi > 0 && Wi > 0 && C > 0 && C % 8 == 0, "dl_stuff2x2: bad args (C % 8)");
  hipLaunchKernelGGL(stuff2x2_k, grid_for(B * 4 * Hi * Wi * C / 8), 256, 0, (hipStream_t)stream, (const bf16_t*)dy, (bf16_t*)out,
                     (int)B, (int)Hi, (int)Wi, (int)(C / 8));
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ---------------------------------------------------------------- additive conditioning (ResBlock, use_scale_shift_norm=False)
// forward  out[b, p, c] = bf16(x[b, p, c] + e[b, c])          (unet.py:236 `h + emb_out`, both bf16 under autocast)
// backward de[b, c]     = bf16(sum_p dy[b, p, c])             one workgroup per (sample, 64-channel slab): 32 pixel lanes x 8 chunks
__global__ void rowbias_add_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ e, int64_t lde, bf16_t* __restrict__ o, int B,
                              int HW, int C8) {
  const int64_t n = (int64_t)B * HW * C8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8);
    const int b = (int)(i / ((int64_t)C8 * HW));
    float v[8], a[8];
    unpack8(((const u32x4_t*)x)[i], v);
    unpack8(*(const u32x4_t*)(e + (int64_t)b * lde + c * 8), a);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += a[k];
    ((u32x4_t*)o)[i] = pack8(v);
  }
}
__global__ __launch_bounds__(256) void rowbias_bwd_k(const bf16_t* __restrict__ dy, bf16_t* __restrict__ de, int64_t lde, int HW, int C) {
  __shared__ float red[32][65];
  const int b = blockIdx.y, c0 = blockIdx.x * 64;
  const int chunk = threadIdx.x & 7, lane = threadIdx.x >> 3;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool live = c0 + chunk * 8 < C;
  if (live)
    for (int p = lane; p < HW; p += 32) {
      float v[8];
      unpack8(*(const u32x4_t*)(dy + ((int64_t)b * HW + p) * C + c0 + chunk * 8), v);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[lane][chunk * 8 + k] = acc[k];
  __syncthreads();
  if (threadIdx.x < 64 && c0 + threadIdx.x < C) {
    float s = 0.f;
    for (int l = 0; l < 32; ++l) s += red[l][threadIdx.x];
    de[(int64_t)b * lde + c0 + threadIdx.x] = f2bf(s);
  }
}
extern "C" int dl_rowbias_add(const void* x, const void* e, int64_t lde, void* out, int64_t B, int64_t HW, int64_t C, dl_stream_t stream) {
  DL_CHECK_ARG(x && e && out && B > 0 && HW > 0 && C > 0 && C % 8 == 0 && lde % 8 == 0, "dl_rowbias_add: bad args (C, lde % 8)");
  hipLaunchKernelGGL(rowbias_add_k, grid_for(B * HW * C / 8), 256, 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)e, lde,
                     (bf16_t*)out, (int)B, (int)HW, (int)(C / 8));
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_rowbias_bwd(const void* dy, void* de, int64_t lde, int64_t B, int64_t HW, int64_t C, dl_stream_t stream) {
  DL_CHECK_ARG(dy && de && B > 0 && B < 65536 && HW > 0 && C > 0 && C % 8 == 0, "dl_rowbias_bwd: bad args (C % 8)");
  hipLaunchKernelGGL(rowbias_bwd_k, dim3((unsigned)((C + 63) / 64), (unsigned)B), 256, 0, (hipStream_t)stream, (const bf16_t*)dy,
                     (bf16_t*)de, lde, (int)HW, (int)C);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ---------------------------------------------------------------- small attention (n <= 64 tokens), one workgroup per (b, head)
// q,k,v,out: token rows [B*n, ld] with head h at columns [h*dh, (h+1)*dh); probs f32 [B,H,n,n] saved for the backward
#define AS_MAXN 64
__global__ __launch_bounds__(256) void attn_small_fwd_k(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                        const bf16_t* __restrict__ v, int64_t ldq, int64_t ldkv,
                                                        bf16_t* __restrict__ out, int64_t ldo, float* __restrict__ probs, int n,
                                                        int H, int dh, float scale) {
  __shared__ float P[AS_MAXN][AS_MAXN + 1];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const bf16_t* qb = q + (int64_t)b * n * ldq + h * dh;
  const bf16_t* kb = k + (int64_t)b * n * ldkv + h * dh;
  const bf16_t* vb = v + (int64_t)b * n * ldkv + h * dh;
  for (int idx = threadIdx.x; idx < n * n; idx += 256) {
    const int i = idx / n, j = idx % n;
    float s = 0.f;
    for (int d = 0; d < dh; d += 8) {
      float a[8], c[8];
      unpack8(*(const u32x4_t*)(qb + (int64_t)i * ldq + d), a);
      unpack8(*(const u32x4_t*)(kb + (int64_t)j * ldkv + d), c);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += a[e] * c[e];
    }
    P[i][j] = s * scale;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {  // row softmax (n <= 64 rows: one thread per row)
    float m = -INFINITY;
    for (int j = 0; j < n; ++j) m = fmaxf(m, P[i][j]);
    float l = 0.f;
    for (int j = 0; j < n; ++j) {
      const float e = __expf(P[i][j] - m);
      P[i][j] = e;
      l += e;
    }
    const float inv = 1.0f / l;
    for (int j = 0; j < n; ++j) {
      P[i][j] *= inv;
      probs[(((int64_t)b * H + h) * n + i) * n + j] = P[i][j];
    }
  }
  __syncthreads();
  const int d8 = dh >> 3;
  for (int idx = threadIdx.x; idx < n * d8; idx += 256) {
    const int i = idx / d8, dc = (idx % d8) * 8;
    float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < n; ++j) {
      float c[8];
      unpack8(*(const u32x4_t*)(vb + (int64_t)j * ldkv + dc), c);
      const float p = P[i][j];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += p * c[e];
    }
    *(u32x4_t*)(out + ((int64_t)b * n + i) * ldo + h * dh + dc) = pack8(o);
  }
}
__global__ __launch_bounds__(256) void attn_small_bwd_k(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                        const bf16_t* __restrict__ v, int64_t ldq, int64_t ldkv,
                                                        const bf16_t* __restrict__ dout, int64_t ldo,
                                                        const float* __restrict__ probs, bf16_t* __restrict__ dq,
                                                        bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int n, int H, int dh,
                                                        float scale) {
  __shared__ float P[AS_MAXN][AS_MAXN + 1];
  __shared__ float dS[AS_MAXN][AS_MAXN + 1];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const bf16_t* qb = q + (int64_t)b * n * ldq + h * dh;
  const bf16_t* kb = k + (int64_t)b * n * ldkv + h * dh;
  const bf16_t* vb = v + (int64_t)b * n * ldkv + h * dh;
  const bf16_t* dob = dout + (int64_t)b * n * ldo + h * dh;
  for (int idx = threadIdx.x; idx < n * n; idx += 256) {
    const int i = idx / n, j = idx % n;
    P[i][j] = probs[(((int64_t)b * H + h) * n + i) * n + j];
    float s = 0.f;  // dP[i][j] = dO[i,:] . v[j,:]
    for (int d = 0; d < dh; d += 8) {
      float a[8], c[8];
      unpack8(*(const u32x4_t*)(dob + (int64_t)i * ldo + d), a);
      unpack8(*(const u32x4_t*)(vb + (int64_t)j * ldkv + d), c);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += a[e] * c[e];
    }
    dS[i][j] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    float dot = 0.f;
    for (int j = 0; j < n; ++j) dot += P[i][j] * dS[i][j];
    for (int j = 0; j < n; ++j) dS[i][j] = P[i][j] * (dS[i][j] - dot) * scale;
  }
  __syncthreads();
  const int d8 = dh >> 3;
  for (int idx = threadIdx.x; idx < n * d8; idx += 256) {
    const int r = idx / d8, dc = (idx % d8) * 8;
    float gq[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < n; ++j) {
      float kk[8], qq[8], dd[8];
      unpack8(*(const u32x4_t*)(kb + (int64_t)j * ldkv + dc), kk);
      unpack8(*(const u32x4_t*)(qb + (int64_t)j * ldq + dc), qq);
      unpack8(*(const u32x4_t*)(dob + (int64_t)j * ldo + dc), dd);
      const float s_rj = dS[r][j], s_jr = dS[j][r], p_jr = P[j][r];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        gq[e] += s_rj * kk[e];  // dQ[r] = sum_j dS[r][j] K[j]
        gk[e] += s_jr * qq[e];  // dK[r] = sum_j dS[j][r] Q[j]
        gv[e] += p_jr * dd[e];  // dV[r] = sum_j P[j][r] dO[j]
      }
    }
    const int64_t o = ((int64_t)b * n + r);
    *(u32x4_t*)(dq + o * ldq + h * dh + dc) = pack8(gq);
    *(u32x4_t*)(dk + o * ldkv + h * dh + dc) = pack8(gk);
    *(u32x4_t*)(dv + o * ldkv + h * dh + dc) = pack8(gv);
  }
}
// MFMA form for head_dim % 64 == 0 up to 512 (csrc/attention.hip); DL_ATTN_SMALL_MFMA=0 keeps the f32 VALU kernels below
bool launch_attn_small_mfma_fwd(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldkv, void* out, int64_t ldo,
                                float* probs, int64_t B, int64_t n, int64_t H, int64_t dh, float scale, hipStream_t stream);
bool launch_attn_small_mfma_bwd(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldkv, const void* dout,
                                int64_t ldo, const float* probs, void* dq, void* dk, void* dv, int64_t B, int64_t n, int64_t H,
                                int64_t dh, float scale, hipStream_t stream);
static bool attn_small_use_mfma() { return true; }  // (head widths the MFMA kernels do not take fall to the f32 VALU kernels)
extern "C" int dl_attn_small_fwd(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldkv, void* out, int64_t ldo,
                                 float* probs, int64_t B, int64_t n, int64_t H, int64_t dh, float scale, dl_stream_t stream) {
  DL_CHECK_ARG(q && k && v && out && probs && B > 0 && n > 0 && n <= AS_MAXN && dh % 8 == 0, "dl_attn_small_fwd: n=%lld dh=%lld",
               (long long)n, (long long)dh);
  if (attn_small_use_mfma() && launch_attn_small_mfma_fwd(q, k, v, ldq, ldkv, out, ldo, probs, B, n, H, dh, scale, (hipStream_t)stream)) {
    DL_LAUNCH_CHECK();
    return DL_OK;
  }
  hipLaunchKernelGGL(attn_small_fwd_k, (int)(B * H), 256, 0, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k,
                     (const bf16_t*)v, ldq, ldkv, (bf16_t*)out, ldo, probs, (int)n, (int)H, (int)dh, scale);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_attn_small_bwd(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldkv, const void* dout,
                                 int64_t ldo, const float* probs, void* dq, void* dk, void* dv, int64_t B, int64_t n, int64_t H,
                                 int64_t dh, float scale, dl_stream_t stream) {
  DL_CHECK_ARG(q && k && v && dout && probs && dq && dk && dv && B > 0 && n > 0 && n <= AS_MAXN && dh % 8 == 0,
               "dl_attn_small_bwd: bad args");
  if (attn_small_use_mfma() &&
      launch_attn_small_mfma_bwd(q, k, v, ldq, ldkv, dout, ldo, probs, dq, dk, dv, B, n, H, dh, scale, (hipStream_t)stream)) {
    DL_LAUNCH_CHECK();
    return DL_OK;
  }
  hipLaunchKernelGGL(attn_small_bwd_k, (int)(B * H), 256, 0, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k,
                     (const bf16_t*)v, ldq, ldkv, (const bf16_t*)dout, ldo, probs, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, (int)n,
                     (int)H, (int)dh, scale);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ---------------------------------------------------------------- glue: gradient fan-in add, strided 2-D copy (concat / split / pad)
__global__ void add_bf16_k(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ o, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    o[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
}
extern "C" int dl_add_bf16(const void* a, const void* b, void* out, int64_t n, dl_stream_t stream) {
  DL_CHECK_ARG(a && b && out && n > 0, "dl_add_bf16: bad args");
  hipLaunchKernelGGL(add_bf16_k, grid_for(n), 256, 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
__global__ void copy2d_bf16_k(const bf16_t* __restrict__ s, int64_t lds, bf16_t* __restrict__ d, int64_t ldd, int64_t rows, int cols) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    d[r * ldd + c] = s[r * lds + c];
  }
}
extern "C" int dl_copy2d_bf16(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols, dl_stream_t stream) {
  DL_CHECK_ARG(src && dst && rows > 0 && cols > 0 && lds >= cols && ldd >= cols, "dl_copy2d_bf16: bad args");
  hipLaunchKernelGGL(copy2d_bf16_k, grid_for(rows * cols), 256, 0, (hipStream_t)stream, (const bf16_t*)src, lds, (bf16_t*)dst, ldd,
                     rows, (int)cols);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
