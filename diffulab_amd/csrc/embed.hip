// Stem / head / conditioning kernels of the DiT: patchify (im2row), unpatchify, sinusoidal timestep
// embedding, label-embedding combine, small reductions.  All tiny and HBM/latency bound.
#include "common.h"

// ---------------------------------------------------------------- patchify: x f32 [B,C,H,W] -> tok bf16 [M, ld]
// one thread per (token, feature); feature order CPP: f = (c*p + p1)*p + p2 ; PPC: f = (p1*p + p2)*C + c
__global__ void patchify_k(const float* __restrict__ x, bf16_t* __restrict__ tok, int B, int C, int H, int W, int p,
                           int ld, int order) {
  const int gh = H / p, gw = W / p, F = C * p * p;
  const int64_t total = (int64_t)B * gh * gw * ld, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int f = (int)(i % ld);
    const int64_t t = i / ld;
    float val = 0.f;
    if (f < F) {
      const int wq = (int)(t % gw), hq = (int)((t / gw) % gh), b = (int)(t / ((int64_t)gw * gh));
      int c, p1, p2;
      if (order == DL_PATCH_CPP) {
        p2 = f % p;
        p1 = (f / p) % p;
        c = f / (p * p);
      } else {
        c = f % C;
        p2 = (f / C) % p;
        p1 = f / (C * p);
      }
      val = x[(((int64_t)b * C + c) * H + hq * p + p1) * W + wq * p + p2];
    }
    tok[i] = f2bf(val);
  }
}
extern "C" int dl_patchify(const float* x, void* tok, int64_t B, int64_t C, int64_t H, int64_t W, int64_t p, int64_t ld,
                           int order, dl_stream_t stream) {
  DL_CHECK_ARG(x && tok && B > 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0 && ld >= C * p * p,
               "dl_patchify: bad dims (C=%lld H=%lld W=%lld p=%lld ld=%lld)", (long long)C, (long long)H, (long long)W,
               (long long)p, (long long)ld);
  const int64_t total = B * (H / p) * (W / p) * ld;
  int64_t g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(patchify_k, (int)g, 256, 0, (hipStream_t)stream, x, (bf16_t*)tok, (int)B, (int)C, (int)H, (int)W,
                     (int)p, (int)ld, order);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ---------------------------------------------------------------- unpatchify: tok f32 [M, ld] (p1 p2 c) -> img f32 [B,C,H,W]
__global__ void unpatchify_k(const float* __restrict__ tok, float* __restrict__ img, int B, int C, int H, int W, int p,
                             int ld) {
  const int gh = H / p, gw = W / p;
  const int64_t total = (int64_t)B * C * H * W, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int xw = (int)(i % W), yh = (int)((i / W) % H), c = (int)((i / ((int64_t)W * H)) % C);
    const int b = (int)(i / ((int64_t)W * H * C));
    const int hq = yh / p, p1 = yh % p, wq = xw / p, p2 = xw % p;
    img[i] = tok[(((int64_t)b * gh + hq) * gw + wq) * ld + (p1 * p + p2) * C + c];
  }
}
extern "C" int dl_unpatchify(const float* tok, float* img, int64_t B, int64_t C, int64_t H, int64_t W, int64_t p,
                             int64_t ld, dl_stream_t stream) {
  DL_CHECK_ARG(tok && img && B > 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0 && ld >= C * p * p, "dl_unpatchify: bad dims");
  const int64_t total = B * C * H * W;
  int64_t g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(unpatchify_k, (int)g, 256, 0, (hipStream_t)stream, tok, img, (int)B, (int)C, (int)H, (int)W,
                     (int)p, (int)ld);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ---------------------------------------------------------------- sinusoidal timestep embedding
__global__ void timestep_embedding_k(const float* __restrict__ t, bf16_t* __restrict__ out, int B, int dim,
                                     float neg_log_p_over_half) {
  const int half = dim >> 1;
  const int64_t total = (int64_t)B * half;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / half), j = (int)(i % half);
    const float f = expf((float)j * neg_log_p_over_half);  // exp(-ln(P) * j / half), fp32 like nn.py:107
    const float a = t[b] * f;
    out[(int64_t)b * dim + j] = f2bf(cosf(a));
    out[(int64_t)b * dim + half + j] = f2bf(sinf(a));
  }
}
extern "C" int dl_timestep_embedding(const float* t, void* out, int64_t B, int64_t dim, float max_period,
                                     dl_stream_t stream) {
  DL_CHECK_ARG(t && out && B > 0 && dim > 0 && dim % 2 == 0, "dl_timestep_embedding: dim must be even");
  const int half = (int)(dim / 2);
  int64_t g = (B * half + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(timestep_embedding_k, (int)g, 256, 0, (hipStream_t)stream, t, (bf16_t*)out, (int)B, (int)dim,
                     (float)(-log((double)max_period) / (double)half));
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ---------------------------------------------------------------- emb = e + table[idx]; act = silu(emb)
__global__ void cond_combine_fwd_k(const float* __restrict__ e, const float* __restrict__ table,
                                   const int64_t* __restrict__ idx, float* __restrict__ emb, bf16_t* __restrict__ act,
                                   int B, int E) {
  const int64_t total = (int64_t)B * E;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / E), c = (int)(i % E);
    float v = e[i];
    if (idx) v += table[idx[b] * E + c];
    emb[i] = v;
    act[i] = f2bf(silu_f(v));
  }
}
extern "C" int dl_cond_combine_fwd(const float* e, const float* table, const int64_t* idx, float* emb, void* act,
                                   int64_t B, int64_t E, dl_stream_t stream) {
  DL_CHECK_ARG(e && emb && act && B > 0 && E > 0 && (!idx || table), "dl_cond_combine_fwd: bad args");
  int64_t g = (B * E + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(cond_combine_fwd_k, (int)g, 256, 0, (hipStream_t)stream, e, table, idx, emb, (bf16_t*)act, (int)B,
                     (int)E);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
__global__ void cond_combine_bwd_k(const float* __restrict__ dact, const float* __restrict__ emb,
                                   const int64_t* __restrict__ idx, float* __restrict__ demb,
                                   bf16_t* __restrict__ demb16, float* __restrict__ dtable, int B, int E) {
  const int64_t total = (int64_t)B * E;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / E), c = (int)(i % E);
    const float g = dact[i] * dsilu_f(emb[i]);
    demb[i] = g;
    if (demb16) demb16[i] = f2bf(g);
    if (dtable && idx) {
      // nn.Embedding backward without atomics: the FIRST sample that carries a label owns that table row and adds the
      // contributions of every sample with the same label in batch order (one writer per row, a fixed order: bit-reproducible)
      const int64_t lab = idx[b];
      bool first = true;
      for (int b2 = 0; b2 < b; ++b2) first = first && idx[b2] != lab;
      if (first) {
        float acc = g;
        for (int b2 = b + 1; b2 < B; ++b2)
          if (idx[b2] == lab) acc += dact[(int64_t)b2 * E + c] * dsilu_f(emb[(int64_t)b2 * E + c]);
        dtable[lab * E + c] += acc;
      }
    }
  }
}
extern "C" int dl_cond_combine_bwd(const float* dact, const float* emb, const int64_t* idx, float* demb, void* demb_bf16,
                                   float* dtable, int64_t B, int64_t E, dl_stream_t stream) {
  DL_CHECK_ARG(dact && emb && demb && B > 0 && E > 0, "dl_cond_combine_bwd: bad args");
  int64_t g = (B * E + 255) / 256;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(cond_combine_bwd_k, (int)g, 256, 0, (hipStream_t)stream, dact, emb, idx, demb, (bf16_t*)demb_bf16,
                     dtable, (int)B, (int)E);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

__global__ void silu_bwd_k(const float* __restrict__ dy, const bf16_t* __restrict__ pre, bf16_t* __restrict__ dx,
                           int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dx[i] = f2bf(dy[i] * dsilu_f(bf2f(pre[i])));
}
extern "C" int dl_silu_bwd(const float* dy, const void* pre, void* dx, int64_t n, dl_stream_t stream) {
  DL_CHECK_ARG(dy && pre && dx && n > 0, "dl_silu_bwd: bad args");
  int64_t g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(silu_bwd_k, (int)g, 256, 0, (hipStream_t)stream, dy, (const bf16_t*)pre, (bf16_t*)dx, n);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

__global__ void gelu_bwd_k(const float* __restrict__ dy, const bf16_t* __restrict__ pre, bf16_t* __restrict__ dx,
                           int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dx[i] = f2bf(dy[i] * dgelu_f(bf2f(pre[i])));
}
extern "C" int dl_gelu_bwd(const float* dy, const void* pre, void* dx, int64_t n, dl_stream_t stream) {
  DL_CHECK_ARG(dy && pre && dx && n > 0, "dl_gelu_bwd: bad args");
  int64_t g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(gelu_bwd_k, (int)g, 256, 0, (hipStream_t)stream, dy, (const bf16_t*)pre, (bf16_t*)dx, n);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ---------------------------------------------------------------- head split / merge with optional rotary embedding
// one thread per (token row, head, 8-channel chunk): 16-byte moves; the rotary pairs (2i, 2i+1) of a chunk stay in the thread
__global__ void heads_split_rope_k(const bf16_t* __restrict__ src, int64_t ld, bf16_t* __restrict__ dst, int B, int H, int n_src,
                                   int n_dst, int n_off, const float* __restrict__ cs, const float* __restrict__ sn, int rot,
                                   int backward, int accumulate) {
  const int64_t total = (int64_t)B * n_src * H * 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i & 7);
    const int h = (int)((i >> 3) % H);
    const int64_t row = i / ((int64_t)8 * H);  // b * n_src + n
    const int n = (int)(row % n_src), b = (int)(row / n_src);
    bf16_t* ps = (bf16_t*)src + row * ld + h * 64 + c8 * 8;
    bf16_t* pd = dst + (((int64_t)b * H + h) * n_dst + n_off + n) * 64 + c8 * 8;
    float v[8];
    unpack8(*(const u32x4_t*)(backward ? pd : ps), v);
    const int d0 = c8 * 8;
    if (cs && d0 < rot) {
      const f32x4_t cc = *(const f32x4_t*)(cs + (int64_t)n * (rot >> 1) + (d0 >> 1));
      const f32x4_t ss = *(const f32x4_t*)(sn + (int64_t)n * (rot >> 1) + (d0 >> 1));
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float a = v[2 * p], bq = v[2 * p + 1];
        const float s = backward ? -ss[p] : ss[p];  // the transpose of a rotation is the rotation by -theta
        v[2 * p] = a * cc[p] - bq * s;
        v[2 * p + 1] = a * s + bq * cc[p];
      }
    }
    if (backward) {
      if (accumulate) {
        float o[8];
        unpack8(*(const u32x4_t*)ps, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += o[e];
      }
      *(u32x4_t*)ps = pack8(v);
    } else {
      *(u32x4_t*)pd = pack8(v);
    }
  }
}
static int heads_launch(const void* src, int64_t ld, void* dst, int64_t B, int64_t H, int64_t n_src, int64_t n_dst, int64_t n_off,
                        const float* cs, const float* sn, int64_t rot, int backward, int accumulate, dl_stream_t stream) {
  int64_t g = (B * n_src * H * 8 + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(heads_split_rope_k, (int)g, 256, 0, (hipStream_t)stream, (const bf16_t*)src, ld, (bf16_t*)dst, (int)B, (int)H,
                     (int)n_src, (int)n_dst, (int)n_off, cs, sn, (int)rot, backward, accumulate);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_heads_split_rope(const void* src, int64_t ld, void* dst, int64_t B, int64_t H, int64_t n_src, int64_t n_dst,
                                   int64_t n_off, const float* cs, const float* sn, int64_t rot, dl_stream_t stream) {
  DL_CHECK_ARG(src && dst && B > 0 && H > 0 && n_src > 0 && n_off >= 0 && n_off + n_src <= n_dst && ld % 8 == 0 && ld >= H * 64 &&
                   rot % 8 == 0 && rot <= 64 && ((cs == nullptr) == (sn == nullptr)),
               "dl_heads_split_rope: bad args");
  return heads_launch(src, ld, dst, B, H, n_src, n_dst, n_off, cs, sn, rot, 0, 0, stream);
}
extern "C" int dl_heads_merge_rope_bwd(const void* dst_grad, void* src_grad, int64_t ld, int64_t B, int64_t H, int64_t n_src,
                                       int64_t n_dst, int64_t n_off, const float* cs, const float* sn, int64_t rot,
                                       int accumulate, dl_stream_t stream) {
  DL_CHECK_ARG(dst_grad && src_grad && B > 0 && H > 0 && n_src > 0 && n_off >= 0 && n_off + n_src <= n_dst && ld % 8 == 0 &&
                   ld >= H * 64 && rot % 8 == 0 && rot <= 64 && ((cs == nullptr) == (sn == nullptr)),
               "dl_heads_merge_rope_bwd: bad args");
  return heads_launch(src_grad, ld, (void*)dst_grad, B, H, n_src, n_dst, n_off, cs, sn, rot, 1, accumulate, stream);
}

// ---------------------------------------------------------------- column sums: out[c] += sum_r x[r, c]
// block = 256 threads = 64 columns x 4 row-lanes; grid.x over column blocks, grid.y over row slabs
template <typename T>
__device__ __forceinline__ float ldf(const T* p);
template <>
__device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }

template <typename T>
__global__ void colsum_k(const T* __restrict__ x, int64_t ld, float* __restrict__ out, int64_t R, int C,
                         int rows_per_slab) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
  const int64_t r1 = r0 + rows_per_slab < R ? r0 + rows_per_slab : R;
  float acc = 0.f;
  if (c < C)
    for (int64_t r = r0 + rl; r < r1; r += 4) acc += ldf<T>(x + r * ld + c);
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && c < C) unsafeAtomicAdd(&out[c], red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}
// the same sums with ONE writer per column and a fixed order: every row slab stores its partial row in `part` [slabs, C]
template <typename T>
__global__ void colsum_part_k(const T* __restrict__ x, int64_t ld, float* __restrict__ part, int64_t R, int C, int rows_per_slab) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
  const int64_t r1 = r0 + rows_per_slab < R ? r0 + rows_per_slab : R;
  float acc = 0.f;
  if (c < C)
    for (int64_t r = r0 + rl; r < r1; r += 4) acc += ldf<T>(x + r * ld + c);
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && c < C) part[(int64_t)blockIdx.y * C + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
// bf16 rows, 16 bytes per lane: VL lanes (a power of two <= 64) cover VL * 8 consecutive columns, the other 256 / VL thread rows
// take interleaved rows of the slab.  (One bf16 per lane -- 128 bytes per wave-load -- read the [65536, 13312] per-token adaLN gradient
// of the DDT decoder at 2.1 TB/s: 800 us per step.)  ATOMIC: slabs meet in out[] through f32 atomics; else one partial row per slab.
template <bool ATOMIC>
__global__ __launch_bounds__(256) void colsum_vec_k(const bf16_t* __restrict__ x, int64_t ld, float* __restrict__ dst, int64_t R, int C,
                                                    int rows_per_slab, int vl) {
  __shared__ float red[256 * 8];
  const int lc = threadIdx.x & (vl - 1), lr = threadIdx.x / vl, nr = 256 / vl;
  const int c = (blockIdx.x * vl + lc) * 8;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
  const int64_t r1 = r0 + rows_per_slab < R ? r0 + rows_per_slab : R;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    int64_t r = r0 + lr;
    for (; r + nr < r1; r += 2 * nr) {  // two independent loads in flight per lane
      float a[8], b[8];
      const u32x4_t va = *(const u32x4_t*)(x + r * ld + c), vb = *(const u32x4_t*)(x + (r + nr) * ld + c);
      unpack8(va, a);
      unpack8(vb, b);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += a[e] + b[e];
    }
    if (r < r1) {
      float a[8];
      unpack8(*(const u32x4_t*)(x + r * ld + c), a);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += a[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[(lr * vl + lc) * 8 + e] = acc[e];
  __syncthreads();
  // thread t < vl * 8 owns column t of the block: fixed order over the nr thread rows
  for (int t = threadIdx.x; t < vl * 8; t += 256) {
    const int col = blockIdx.x * vl * 8 + t;
    if (col >= C) continue;
    float sum = 0.f;
    for (int q = 0; q < nr; ++q) sum += red[q * vl * 8 + t];
    if (ATOMIC) unsafeAtomicAdd(&dst[col], sum);
    else dst[(int64_t)blockIdx.y * C + col] = sum;
  }
}
static bool colsum_vec_ok(const void* x, int dtype, int64_t ld, int64_t C) {
  return dtype != DL_F32 && C % 8 == 0 && ld % 8 == 0 && (((uintptr_t)x) & 15) == 0;
}
static int colsum_vl(int64_t C) {
  int vl = 1;
  while (vl * 8 < C && vl < 64) vl <<= 1;
  return vl;
}
int launch_fold_partials(const float* part, int64_t stride, int splits, int64_t M, int64_t N, int64_t ldp, float* C, int64_t ldc,
                         int accumulate, hipStream_t stream);  // gemm.hip
extern "C" int dl_colsum_det(const void* x, int dtype, int64_t ld, float* out, int64_t R, int64_t C, float* scratch,
                             int64_t scratch_floats, dl_stream_t stream) {
  DL_CHECK_ARG(x && out && scratch && R > 0 && C > 0 && ld >= C && scratch_floats >= C, "dl_colsum_det: bad args");
  int slabs = (int)((R + 255) / 256);
  if (slabs > 512) slabs = 512;
  if ((int64_t)slabs * C > scratch_floats) slabs = (int)(scratch_floats / C);
  const int rps = (int)((R + slabs - 1) / slabs);
  slabs = (int)((R + rps - 1) / rps);
  dim3 grid(cdiv(C, 64), slabs);
  if (colsum_vec_ok(x, dtype, ld, C)) {
    const int vl = colsum_vl(C);
    hipLaunchKernelGGL(colsum_vec_k<false>, dim3(cdiv(C, vl * 8), slabs), 256, 0, (hipStream_t)stream, (const bf16_t*)x, ld, scratch, R,
                       (int)C, rps, vl);
  } else if (dtype == DL_F32)
    hipLaunchKernelGGL(colsum_part_k<float>, grid, 256, 0, (hipStream_t)stream, (const float*)x, ld, scratch, R, (int)C, rps);
  else
    hipLaunchKernelGGL(colsum_part_k<bf16_t>, grid, 256, 0, (hipStream_t)stream, (const bf16_t*)x, ld, scratch, R, (int)C, rps);
  DL_LAUNCH_CHECK();
  return launch_fold_partials(scratch, C, slabs, 1, C, C, out, C, 1, (hipStream_t)stream);
}
extern "C" int dl_colsum(const void* x, int dtype, int64_t ld, float* out, int64_t R, int64_t C, dl_stream_t stream) {
  DL_CHECK_ARG(x && out && R > 0 && C > 0 && ld >= C, "dl_colsum: bad args");
  int slabs = (int)((R + 255) / 256);
  if (slabs > 512) slabs = 512;
  const int rps = (int)((R + slabs - 1) / slabs);
  dim3 grid(cdiv(C, 64), slabs);
  if (colsum_vec_ok(x, dtype, ld, C)) {
    const int vl = colsum_vl(C);
    hipLaunchKernelGGL(colsum_vec_k<true>, dim3(cdiv(C, vl * 8), slabs), 256, 0, (hipStream_t)stream, (const bf16_t*)x, ld, out, R, (int)C,
                       rps, vl);
  } else if (dtype == DL_F32)
    hipLaunchKernelGGL(colsum_k<float>, grid, 256, 0, (hipStream_t)stream, (const float*)x, ld, out, R, (int)C, rps);
  else
    hipLaunchKernelGGL(colsum_k<bf16_t>, grid, 256, 0, (hipStream_t)stream, (const bf16_t*)x, ld, out, R, (int)C, rps);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// several column sums in ONE launch (the bias gradients of a stretch of the UNet's backward: 105 launches of 5-40 us per step on the
// side stream otherwise): the descriptors travel by value in the kernel arguments (the operands are activation gradients whose
// addresses change from step to step: no device table to refresh); a block finds its problem by its first block index
#define DL_COLSUM_BATCH_MAX 24
struct ColsumBatch {
  const bf16_t* x[DL_COLSUM_BATCH_MAX];
  float* out[DL_COLSUM_BATCH_MAX];
  int64_t ld[DL_COLSUM_BATCH_MAX];
  int64_t R[DL_COLSUM_BATCH_MAX];
  int C[DL_COLSUM_BATCH_MAX], rps[DL_COLSUM_BATCH_MAX], vl[DL_COLSUM_BATCH_MAX], begin[DL_COLSUM_BATCH_MAX + 1];
  int n;
};
__global__ __launch_bounds__(256) void colsum_batched_k(ColsumBatch a) {
  __shared__ float red[256 * 8];
  int e = 0;
  while (e + 1 < a.n && a.begin[e + 1] <= (int)blockIdx.x) ++e;
  const int local = (int)blockIdx.x - a.begin[e];
  const int vl = a.vl[e], C = a.C[e], rows_per_slab = a.rps[e];
  const int ncb = (C + vl * 8 - 1) / (vl * 8);
  const int bx = local % ncb, by = local / ncb;
  const bf16_t* x = a.x[e];
  const int64_t ld = a.ld[e], R = a.R[e];
  const int lc = threadIdx.x & (vl - 1), lr = threadIdx.x / vl, nr = 256 / vl;
  const int c = (bx * vl + lc) * 8;
  const int64_t r0 = (int64_t)by * rows_per_slab;
  const int64_t r1 = r0 + rows_per_slab < R ? r0 + rows_per_slab : R;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    int64_t r = r0 + lr;
    for (; r + nr < r1; r += 2 * nr) {
      float p[8], q[8];
      const u32x4_t va = *(const u32x4_t*)(x + r * ld + c), vb = *(const u32x4_t*)(x + (r + nr) * ld + c);
      unpack8(va, p);
      unpack8(vb, q);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += p[k] + q[k];
    }
    if (r < r1) {
      float p[8];
      unpack8(*(const u32x4_t*)(x + r * ld + c), p);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += p[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[(lr * vl + lc) * 8 + k] = acc[k];
  __syncthreads();
  for (int t = threadIdx.x; t < vl * 8; t += 256) {
    const int col = bx * vl * 8 + t;
    if (col >= C) continue;
    float sum = 0.f;
    for (int q = 0; q < nr; ++q) sum += red[q * vl * 8 + t];
    unsafeAtomicAdd(&a.out[e][col], sum);
  }
}
extern "C" int dl_colsum_batched(const dl_colsum_desc_t* desc, int n, dl_stream_t stream) {
  DL_CHECK_ARG(desc && n > 0, "dl_colsum_batched: bad args");
  ColsumBatch a;
  int k = 0, blocks = 0;
  auto flush = [&]() {
    if (k == 0) return;
    a.n = k;
    a.begin[k] = blocks;
    hipLaunchKernelGGL(colsum_batched_k, blocks, 256, 0, (hipStream_t)stream, a);
    k = 0;
    blocks = 0;
  };
  for (int i = 0; i < n; ++i) {
    const dl_colsum_desc_t& d = desc[i];
    DL_CHECK_ARG(d.x && d.out && d.R > 0 && d.C > 0 && d.ld >= d.C, "dl_colsum_batched: bad descriptor %d", i);
    if (!colsum_vec_ok(d.x, DL_BF16, d.ld, d.C)) {  // (odd widths / alignment: the scalar kernel, a launch of its own)
      const int rc = dl_colsum(d.x, DL_BF16, d.ld, d.out, d.R, d.C, stream);
      if (rc != DL_OK) return rc;
      continue;
    }
    int slabs = (int)((d.R + 255) / 256);
    if (slabs > 512) slabs = 512;
    const int rps = (int)((d.R + slabs - 1) / slabs);
    slabs = (int)((d.R + rps - 1) / rps);
    const int vl = colsum_vl(d.C);
    a.x[k] = (const bf16_t*)d.x;
    a.out[k] = d.out;
    a.ld[k] = d.ld;
    a.R[k] = d.R;
    a.C[k] = (int)d.C;
    a.rps[k] = rps;
    a.vl[k] = vl;
    a.begin[k] = blocks;
    blocks += (int)cdiv(d.C, vl * 8) * slabs;
    if (++k == DL_COLSUM_BATCH_MAX) flush();
  }
  flush();
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// out[j] += sum_g partial[g, j]: 64 columns x 4 row-lanes per block, the G rows cut into slices of 64 (one block per
// (column group, slice), partial sums meet in `out` through f32 atomics); `clear` zeroes every partial element right after
// it is read, so an accumulate-into partial buffer needs no separate memset
__global__ void reduce_rows_k(float* __restrict__ partial, float* __restrict__ out, int G, int64_t n, int clear) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + cl;
  const int g0 = blockIdx.y * 64, g1 = g0 + 64 < G ? g0 + 64 : G;
  float acc = 0.f;
  if (c < n)
    for (int g = g0 + rl; g < g1; g += 4) {
      acc += partial[(int64_t)g * n + c];
      if (clear) partial[(int64_t)g * n + c] = 0.f;
    }
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && c < n) unsafeAtomicAdd(&out[c], red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]);
}
extern "C" int dl_reduce_rows_f32(float* partial, float* out, int64_t G, int64_t n, int clear_partial, dl_stream_t stream) {
  DL_CHECK_ARG(partial && out && G > 0 && n > 0, "dl_reduce_rows_f32: bad args");
  hipLaunchKernelGGL(reduce_rows_k, dim3(cdiv(n, 64), cdiv(G, 64)), 256, 0, (hipStream_t)stream, partial, out, (int)G, n,
                     clear_partial);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// K independent folds in ONE launch, each with a single writer per output element and a fixed summation order (deterministic):
// out[k * out_stride + j] += sum_g partial[k * partial_stride + g * n + j].  The LayerNorm-affine gradients of all blocks are
// folded by two of these at the end of the backward (norm_1 and norm_2 families) instead of 2 * depth starved launches in
// between the side-stream weight-gradient GEMMs.
__global__ __launch_bounds__(256) void reduce_rows_batched_k(const float* __restrict__ partial, int64_t partial_stride,
                                                             float* __restrict__ out, int64_t out_stride, int G, int64_t n) {
  // 16 columns x 16 row lanes per block (a thread walks G / 16 rows: 64 columns x 4 lanes left each thread 64 dependent-latency
  // loads at G = 256, 28 us per launch for 9 MB); the lanes meet in LDS in a fixed tree
  __shared__ float red[16][16];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int64_t c = (int64_t)blockIdx.x * 16 + cl;
  const float* p = partial + (int64_t)blockIdx.y * partial_stride;
  float acc = 0.f;
  if (c < n)
    for (int g = rl; g < G; g += 16) acc += p[(int64_t)g * n + c];
  red[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && c < n) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; q += 4) s += (red[q][cl] + red[q + 1][cl]) + (red[q + 2][cl] + red[q + 3][cl]);
    out[(int64_t)blockIdx.y * out_stride + c] += s;
  }
}
extern "C" int dl_reduce_rows_batched_f32(const float* partial, int64_t partial_stride, float* out, int64_t out_stride, int64_t K,
                                          int64_t G, int64_t n, dl_stream_t stream) {
  DL_CHECK_ARG(partial && out && K > 0 && G > 0 && n > 0, "dl_reduce_rows_batched_f32: bad args");
  hipLaunchKernelGGL(reduce_rows_batched_k, dim3(cdiv(n, 16), (unsigned)K), 256, 0, (hipStream_t)stream, partial, partial_stride, out,
                     out_stride, (int)G, n);
  DL_LAUNCH_CHECK();
  return DL_OK;
}

// ------------------------------------------------------------------------------------------------ cosine similarity rows (RePA)
// F.cosine_similarity(p, d, dim=-1) (training/losses/repa.py:196): cos = <p,d> / sqrt(max(|p|^2 |d|^2, eps^2)); one wave per row
__global__ __launch_bounds__(256) void cosine_rows_fwd_k(const bf16_t* __restrict__ p, int64_t ldp, const float* __restrict__ d,
                                                         int64_t ldd, float* __restrict__ cosv, float* __restrict__ pn2,
                                                         float* __restrict__ dn2, int64_t M, int E, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
    float s = 0.f, a = 0.f, b = 0.f;
    for (int c = lane * 8; c < E; c += 512) {
      float pv[8];
      unpack8(*(const u32x4_t*)(p + row * ldp + c), pv);
      const f32x4_t d0 = *(const f32x4_t*)(d + row * ldd + c), d1 = *(const f32x4_t*)(d + row * ldd + c + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dv = e < 4 ? d0[e] : d1[e - 4];
        s += pv[e] * dv;
        a += pv[e] * pv[e];
        b += dv * dv;
      }
    }
    s = wave_sum(s);
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) {
      cosv[row] = s / sqrtf(fmaxf(a * b, eps * eps));
      pn2[row] = a;
      dn2[row] = b;
    }
  }
}
// dp = g * ( d / den - cos * p / |p|^2 ),  den = sqrt(max(|p|^2 |d|^2, eps^2)),  g = gscale * (*gscale_dev)
__global__ __launch_bounds__(256) void cosine_rows_bwd_k(const bf16_t* __restrict__ p, int64_t ldp, const float* __restrict__ d,
                                                         int64_t ldd, const float* __restrict__ cosv, const float* __restrict__ pn2,
                                                         const float* __restrict__ dn2, float gscale,
                                                         const float* __restrict__ gscale_dev, bf16_t* __restrict__ dp,
                                                         int64_t lddp, int64_t M, int E, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float g = gscale * (gscale_dev ? *gscale_dev : 1.0f);
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
    const float a = pn2[row], b = dn2[row], cs = cosv[row];
    const bool clamped = a * b < eps * eps;  // the clamp is active: the denominator is a constant there
    const float inv_den = 1.0f / sqrtf(fmaxf(a * b, eps * eps));
    const float k = clamped ? 0.f : cs / fmaxf(a, 1e-30f);
    for (int c = lane * 8; c < E; c += 512) {
      float pv[8];
      unpack8(*(const u32x4_t*)(p + row * ldp + c), pv);
      const f32x4_t d0 = *(const f32x4_t*)(d + row * ldd + c), d1 = *(const f32x4_t*)(d + row * ldd + c + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) pv[e] = g * ((e < 4 ? d0[e] : d1[e - 4]) * inv_den - k * pv[e]);
      *(u32x4_t*)(dp + row * lddp + c) = pack8(pv);
    }
  }
}
extern "C" int dl_cosine_rows_fwd(const void* p, int64_t ldp, const float* d, int64_t ldd, float* cosv, float* pn2, float* dn2,
                                  int64_t M, int64_t E, float eps, dl_stream_t stream) {
  DL_CHECK_ARG(p && d && cosv && pn2 && dn2 && M > 0 && E > 0 && E % 8 == 0 && ldp % 8 == 0 && ldd % 4 == 0 &&
                   (((uintptr_t)p | (uintptr_t)d) & 15) == 0,
               "dl_cosine_rows_fwd: bad args");
  int grid = cdiv(M, 4);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(cosine_rows_fwd_k, grid, 256, 0, (hipStream_t)stream, (const bf16_t*)p, ldp, d, ldd, cosv, pn2, dn2, M, (int)E,
                     eps);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
extern "C" int dl_cosine_rows_bwd(const void* p, int64_t ldp, const float* d, int64_t ldd, const float* cosv, const float* pn2,
                                  const float* dn2, float gscale, const float* gscale_dev, void* dp, int64_t lddp, int64_t M,
                                  int64_t E, float eps, dl_stream_t stream) {
  DL_CHECK_ARG(p && d && cosv && pn2 && dn2 && dp && M > 0 && E > 0 && E % 8 == 0 && ldp % 8 == 0 && lddp % 8 == 0 && ldd % 4 == 0,
               "dl_cosine_rows_bwd: bad args");
  int grid = cdiv(M, 4);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(cosine_rows_bwd_k, grid, 256, 0, (hipStream_t)stream, (const bf16_t*)p, ldp, d, ldd, cosv, pn2, dn2, gscale,
                     gscale_dev, (bf16_t*)dp, lddp, M, (int)E, eps);
  DL_LAUNCH_CHECK();
  return DL_OK;
}
