/*
 * libdiffulab_hip.so -- C ABI of the MI355X (gfx950) denoising hot path of DiffuLab.
 *
 * The reference (LouisRouss/DiffuLab) is pure Python on stock ATen ops and has NO FFI of its own
 * (SURVEY.md §2.1); this header is therefore the boundary the build introduces, and every entry point
 * names the reference site (path relative to /root/reference/src/diffulab, file:line) whose arithmetic
 * it replaces.  INTEGRATION.md shows the ctypes stubs a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 (DL_OK) or a negative DL_ERR_*; dl_last_error() gives the thread-local text;
 *     nothing throws across the ABI.
 *   - all pointers are DEVICE pointers owned by the caller (torch allocations in the Python host);
 *     the library allocates no device memory and keeps no global mutable state.
 *   - `stream` is the caller's hipStream_t (torch.cuda.current_stream().cuda_stream); launches are async.
 *   - "bf16" tensors are raw uint16 bfloat16 bits, row-major; `ld*` are row strides in ELEMENTS.
 *   - token tensors are [M, D] with M = batch * tokens; `rows_per_*` = tokens per batch element for
 *     per-sample broadcast operands (modulation / gates), 1 for per-token operands (DDT-style).
 */
#ifndef DIFFULAB_HIP_H
#define DIFFULAB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DL_API __attribute__((visibility("default")))

typedef void* dl_stream_t;

enum { DL_OK = 0, DL_ERR_INVALID = -1, DL_ERR_LAUNCH = -2, DL_ERR_UNSUPPORTED = -3 };
enum { DL_BF16 = 0, DL_F32 = 1 };
enum { DL_ACT_NONE = 0, DL_ACT_SILU = 1, DL_ACT_GELU = 2 };  /* GELU: exact erf form (nn.GELU default) */
enum { DL_LOSS_FLOW = 0, DL_LOSS_EPS = 1 };                       /* target = a - b | target = a */
enum { DL_MEAN_EPSILON = 0, DL_MEAN_XSTART = 1, DL_MEAN_XPREV = 2 };
enum { DL_PATCH_CPP = 0, DL_PATCH_PPC = 1 };                      /* feature order (c p1 p2) | (p1 p2 c) */

/* ------------------------------------------------------------------ library */
DL_API int dl_version(void);
DL_API const char* dl_last_error(void);
/* name_len bytes of `arch` receive e.g. "gfx950:sramecc+:xnack-" */
DL_API int dl_device_info(int device, int* cu_count, int* lds_bytes_per_cu, int64_t* hbm_bytes, char* arch,
                          int arch_len);

/* a HIP stream whose kernels may only run on the compute units whose bit is set in cu_mask[words] (bit i of word i/32 = CU i);
 * the caller owns the handle (wrap it, e.g. torch.cuda.ExternalStream) and destroys it with dl_stream_destroy */
DL_API int dl_stream_create_masked(const uint32_t* cu_mask, int words, void** stream_out);
/* a HIP stream of the device's lowest priority (side stream of the weight gradients); range_out (may be NULL) = {least, greatest} */
DL_API int dl_stream_create_low_priority(void** stream_out, int* range_out);
DL_API int dl_stream_destroy(void* stream);
/* stream hand-off without host objects: everything queued on `waited` so far happens before anything queued on `waiter` from now
 * on (event record + stream wait); dl_memset_zero = hipMemsetAsync(p, 0, bytes) on the caller's stream.  With these two every
 * host-side action of an engine's training step is a C call with plain arguments, i.e. a step can be recorded once and re-issued
 * as a list (diffulab_amd/ops.py::LaunchPlan) -- the host-bound configurations' answer to a graph launch that costs more than the
 * eager launches on this runtime. */
DL_API int dl_stream_wait_stream(dl_stream_t waiter, dl_stream_t waited);
DL_API int dl_memset_zero(void* p, int64_t bytes, dl_stream_t stream);

/* ------------------------------------------------------------------ diffusion heads (f32 images [B, chw]) */
/* Flow.add_noise, diffuse/modelizations/flow.py:401-408:  z = (1 - t[b]) x + t[b] eps */
DL_API int dl_flow_add_noise(const float* x, const float* noise, const float* t, float* z, int64_t batch,
                             int64_t chw, dl_stream_t stream);
/* GaussianDiffusion.add_noise, gaussian_diffusion.py:338-341 + diffuse/utils.py:16:
 * xt = sqrt_ab[t] x + sqrtf(1 - ab[t]) eps ; tables are the .float() casts of the fp64 tables */
DL_API int dl_ddpm_add_noise(const float* x, const float* noise, const int32_t* t, const float* sqrt_ab,
                             const float* ab, float* xt, int64_t batch, int64_t chw, dl_stream_t stream);
/* Flow.compute_loss flow.py:306-309 (mode DL_LOSS_FLOW, a=noise, b=x0) and GaussianDiffusion.compute_loss
 * gaussian_diffusion.py:306 (mode DL_LOSS_EPS, a=noise): loss = mean((target - pred)^2).
 * `partial` is caller scratch of dl_mse_loss_partials(n) floats; reduction order is fixed (deterministic). */
DL_API int64_t dl_mse_loss_partials(int64_t n);
DL_API int dl_mse_loss_fwd(const float* pred, const float* a, const float* b, float* partial, float* loss,
                           int64_t n, int mode, dl_stream_t stream);
/* dpred = gscale * (*gscale_dev) * 2 (pred - target) / n   (upstream d(total)/d(loss): host scalar times an
 * optional DEVICE scalar, so autograd's incoming gradient never forces a device->host sync) */
DL_API int dl_mse_loss_bwd(const float* pred, const float* a, const float* b, float gscale,
                           const float* gscale_dev, float* dpred, int64_t n, int mode, dl_stream_t stream);
/* Flow.compute_loss x-prediction branch flow.py:300-303: v = (z - xhat) / t[b] ; backward dxhat = -dv / t[b] */
DL_API int dl_flow_x_to_v(const float* z, const float* xhat, const float* t, float* v, int64_t batch,
                          int64_t chw, dl_stream_t stream);
DL_API int dl_flow_x_to_v_bwd(const float* dv, const float* t, float* dxhat, int64_t batch, int64_t chw,
                              dl_stream_t stream);

/* ------------------------------------------------------------------ sampler steps */
/* Euler.step euler.py:37-41 fused with the CFG combine flow.py:257-259 (v_uncond may be NULL):
 * v = v_uncond + g (v - v_uncond); x_prev = x - v dt; x0 = x - v t_curr   (x0_est may be NULL) */
DL_API int dl_euler_step(const float* x, const float* v, const float* v_uncond, float guidance, float t_curr,
                         float dt, float* x_prev, float* x0_est, int64_t n, dl_stream_t stream);
/* EulerMaruyama.step euler_meruyama.py:39-57.  sigma, std = sigma*sqrt(dt) are host scalars (python floats in
 * the reference).  Exactly one of noise / x_prev_in is non-NULL.  logprob may be NULL. */
DL_API int dl_euler_maruyama_step(const float* x, const float* v, const float* v_uncond, float guidance,
                                  const float* noise, const float* x_prev_in, float t_curr, float dt, float sigma,
                                  float std, float* x_prev, float* mean, float* x0_est, float* logprob,
                                  int64_t n, dl_stream_t stream);
/* DDPM.step ddpm.py:330-363 (+ CFG combine gaussian_diffusion.py:253-255), fixed variances.
 * tables: f32 [6][T] = sqrt_ab, ab, posterior_mean_coef1, coef2, variance, log_variance (already
 * selected for fixed_small / fixed_large by the host).  noise replaces randn_like (ddpm.py:302). */
DL_API int dl_ddpm_step(const float* pred, const float* pred_uncond, float guidance, const float* xt,
                        const float* noise, const int32_t* t, const float* tables, int32_t T, int mean_type,
                        int clamp_x, float* x_prev, float* x0_est, float* mean, float* std, float* logprob,
                        int64_t batch, int64_t chw, dl_stream_t stream);
/* DDIM.step ddim.py:68-103.  tables: f32 [3][T] = sqrt_ab, ab, ab_prev.  std/logprob may be NULL (eta == 0). */
DL_API int dl_ddim_step(const float* pred, const float* pred_uncond, float guidance, const float* xt,
                        const float* noise, const int32_t* t, const float* tables, int32_t T, int mean_type,
                        const float* ddpm_coefs /* f32 [2][T] coef1, coef2; only for DL_MEAN_XPREV */,
                        int clamp_x, float eta, float* x_prev, float* x0_est, float* mean, float* std,
                        float* logprob, int64_t batch, int64_t chw, dl_stream_t stream);

/* ------------------------------------------------------------------ GEMMs (bf16 in, f32 MFMA accumulate) */
/* nn.Linear forward / dgrad (mmdit.py:81,102,260-264; nn.py:530; mmdit.py:542-548):
 *   acc[m,n] = sum_k A[m,k] * B[n,k]           A:[M,K] lda, B:[N,K] ldb (torch Linear weight layout)
 *   pre      = acc + bias[n]                    (bias f32, may be NULL)
 *   if pre_out: pre_out[m,n] = bf16(pre)        (saved pre-activation, ld = ldc)
 *   val      = act(pre)
 *   if resid: val = resid[m,n] + gate[m / rows_per_gate, n] * val      (gate may be NULL -> 1)
 *   C[m,n]   = val  (bf16 or f32 per out_dtype)
 * K must be a multiple of 64 (callers zero-pad tiny K); M, N arbitrary. */
DL_API int dl_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M,
                      int64_t N, int64_t K, const float* bias, int act, int out_dtype, void* pre_out,
                      const void* resid, int64_t ldr, const void* gate, int64_t ldg, int64_t rows_per_gate,
                      dl_stream_t stream);
/* Two independent bf16 products C_i = A_i B_i^T (+ bias_i f32 [N_i], + resid_i bf16 rows) in ONE launch when both are small (together
 * at most 1.5 of the 128 x 128 kernel's workgroups per CU); otherwise, and for n == 1, exactly the dl_gemm_nt calls it stands for.
 * Bit-identical to those calls.  (The q and kv projections of a UNet AttentionBlock -- unet.py:497-533 -- and their data gradients at
 * the low-resolution levels: 64 + 128 tiles on 256 CUs.) */
typedef struct dl_nt_problem_t {
  const void* A;      /* bf16 [M, lda >= K] */
  int64_t lda;
  const void* B;      /* bf16 [N, ldb >= K] */
  int64_t ldb;
  void* C;            /* bf16 [M, ldc >= N] */
  int64_t ldc;
  int64_t M, N, K;    /* K % 64 == 0 */
  const float* bias;  /* f32 [N] or NULL */
  const void* resid;  /* bf16 [M, ldr] or NULL */
  int64_t ldr;
} dl_nt_problem_t;
DL_API int dl_gemm_nt_pair(const dl_nt_problem_t* problems, int n, dl_stream_t stream);
/* mlp_input[0] + PackedSwiGLU fused (mmdit.py:260-264, nn.py:484-486): U[M,2F] = X Wp^T written in the reference
 * layout [x1 | x3] (kept for the backward; U == NULL skips it: inference / sampler loops) and H[M,F] = silu(x1) * x3, in
 * ONE pass over the accumulators.
 * Wp = row-permuted bf16 shadow of the [2F, K] weight made by dl_cast_weight_swiglu.  Only shapes served by the
 * big-tile kernels (M % 256 == 0, 2F % 128 == 0, >= 64 tiles) -- otherwise DL_ERR_UNSUPPORTED and the caller runs
 * dl_gemm_nt + dl_swiglu_fwd. */
DL_API int dl_gemm_nt_swiglu(const void* X, int64_t ldx, const void* Wp, int64_t ldw, void* U, int64_t ldu, void* H,
                             int64_t ldh, int64_t M, int64_t F, int64_t K, dl_stream_t stream);
/* PackedSwiGLU MLP backward WITHOUT saved pre-activations (nn.py:478-486, mmdit.py:260-264): dU = [dH x3 silu'(x1) | dH silu(x1)]
 * with u = [x1 | x3] = X Wp^T RECOMPUTED per 256 x 128-unit tile (rounded to bf16 as the forward would have stored it) and
 * dH = dT W2t^T, neither of them written to memory.  X = the MLP input (modulated LayerNorm output) [M, K1], Wp = the row-permuted
 * [2F, K1] shadow of dl_cast_weight_swiglu, dT [M, K2] = gradient of the MLP output, W2t [F, K2] = transposed shadow of the
 * MLP-down weight.  With it the training forward calls dl_gemm_nt_swiglu with U = NULL (h only).  M % 256 == 0, F % 128 == 0,
 * K1, K2 % 64 == 0, >= 64 tiles -- otherwise DL_ERR_UNSUPPORTED. */
DL_API int dl_mlp_dswiglu_recompute(const void* X, int64_t ldx, const void* Wp, int64_t ldwp, const void* dT, int64_t ldt,
                                    const void* W2t, int64_t ldw2, void* dU, int64_t lddu, int64_t M, int64_t F, int64_t K1,
                                    int64_t K2, dl_stream_t stream);
/* nn.Linear wgrad:  C[m,n] += sum_r A[r,m] * B[r,n]   A:[R,M] lda, B:[R,N] ldb, C f32 [M,N] ldc (atomic
 * accumulate across the split of R; caller zeroes C once per optimizer step).  R multiple of 64. */
DL_API int dl_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                      int64_t N, int64_t R, dl_stream_t stream);
/* same, with a cap on the persistent workgroups of the big-tile kernel (0 = one per CU): a wgrad GEMM that runs on a side
 * stream beside the main dependency chain leaves CUs free for that chain's latency-bound kernels */
DL_API int dl_gemm_tn_ex(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                         int64_t N, int64_t R, int max_workgroups, dl_stream_t stream);
/* Up to four nn.Linear weight gradients over the SAME R token rows in ONE launch, WITHOUT atomics (the four linears of a
 * transformer block: mmdit.py:75-104 qkv / proj_out, mmdit.py:260-264 the two MLP linears):  g_p[m,n] += sum_r dy_p[r,m] x_p[r,n].
 * Every (token range, 384 x 192 tile) workgroup writes its f32 partial tile into slab[range] with plain stores and a fold kernel adds
 * the ranges in a fixed order, so two runs give bit-identical gradients.  EVERY problem a whole number of 384 x 192 tiles (m_out %
 * 384 == 0, n_in % 192 == 0: inner widths 384, 768, ...) or every problem a whole number of 256 x 256 tiles (512-wide models);
 * R % 32 == 0, R >= 2048 -- otherwise DL_ERR_UNSUPPORTED (the caller keeps dl_gemm_tn_ex per problem).  slab: caller-owned f32 scratch of slab_floats >=
 * sum_p m_out_p * n_in_p elements (contents irrelevant on entry, undefined on return); the number of token ranges is
 * min(workgroup budget / tiles, slab_floats / that sum), so 8 x the sum lets one launch fill the chip at D = 384.  max_workgroups
 * caps the grid like dl_gemm_tn_ex (0 = one workgroup per CU). */
typedef struct dl_wgrad_t {
  const void* dy;   /* bf16 [R, m_out], row stride ld_dy: gradient of the linear's output */
  int64_t ld_dy;
  const void* x;    /* bf16 [R, n_in], row stride ld_x: the linear's input */
  int64_t ld_x;
  float* g;         /* f32 [m_out, n_in] contiguous: the weight gradient, accumulated into (+=) */
  int64_t m_out, n_in;
} dl_wgrad_t;
DL_API int dl_gemm_tn_group(const dl_wgrad_t* probs, int n_probs, int64_t R, float* slab, int64_t slab_floats, int max_workgroups,
                            dl_stream_t stream);
/* Bit-reproducible forms of the small, skinny GEMMs and column sums whose default kernels split their contraction over
 * workgroups and meet in f32 atomics (the head's and the patch embedding's weight gradients over all tokens, the conditioning
 * path's products, bias gradients): every split stores its partial image into the caller-owned f32 `scratch` and one fold adds
 * the images in a fixed order.  The number of splits is min(what fills the chip, scratch_floats / (M N)), at least 1.
 *   dl_gemm_tn_det      C[M,N] (f32) += A[R,M]^T B[R,N]          (same operand rules as dl_gemm_tn)
 *   dl_gemm_nt_f32_det  C[M,N] (f32)  = A[M,K] B[N,K]^T          (the plain f32-output form of dl_gemm_nt)
 *   dl_colsum_det       out[C] (f32) += sum_r x[r, :]            (dtype DL_BF16 / DL_F32 like dl_colsum) */
DL_API int dl_gemm_tn_det(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N,
                          int64_t R, float* scratch, int64_t scratch_floats, dl_stream_t stream);
DL_API int dl_gemm_nt_f32_det(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N,
                              int64_t K, float* scratch, int64_t scratch_floats, dl_stream_t stream);
DL_API int dl_colsum_det(const void* x, int dtype, int64_t ld, float* out, int64_t R, int64_t C, float* scratch,
                         int64_t scratch_floats, dl_stream_t stream);

/* ------------------------------------------------------------------ adaLN / norms */
/* modulate(LayerNorm(x)) mmdit.py:299,305,547 + nn.py:539:  out = (LN(x) * w + b) * (1 + scale) + shift.
 * w/b f32 [D] or NULL (norm_final has no affine); scale/shift bf16 rows with stride ld_mod, one row per
 * rows_per_mod tokens.  mean/rstd f32 [M] saved for the backward.
 * Optional fused gated residual of the PREVIOUS sub-layer (mmdit.py:302,308: x = x + gate * f(...)): when t != NULL the
 * row is first updated to x + gate[m / rows_per_mod, :] * t[m, :], written to x_out (bf16) and then normalised, so the
 * projection / MLP-down GEMMs stay plain stores and the residual stream is touched once per sub-layer. */
DL_API int dl_ln_modulate_fwd(const void* x, const float* w, const float* b, const void* scale, const void* shift,
                              int64_t ld_mod, int64_t rows_per_mod, float eps, void* out, float* mean,
                              float* rstd, const void* t, const void* gate, int64_t ld_gate, void* x_out, int64_t M,
                              int64_t D, dl_stream_t stream);
/* backward of the above, fused with the residual-stream add:
 *   dx[m,:]     = dres[m,:] + dLN(...)           (dres may be NULL)
 *   dscale[g,:] += sum_{m in g} dout * (xhat*w+b) ; dshift[g,:] += sum dout     (f32 rows, stride ld_dmod: the f32 image
 *                  of the modulation-gradient matrix, zeroed by the caller once per step, cast to bf16 once at the end)
 *   dwb_partial f32 [groups, 2, D] += per-group sums of dw, db (NULL when w == NULL); folded by dl_reduce_rows_f32.
 * The workgroups that share a sample meet in these accumulators through f32 atomics (no second pass).
 * Optional fused backward of the gated residual that follows in the backward chain (gate_t != NULL; the same arithmetic as
 * dl_gate_bwd on the dx just produced): dt[m,:] = gate[g,:] * dx[m,:] (bf16), dgate[g,:] += sum_{m in g} dx[m,:] * gate_t[m,:]
 * (f32, row stride ld_dmod). */
DL_API int dl_ln_modulate_bwd(const void* dout, const void* x, const float* w, const float* b, const void* scale,
                              int64_t ld_mod, int64_t rows_per_mod, const float* mean, const float* rstd,
                              const void* dres, void* dx, float* dscale, float* dshift, int64_t ld_dmod,
                              float* dwb_partial, const void* gate_t, const void* gate, int64_t ld_gate, void* dt,
                              float* dgate, int64_t M, int64_t D, dl_stream_t stream);
/* dl_ln_modulate_bwd for PER-TOKEN modulation (DDT decoder, ddt.py:423-424,455-457: Modulation on [B, S, D]): scale / gate have
 * one row per token (row stride ld_mod / ld_gate); dscale / dshift / dgate are bf16 rows WRITTEN per token (row stride ld_dmod:
 * windows of the [tokens, mod_rows] modulation-gradient matrix); dwb_partial f32 [n_part, 2, D] accumulates the affine
 * gradients (NULL with w == NULL), folded by dl_reduce_rows_f32. */
DL_API int dl_ln_modulate_bwd_tok(const void* dout, const void* x, const float* w, const float* b, const void* scale,
                                  int64_t ld_mod, const float* mean, const float* rstd, const void* dres, void* dx, void* dscale,
                                  void* dshift, int64_t ld_dmod, float* dwb_partial, int64_t n_part, const void* gate_t,
                                  const void* gate, int64_t ld_gate, void* dt, void* dgate, int64_t M, int64_t D,
                                  dl_stream_t stream);
/* ---- row-complete GEMMs (csrc/gemm_ln.hip): the LayerNorm-modulate / QK-norm row kernels as EPILOGUES of the GEMM that produces
 * their input.  On persistent 256 x 384 tiles a tile holds whole rows (D == 384) and one whole modulation group (rows_per_mod ==
 * 256), so row statistics and per-sample column sums complete on chip.  Every per-row output is bit-identical to the unfused launch
 * pair; the per-sample sums are plain stores in a fixed summation order (no atomics).  DL_ERR_UNSUPPORTED for any other shape
 * (the caller then issues dl_gemm_nt + the row kernel).
 *
 * dl_ln_modulate_gemm_fwd (mmdit.py:296-308 + nn.py:539; DiTBlock._forward's `x = x + gate * f(modulate(norm(x)))` chain):
 *   t = A[M,K] . W[D,K]^T (bf16) ; x' = resid + gate[m / rows_per_mod] * t (resid NULL: x' = t; gate NULL: x' = resid + t) ;
 *   xm = (LN(x') * ln_w + ln_b) * (1 + scale) + shift.  Writes t_out (may be NULL), x_out (x'), xm_out, mean, rstd.
 *   = dl_gemm_nt followed by dl_ln_modulate_fwd(t, gate, x_out). */
DL_API int dl_ln_modulate_gemm_fwd(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int64_t K, const void* resid,
                                   const void* gate, int64_t ld_gate, const float* ln_w, const float* ln_b, const void* scale,
                                   const void* shift, int64_t ld_mod, int64_t rows_per_mod, float eps, void* t_out, void* x_out,
                                   void* xm_out, float* mean, float* rstd, int64_t D, dl_stream_t stream);
/* dl_ln_modulate_gemm_bwd: dout = A[M,K] . Wt[D,K]^T (the data gradient of the linear that consumed the modulated rows, never
 * written) followed by exactly dl_ln_modulate_bwd(dout, x, ...) -- except that dscale / dshift / dgate / dwb_partial are WRITTEN
 * (=), not accumulated: one tile is one sample, so each of their elements has a single producer. */
DL_API int dl_ln_modulate_gemm_bwd(const void* A, int64_t lda, const void* Wt, int64_t ldw, int64_t M, int64_t K, const void* x,
                                   const float* ln_w, const float* ln_b, const void* scale, int64_t ld_mod, int64_t rows_per_mod,
                                   const float* mean, const float* rstd, const void* dres, void* dx, float* dscale, float* dshift,
                                   int64_t ld_dmod, float* dwb_partial, const void* gate_t, const void* gate, int64_t ld_gate,
                                   void* dt, float* dgate, int64_t D, dl_stream_t stream);
/* dl_gemm_nt_qk_norm_rope (DiTAttention.forward mmdit.py:81-91): qkv[M, 3D] = A[M, D] . Wqkv[3D, D]^T stored as bf16 rows, and in
 * the same launch the q and k thirds go through RMSNorm over the full D-wide row (nn.py:430), RoPE on the first `rot` channels of
 * every head (nn.py:345-352) and the head split: q, k [B, H, n_dst, dh] rows [n_off, n_off + N), rrms f32 [M, 2].
 * = dl_gemm_nt followed by dl_qk_norm_rope_fwd_ex(v = NULL). */
DL_API int dl_gemm_nt_qk_norm_rope(const void* A, int64_t lda, const void* Wqkv, int64_t ldw, int64_t B, int64_t N, int64_t H,
                                   int64_t dh, int64_t rot, float eps, const float* scale_q, const float* scale_k, const float* cos,
                                   const float* sin, void* qkv, void* q, void* k, float* rrms, int64_t n_dst, int64_t n_off,
                                   dl_stream_t stream);
/* DDT decoder conditioning (ddt.py:423-424 followed by the SiLU of Modulation / adaLN_modulation, nn.py:530, mmdit.py:543):
 * out[m, :] = silu(silu(enc[m, :] + temb[m / N, :])) (bf16 rows, the operand of the stacked per-token adaLN GEMM); backward:
 * denc = dout * d(silu o silu), dtemb[b, :] += sum over the N tokens of sample b (f32, atomically accumulated) */
DL_API int dl_ddt_cond_fwd(const void* enc, int64_t ld, const float* temb, int64_t ld_t, int64_t B, int64_t N, int64_t D, void* out,
                           dl_stream_t stream);
DL_API int dl_ddt_cond_bwd(const void* dsz, const void* enc, int64_t ld, const float* temb, int64_t ld_t, int64_t B, int64_t N,
                           int64_t D, void* denc, float* dtemb, dl_stream_t stream);
/* x_new = x + gate * t backward (mmdit.py:296-307): dt = gate * dout (bf16) ; dgate[g,:] = sum_{m in g} dout * t (f32,
 * written, row stride ld_dmod) */
DL_API int dl_gate_bwd(const void* dout, const void* t, const void* gate, int64_t ld_mod, int64_t rows_per_mod,
                       void* dt, float* dgate, int64_t ld_dmod, int64_t M, int64_t D, dl_stream_t stream);
/* QKNorm (nn.py:427-431,473-475: RMS over the FULL inner dim, eps 1e-6) + N-D RoPE on interleaved pairs
 * (nn.py:345-353,377-400) + head split 'b n (h d) -> b h n d' (mmdit.py:85-91).
 * qkv bf16 [B*N, 3D]; cos/sin f32 [N, rot/2]; q,k,v out bf16 [B,H,N,dh]; rrms f32 [B*N, 2] saved.
 * v (forward) / dv (backward) may be NULL: V is then neither copied nor its gradient written (see dl_attn_fwd_sv). */
DL_API int dl_qk_norm_rope_fwd(const void* qkv, const float* scale_q, const float* scale_k, const float* cos,
                               const float* sin, void* q, void* k, void* v, float* rrms, int64_t B, int64_t N,
                               int64_t H, int64_t dh, int64_t rot, float eps, dl_stream_t stream);
/* dscale f32 [2, D] is atomically accumulated (caller zeroes once per step). */
DL_API int dl_qk_norm_rope_bwd(const void* dq, const void* dk, const void* dv, const void* qkv,
                               const float* scale_q, const float* scale_k, const float* cos, const float* sin,
                               const float* rrms, void* dqkv, float* dscale, int64_t B, int64_t N, int64_t H,
                               int64_t dh, int64_t rot, dl_stream_t stream);
/* same with (a) an optional position index: pos int32 [B*N] gives the cos/sin table row of every token (NULL: row = n) -- SPRINT
 * runs its deep blocks on a per-sample subset of the image tokens whose RoPE rows are gathered with the kept indices
 * (sprint.py:348-353); the index replaces the gathered [B, k, rot/2] tables -- and (b) a row window: q, k, v are
 * [B, H, n_dst, dh] and the N tokens go to rows [n_off, n_off + N) -- the joint text-image attention concatenates the context
 * and image tokens along the sequence (mmdit.py:181-185); each stream writes its window of the joint buffers. */
DL_API int dl_qk_norm_rope_fwd_ex(const void* qkv, const float* scale_q, const float* scale_k, const float* cos,
                                  const float* sin, void* q, void* k, void* v, float* rrms, int64_t B, int64_t N,
                                  int64_t H, int64_t dh, int64_t rot, float eps, const int32_t* pos, int64_t n_dst,
                                  int64_t n_off, dl_stream_t stream);
DL_API int dl_qk_norm_rope_bwd_ex(const void* dq, const void* dk, const void* dv, const void* qkv,
                                  const float* scale_q, const float* scale_k, const float* cos, const float* sin,
                                  const float* rrms, void* dqkv, float* dscale, int64_t B, int64_t N, int64_t H,
                                  int64_t dh, int64_t rot, const int32_t* pos, int64_t n_dst, int64_t n_off,
                                  dl_stream_t stream);
/* backward of dl_qk_norm_rope_fwd IN PLACE on token-major gradient rows (nn.py:427-431, 345-353): on entry the q / k thirds of
 * dqkv [B*N, 3D] hold d(loss)/d(q after norm + RoPE) and d(loss)/d(k ...) in token-major order (dl_attn_bwd_tok writes them
 * there); on return they hold the gradient of the pre-norm q / k (the v third is untouched).  The scale gradients are summed
 * WITHOUT atomics: every workgroup stores its [2, D] partial into dscale_partials (f32 scratch, >= 1024 * 2 * D floats) and one
 * fold adds them to dscale f32 [2, D] in a fixed order (bit-reproducible).  pos: optional RoPE table row per token (NULL: n). */
DL_API int dl_qk_norm_rope_bwd_inplace(const void* qkv, const float* scale_q, const float* scale_k, const float* cos,
                                       const float* sin, const float* rrms, void* dqkv, float* dscale, float* dscale_partials,
                                       int64_t B, int64_t N, int64_t H, int64_t dh, int64_t rot, const int32_t* pos,
                                       dl_stream_t stream);
/* F.scaled_dot_product_attention mmdit.py:92-100, no mask: out = softmax(q k^T * scale) v, written as
 * 'b h n d -> b n (h d)'.  lse f32 [B,H,N] = natural-log-sum-exp of the scaled scores (for the backward).
 * dh must be 64; N a multiple of 64 up to 256 (K and V of one head stay resident in LDS), or a multiple of 256 up to 2048
 * (one workgroup per 256-row chunk, the other operand streamed through LDS in 256-row chunks). */
DL_API int dl_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int64_t B, int64_t H,
                       int64_t N, int64_t dh, float scale, dl_stream_t stream);
DL_API int dl_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                       const float* lse, void* dq, void* dk, void* dv, int64_t B, int64_t H, int64_t N,
                       int64_t dh, float scale, dl_stream_t stream);
/* general form (joint text-image attention mmdit.py:172-190 with its key-padding mask, cross-attention of a resampler):
 * q [B,H,Nq,64] against k, v [B,H,Nk,64]; Nq, Nk multiples of 256 up to 2048 (pad); key_bias f32 [B, Nk] is ADDED to the
 * scaled scores (0 = attend, -inf = masked / padded key) or NULL; out [B, Nq, H*64], lse [B,H,Nq]. */
DL_API int dl_attn_fwd_ex(const void* q, const void* k, const void* v, void* out, float* lse, int64_t B, int64_t H,
                          int64_t Nq, int64_t Nk, int64_t dh, float scale, const float* key_bias, dl_stream_t stream);
DL_API int dl_attn_bwd_ex(const void* q, const void* k, const void* v, const void* out, const void* dout,
                          const float* lse, void* dq, void* dk, void* dv, int64_t B, int64_t H, int64_t Nq, int64_t Nk,
                          int64_t dh, float scale, const float* key_bias, dl_stream_t stream);
/* the N <= 256 kernels with V (and dV) addressed in place: head (b, h) of V starts at element b*v_batch_stride +
 * h*v_head_stride of `v`, its rows are v_pitch elements apart.  With v = qkv + 2*D, strides {N*3D, 64, 3D} the attention
 * reads the v third of the token-major qkv rows [B*N, 3D] (the reference's `qkv.chunk(3)`, mmdit.py:85-93) and the backward
 * writes dV straight into the v third of dqkv, so dl_qk_norm_rope_{fwd,bwd} are called with v / dv = NULL and never copy V. */
DL_API int dl_attn_fwd_sv(const void* q, const void* k, const void* v, int64_t v_batch_stride, int64_t v_head_stride,
                          int64_t v_pitch, void* out, float* lse, int64_t B, int64_t H, int64_t N, int64_t dh, float scale,
                          dl_stream_t stream);
DL_API int dl_attn_bwd_sv(const void* q, const void* k, const void* v, int64_t v_batch_stride, int64_t v_head_stride,
                          int64_t v_pitch, const void* out, const void* dout, const float* lse, void* dq, void* dk, void* dv,
                          int64_t dv_batch_stride, int64_t dv_head_stride, int64_t dv_pitch, int64_t B, int64_t H, int64_t N,
                          int64_t dh, float scale, dl_stream_t stream);
/* dl_attn_bwd_sv whose dQ and dK leave in place as well (N <= 256): V is read from the v third of the token-major qkv rows
 * [B*N, 3D] and ALL THREE gradients are written into the token-major dqkv rows (dq -> columns [0, D), dk -> [D, 2D), dv ->
 * [2D, 3D)); dl_qk_norm_rope_bwd_inplace then turns the q / k thirds into the gradient of the pre-norm qkv in place -- no
 * head-major dq / dk buffers, and the QK-norm backward reads whole 2D-wide rows instead of H 128-byte segments per row. */
DL_API int dl_attn_bwd_tok(const void* q, const void* k, const void* qkv, const void* out, const void* dout, const float* lse,
                           void* dqkv, int64_t B, int64_t H, int64_t N, int64_t dh, float scale, dl_stream_t stream);
/* ---- QK-RMSNorm + RoPE without a pass of its own (round 4).  dl_gemm_nt_ssq: the plain bf16 product C = A B^T on the persistent
 * 256 x 384 tiles (the qkv GEMM, mmdit.py:81) whose epilogue also adds the per-row sums of squares of the rounded outputs of the first
 * `ssq_tiles` 384-wide column tiles into ssq f32 [M, ssq_tiles] (the caller zeroes it; two addends per element: bit-reproducible) --
 * the statistics of RMSNorm over the full D-wide q / k row (nn.py:427-431).  M % 256 == 0, N % 384 == 0, >= 64 tiles, else
 * DL_ERR_UNSUPPORTED.  dl_attn_fwd_qkn: DiTAttention.forward mmdit.py:81-100 from the PRE-NORM token-major qkv rows [B*N, 3D]: q and k
 * are normalised with r = rsqrt(ssq / D + eps), scaled and rotated (nn.py:345-353) as they are staged (K in LDS, Q in registers),
 * then softmax(q k^T scale) v as dl_attn_fwd_sv; the normalised q, k are also written head-major [B, H, N, 64] and r as rrms [B*N, 2]
 * (inputs of the backward kernels; all three NULL = inference, nothing of it is written).  N % 64 == 0 up to 256, dh = 64. */
DL_API int dl_gemm_nt_ssq(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                          int64_t K, float* ssq, int64_t ssq_tiles, dl_stream_t stream);
DL_API int dl_attn_fwd_qkn(const void* qkv, const float* ssq, const float* scale_q, const float* scale_k, const float* cos,
                           const float* sin, float eps, int64_t rot, void* q_out, void* k_out, float* rrms, void* out, float* lse,
                           int64_t B, int64_t H, int64_t N, int64_t dh, float scale, dl_stream_t stream);
/* PackedSwiGLU nn.py:484-486: h = silu(u[:, :F]) * u[:, F:] ; u bf16 [M, 2F] */
DL_API int dl_swiglu_fwd(const void* u, void* h, int64_t M, int64_t F, dl_stream_t stream);
DL_API int dl_swiglu_bwd(const void* dh, const void* u, void* du, int64_t M, int64_t F, dl_stream_t stream);

/* ------------------------------------------------------------------ stem / head / conditioning */
/* im2row for conv_proj (mmdit.py:757-765, order DL_PATCH_CPP) and the transpose of unpatchify
 * (mmdit.py:778-787, order DL_PATCH_PPC): x f32 [B,C,H,W] -> tok bf16 [B*gh*gw, ld] (cols >= C*p*p zeroed) */
DL_API int dl_patchify(const float* x, void* tok, int64_t B, int64_t C, int64_t H, int64_t W, int64_t p,
                       int64_t ld, int order, dl_stream_t stream);
/* unpatchify mmdit.py:778-787: tok f32 [B*gh*gw, ld] (p1 p2 c) -> img f32 [B,C,H,W] */
DL_API int dl_unpatchify(const float* tok, float* img, int64_t B, int64_t C, int64_t H, int64_t W, int64_t p,
                         int64_t ld, dl_stream_t stream);
/* timestep_embedding nn.py:106-114: out[b,:] = [cos(t f_i) | sin(t f_i)], bf16 [B, dim] (dim even) */
DL_API int dl_timestep_embedding(const float* t, void* out, int64_t B, int64_t dim, float max_period,
                                 dl_stream_t stream);
/* emb = e + table[idx] (mmdit.py:867-868, nn.py:162-163; idx NULL -> no label term), act = silu(emb) as
 * bf16 (input of every Modulation / adaLN linear, nn.py:531, mmdit.py:540) */
DL_API int dl_cond_combine_fwd(const float* e, const float* table, const int64_t* idx, float* emb, void* act,
                               int64_t B, int64_t E, dl_stream_t stream);
/* demb = dact * silu'(emb) (f32 + bf16 copies); dtable[idx[b],:] += demb[b,:] (atomic; dtable may be NULL) */
DL_API int dl_cond_combine_bwd(const float* dact, const float* emb, const int64_t* idx, float* demb,
                               void* demb_bf16, float* dtable, int64_t B, int64_t E, dl_stream_t stream);
/* dx = dy * silu'(pre) ; pre bf16 (saved pre-activation), dy f32, dx bf16 */
DL_API int dl_silu_bwd(const float* dy, const void* pre, void* dx, int64_t n, dl_stream_t stream);
/* dx = dy * gelu'(pre)  (exact erf GELU; FeedForward of the Perceiver resampler, perceiver_resampler.py:77-80) */
DL_API int dl_gelu_bwd(const float* dy, const void* pre, void* dx, int64_t n, dl_stream_t stream);
/* 'b n (h d) -> b h n d' into a longer sequence, with optional rotary embedding (interleaved pairs, nn.py:345-353) on the first
 * `rot` channels of every head: dst[b, h, n_off + n, :] = rope(src[b*n_src + n, h*64 : h*64+64]); src bf16 rows with stride ld,
 * dst bf16 [B, H, n_dst, 64]; cos/sin f32 [n_src, rot/2] or NULL (plain head split).  The backward is the same call pattern
 * in the other direction: src[b*n_src + n, h*64+d] = rope^T(dst[b, h, n_off + n, d]) (accumulate != 0: += into src). */
DL_API int dl_heads_split_rope(const void* src, int64_t ld, void* dst, int64_t B, int64_t H, int64_t n_src, int64_t n_dst,
                               int64_t n_off, const float* cos, const float* sin, int64_t rot, dl_stream_t stream);
DL_API int dl_heads_merge_rope_bwd(const void* dst_grad, void* src_grad, int64_t ld, int64_t B, int64_t H, int64_t n_src,
                                   int64_t n_dst, int64_t n_off, const float* cos, const float* sin, int64_t rot,
                                   int accumulate, dl_stream_t stream);
/* out[c] += sum_r x[r,c]  (bias gradients); x bf16 or f32 per dtype */
DL_API int dl_colsum(const void* x, int dtype, int64_t ld, float* out, int64_t R, int64_t C,
                     dl_stream_t stream);
/* several bf16 column sums in one launch per 24 problems: the bias gradients of the nn.Conv2d / conv_nd(1, ...) layers of a stretch of the
 * UNet's backward (unet.py:187,208,302-305,594,745: what autograd computes as dY.sum over every pixel); `desc` is a HOST array (the
 * descriptors travel by value in the kernel arguments: the operands are activation gradients whose addresses change per step) */
typedef struct dl_colsum_desc_t {
  const void* x; /* bf16 [R, ld] */
  int64_t ld;
  float* out;    /* f32 [C], accumulated (+=) */
  int64_t R, C;
} dl_colsum_desc_t;
DL_API int dl_colsum_batched(const dl_colsum_desc_t* desc, int n, dl_stream_t stream);
/* out[j] += sum_g partial[g, j]   (second stage of the LayerNorm affine gradients); clear_partial != 0 zeroes `partial`
 * as it is read, so accumulate-into partial buffers need no memset */
DL_API int dl_reduce_rows_f32(float* partial, float* out, int64_t G, int64_t n, int clear_partial, dl_stream_t stream);
/* K such folds in one launch, deterministic (one writer per element, fixed order):
 * out[k * out_stride + j] += sum_{g < G} partial[k * partial_stride + g * n + j]  for k < K, j < n (strides in elements) */
DL_API int dl_reduce_rows_batched_f32(const float* partial, int64_t partial_stride, float* out, int64_t out_stride, int64_t K,
                                      int64_t G, int64_t n, dl_stream_t stream);

/* RePA alignment loss (training/losses/repa.py:196-198): row-wise F.cosine_similarity(p, d, dim=-1, eps) between the projected
 * denoiser features p (bf16 [M, E]) and the target encoder features d (f32 [M, E]); |p|^2 and |d|^2 are kept for the backward.
 * backward: dp = gscale * (*gscale_dev) * d cos / d p  (bf16). */
DL_API int dl_cosine_rows_fwd(const void* p, int64_t ldp, const float* d, int64_t ldd, float* cosv, float* pn2, float* dn2,
                              int64_t M, int64_t E, float eps, dl_stream_t stream);
DL_API int dl_cosine_rows_bwd(const void* p, int64_t ldp, const float* d, int64_t ldd, const float* cosv, const float* pn2,
                              const float* dn2, float gscale, const float* gscale_dev, void* dp, int64_t lddp, int64_t M,
                              int64_t E, float eps, dl_stream_t stream);

/* ------------------------------------------------------------------ SPRINT token routing (sprint.py:317-387) */
/* drop_tokens' torch.gather (sprint.py:347): dst[b, j, :] = src[b, idx[b, j], :]; src rows [B*N] with stride ld_src, dst rows
 * [B*k] with stride ld_dst, idx int32 [B, k].  keep int32 [B] or NULL: samples with keep[b] == 0 produce zero rows (the
 * backward of restore_tokens for samples whose deep path was dropped). */
DL_API int dl_gather_tokens(const void* src, int64_t ld_src, const int32_t* idx, const int32_t* keep, void* dst,
                            int64_t ld_dst, int64_t B, int64_t N, int64_t k, int64_t D, dl_stream_t stream);
/* adjoint of the gather: dst[b, idx[b, j], :] += src[b, j, :] (indices of one sample are distinct) */
DL_API int dl_scatter_tokens_add(const void* src, int64_t ld_src, const int32_t* idx, void* dst, int64_t ld_dst, int64_t B,
                                 int64_t N, int64_t k, int64_t D, dl_stream_t stream);
/* restore_tokens (sprint.py:355-387): out[b, n, :] = inv[b, n] >= 0 ? xd[b, inv[b, n], :] : mask_token; inv int32 [B, N] is the
 * inverse of the kept indices (-1 = dropped token, or every token of a sample whose deep path is dropped); mask f32 [D]. */
DL_API int dl_restore_tokens(const void* xd, int64_t ld_xd, const int32_t* inv, const float* mask, void* out, int64_t ld_out,
                             int64_t B, int64_t N, int64_t k, int64_t D, dl_stream_t stream);
/* gradient of the mask token: out[c] += sum over rows with sel[row] < 0 of x[row, c] (x bf16, out f32) */
DL_API int dl_masked_colsum(const void* x, int64_t ld, const int32_t* sel, float* out, int64_t R, int64_t C,
                            dl_stream_t stream);
/* strided row-block copy: dst[b, r, 0:cols] = src[b, r, 0:cols] for b < B, r < rows, with independent batch / row strides
 * (elements) on both sides -- slices the [context ; image] halves out of / into the joint attention buffers (mmdit.py:207-208) */
DL_API int dl_copy_rows3d(const void* src, int64_t src_bs, int64_t src_rs, void* dst, int64_t dst_bs, int64_t dst_rs, int64_t B,
                          int64_t rows, int64_t cols, dl_stream_t stream);
/* x + gate * t materialised (mmdit.py:302,308 at a stage boundary, where no LayerNorm follows to absorb it); gate rows as in
 * dl_ln_modulate_fwd */
DL_API int dl_gated_residual_fwd(const void* x, const void* t, const void* gate, int64_t ld_gate, int64_t rows_per_mod,
                                 void* out, int64_t ld_out, int64_t M, int64_t D, dl_stream_t stream);

/* ------------------------------------------------------------------ optimizer side */
/* torch.optim.AdamW single-tensor math (configs/optimizer/adamw.yaml) over a flat f32 buffer:
 *   p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr/bc1) m / (sqrt(v)/sqrt(bc2) + eps) */
DL_API int dl_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                         float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2,
                         float grad_scale, dl_stream_t stream);
/* the same update with its eight scalars in device memory: hyper = {lr, beta1, beta2, eps, weight_decay, lr / bias_corr1,
 * 1 / sqrt(bias_corr2), grad_scale}.  No per-step value travels in the launch, so a captured hipGraph of the whole training step
 * (base_trainer.py:138-151: zero_grad -> loss -> backward -> optimizer.step) replays with the host refreshing 32 bytes */
DL_API int dl_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, dl_stream_t stream);
/* bf16 shadow of an f32 [R,C] weight: dst [R, ld_dst] (cols >= C zeroed up to ld_dst) and/or the transpose
 * dstT [C, ld_t] (cols >= R zeroed up to ld_t).  Either destination may be NULL. */
DL_API int dl_cast_weight(const float* src, int64_t R, int64_t C, void* dst, int64_t ld_dst, void* dstT,
                          int64_t ld_t, dl_stream_t stream);
/* bf16 shadow of the packed-SwiGLU weight [2F, C] with the row order dl_gemm_nt_swiglu expects: inside every group of
 * 32 output rows, rows 0..15 are x1 rows 16q..16q+15 and rows 16..31 are x3 rows 16q..16q+15 (q = group index; F % 16 == 0). */
DL_API int dl_cast_weight_swiglu(const float* src, int64_t F, int64_t C, void* dst, int64_t ld_dst, dl_stream_t stream);
/* all shadows of a network in one launch: `desc_dev` is a DEVICE array of n_desc descriptors ordered by tile_begin (tiles
 * are 32x32 blocks over max(R, ld_t) x max(C, ld_dst, ld_swiglu), tiles_c per row of blocks; total_tiles = sum).  dst /
 * dst_t as in dl_cast_weight, dst_swiglu as in dl_cast_weight_swiglu (R = 2F); any destination may be NULL. */
typedef struct {
  const void* src; /* f32 [R, C] */
  int64_t R, C;
  void* dst;
  int64_t ld_dst;
  void* dst_t;
  int64_t ld_t;
  void* dst_swiglu;
  int64_t ld_swiglu;
  int64_t tile_begin, tiles_c;
} dl_cast_desc_t;
DL_API int dl_cast_weights_batched(const dl_cast_desc_t* desc_dev, int n_desc, int64_t total_tiles, dl_stream_t stream);
/* plain casts */
DL_API int dl_cast_f32_to_bf16(const float* src, void* dst, int64_t n, dl_stream_t stream);
DL_API int dl_cast_bf16_to_f32(const void* src, float* dst, int64_t n, dl_stream_t stream);
/* the same over a column window of wider rows: dst[r, c] = bf16(src[r, c]), r < rows, c < cols (cols and both row strides
 * multiples of 4 elements).  The data-parallel DiT backward casts each block's [B, 6D] slice of the f32 modulation-gradient
 * accumulator as soon as that block is done, so that the block's adaLN weight gradient can be reduced early. */
DL_API int dl_cast2d_f32_to_bf16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int64_t cols,
                                 dl_stream_t stream);

/* ema = ema + (1 - beta) (p - ema)  == lerp used by ema_pytorch (base_trainer.py:152-153) */
DL_API int dl_ema_update(float* ema, const float* p, float beta, int64_t n, dl_stream_t stream);

/* ------------------------------------------------------------------ UNet (networks/denoisers/unet.py, networks/utils/nn.py)
 * Activations inside the library are NHWC bf16 token rows [B*H*W, C]; a 3x3 convolution (nn.Conv2d padding=1, unet.py:187,
 * 208,594,745) is dl_im2col3x3 + dl_gemm_nt against the reordered weight shadow; 1x1 / Conv1d(k=1) are plain dl_gemm_nt. */
/* reference boundary tensors are NCHW f32 (unet.py:832): x [B,C,HW] -> out bf16 [B*HW, ld] (columns >= C untouched) */
DL_API int dl_nchw_to_nhwc(const float* x, void* out, int64_t B, int64_t C, int64_t HW, int64_t ld, dl_stream_t stream);
/* x bf16 [B*HW, ld] (first C columns) -> out f32 [B,C,HW] */
DL_API int dl_nhwc_to_nchw(const void* x, float* out, int64_t B, int64_t C, int64_t HW, int64_t ld, dl_stream_t stream);
/* GroupNorm32 statistics (nn.py:11-13: fp32, per sample and group over C/G channels x HW): stats f32 [B, G, 2] = mean, rstd */
DL_API int dl_gn_stats(const void* x, float* stats, int64_t B, int64_t HW, int64_t C, int64_t G, float eps,
                       dl_stream_t stream);
/* out = act( (xhat*w + b) * (1 + film_scale[b,c]) + film_shift[b,c] ) ; film_* bf16 [B, ld_film] or NULL (unet.py:215-237:
 * in_layers GN+SiLU has no FiLM, out_layers GN * (1+scale) + shift then SiLU); act_silu = 0 gives the plain GroupNorm of
 * AttentionBlock.norm_x / norm_context (unet.py:296-322) */
DL_API int dl_gn_apply_fwd(const void* x, const float* stats, const float* w, const float* b, const void* film_scale,
                           const void* film_shift, int64_t ld_film, int act_silu, void* out, int64_t B, int64_t HW,
                           int64_t C, int64_t G, dl_stream_t stream);
/* GroupNorm32 (nn.py:11-13) + FiLM + SiLU (unet.py:215-237) with the statistics taken in the same call: one launch on the training
 * shapes (a workgroup owns whole groups of a sample, its rows stay in registers between the two passes), dl_gn_stats + dl_gn_apply_fwd
 * otherwise; `stats` f32 [B, G, 2] is written for dl_gn_bwd */
DL_API int dl_gn_fwd(const void* x, const float* w, const float* b, const void* film_scale, const void* film_shift, int64_t ld_film,
                     int act_silu, void* out, float* stats, int64_t B, int64_t HW, int64_t C, int64_t G, float eps, dl_stream_t stream);
/* backward of dl_gn_stats + dl_gn_apply_fwd: dx bf16 (= gradient through the norm + dres when dres != NULL: the residual /
 * skip fan-in of ResBlock and AttentionBlock), dw/db f32 [C] ACCUMULATED (+=), dfilm_* bf16 [B, ld_dfilm] written
 * (required iff film_* given).  scratch: f32 [DL_GN_BWD_MAX_RANGES * B*4*C + B*G*2] (partial sums of the pixel ranges the
 * reduction is split into, then the per-group sums) */
#define DL_GN_BWD_MAX_RANGES 8
DL_API int dl_gn_bwd(const void* dout, const void* x, const float* stats, const float* w, const float* b,
                     const void* film_scale, const void* film_shift, int64_t ld_film, int act_silu, const void* dres,
                     void* dx, float* dw, float* db, void* dfilm_scale, void* dfilm_shift, int64_t ld_dfilm, float* scratch, int64_t B,
                     int64_t HW, int64_t C, int64_t G, dl_stream_t stream);
/* x bf16 [B*H*W, ldx]; cols bf16 [rows, ld]: cols[r, (ky*3+kx)*C + c] = x[b, y+ky-1, x+kx-1, c] (zero outside the image); rows >= B*H*W and
 * columns >= 9*C are zero-filled so the buffer can be handed to the 64-aligned GEMMs as is */
DL_API int dl_im2col3x3(const void* x, int64_t ldx, void* cols, int64_t B, int64_t H, int64_t W, int64_t C, int64_t rows,
                        int64_t ld, dl_stream_t stream);
/* Conv2d weight f32 [Co, Ci, 3, 3] -> forward shadow wf bf16 [Co, ldf] (k = tap*Ci + ci) and data-gradient shadow wd bf16
 * [Ci, ldd] (k = tap*Co + co, kernel rotated by 180 degrees) */
DL_API int dl_cast_conv3x3_weight(const float* w, int64_t Co, int64_t Ci, void* wf, int64_t ldf, void* wd, int64_t ldd,
                                  dl_stream_t stream);
/* the shadows of EVERY 3x3 convolution weight of a network in one launch: `desc_dev` is a DEVICE array of n_desc descriptors ordered by
 * tile_begin; a tile is a 32 x 32 (output x input channel) block of one weight, tiles of a weight numbered co-tile major
 * ((Co / 32) * (Ci / 32) of them; total_tiles = sum).  Only weights with Co % 32 == 0, Ci % 32 == 0 and unpadded shadows (ldf == 9 Ci,
 * ldd == 9 Co) belong in the table (the others keep dl_cast_conv3x3_weight); wf / wd may be NULL. */
typedef struct {
  const void* w; /* f32 [Co, Ci, 3, 3] */
  int64_t Co, Ci;
  void* wf;      /* bf16 [Co, ldf]: k = tap * Ci + ci */
  int64_t ldf;
  void* wd;      /* bf16 [Ci, ldd]: k = (8 - tap) * Co + co */
  int64_t ldd;
  int64_t tile_begin;
} dl_cast_conv_desc_t;
DL_API int dl_cast_conv3x3_weights_batched(const dl_cast_conv_desc_t* desc_dev, int n_desc, int64_t total_tiles, dl_stream_t stream);
/* implicit-GEMM 3x3 / pad-1 convolution (no cols matrix: the GEMM's operand loads gather the taps, out-of-image taps read
 * the caller's `zero` line of >= 16 zero bytes):  out[p, co] = bias[co] + sum_{tap,ci} x[p + shift(tap), ci] Wf[co, tap*Ci+ci]
 * (+ resid[p, co]).  Wf = forward shadow of dl_cast_conv3x3_weight; with the rotated shadow and x = dY it is the data
 * gradient.  splitk_scratch: optional caller workspace of `scratch_floats` floats; when given, launches with few output tiles and a
 * deep contraction (the low-resolution UNet levels) split K across workgroups into min(8, scratch_floats / (B*H*W*Co)) partial images
 * and finish in a second pass that adds them in a fixed order (no atomics); a scratch smaller than two images is not used.
 * Returns DL_ERR_UNSUPPORTED when Ci % 64 != 0 (caller then uses dl_im2col3x3 + dl_gemm_nt). */
DL_API int dl_conv3x3_nt(const void* x, int64_t ldx, int64_t B, int64_t H, int64_t W, int64_t Ci, const void* Wf,
                         int64_t ldw, void* out, int64_t ldc, int64_t Co, const float* bias, const void* resid,
                         int64_t ldr, const void* zero, float* splitk_scratch, int64_t scratch_floats, dl_stream_t stream);
/* implicit-GEMM weight gradient, transposed like dl_conv3x3_wgrad_fold expects:
 * g[(tap, ci), co] += sum_p x[p + shift(tap), ci] dY[p, co]; dY rows [R, ldy], R % 64 == 0, rows >= B*H*W zero; Co % 8 == 0.
 * max_workgroups caps the persistent workgroups like dl_gemm_tn_ex (0 = one per CU).  Returns DL_ERR_UNSUPPORTED when Ci % 128 != 0. */
DL_API int dl_conv3x3_wgrad_tn(const void* x, int64_t ldx, int64_t B, int64_t H, int64_t W, int64_t Ci, const void* dY,
                               int64_t ldy, int64_t R, int64_t Co, float* g, int64_t ldg, const void* zero,
                               int max_workgroups, dl_stream_t stream);
/* the same product without atomics (round 6): the R-splits store partial images g + s * part_stride (f32 [9*Ci, ldg] each, written
 * in full with plain stores: nothing is read, nothing needs zeroing) for dl_conv3x3_wgrad_fold_batched (n_img / img_stride of its
 * descriptor) to add in a fixed order -- bit-reproducible.  dl_conv3x3_wgrad_tn_nparts = the number of images the shape and map
 * produce on this device with this workgroup cap (0: unsupported, Ci % 128 != 0); max_parts = the images the caller's buffer holds. */
DL_API int dl_conv3x3_wgrad_tn_nparts(int64_t H, int64_t W, int64_t Ci, int64_t Co, int64_t R, int max_workgroups);
DL_API int dl_conv3x3_wgrad_tn_parts(const void* x, int64_t ldx, int64_t B, int64_t H, int64_t W, int64_t Ci, const void* dY,
                                     int64_t ldy, int64_t R, int64_t Co, float* g, int64_t ldg, int64_t part_stride,
                                     int64_t max_parts, const void* zero, int max_workgroups, dl_stream_t stream);
/* weight gradient from dl_gemm_tn(cols, dY) lands transposed as g f32 [(tap, ci), ldg >= Co]: dw[Co, Ci, 3, 3] += g^T */
DL_API int dl_conv3x3_wgrad_fold(const float* g, int64_t ldg, float* dw, int64_t Co, int64_t Ci, dl_stream_t stream);
/* The same fold for every convolution of a network in ONE launch (end of the backward): `desc_dev` = device array of n_desc
 * descriptors, tile_begin = running sum of the entries' workgroup counts: (Co/32)*(Ci/32) (Co, Ci multiples of 32), times 9 for an
 * entry of DL_FOLD_TAP_SPLIT_MIN_IMAGES or more partial images (one tap of a channel tile per workgroup; such an entry needs
 * ldg % 4 == 0, img_stride % 4 == 0 and a 16-byte aligned g); total_tiles = the sum over all entries.
 * clear_stage != 0 zeroes each staging tile as it is read (persistent staging buffers need no memset before the next backward). */
#define DL_FOLD_TAP_SPLIT_MIN_IMAGES 8
typedef struct dl_fold_conv_desc_t {
  void* g;       /* f32 [9*Ci, ldg]: the staged transposed gradient, row = tap * Ci + ci */
  int64_t ldg;
  void* dw;      /* f32 [Co, Ci, 3, 3]: accumulated (+=) */
  int64_t Co, Ci;
  int64_t tile_begin;
  int64_t n_img;      /* 0: one accumulated image (the atomic form of dl_conv3x3_wgrad_tn; clear_stage applies); >= 1: that many partial */
  int64_t img_stride; /* images of dl_conv3x3_wgrad_tn_parts, img_stride floats apart, added in image order (never cleared)      */
} dl_fold_conv_desc_t;
DL_API int dl_conv3x3_wgrad_fold_batched(const dl_fold_conv_desc_t* desc_dev, int n_desc, int64_t total_tiles, int clear_stage,
                                         dl_stream_t stream);
/* out[b, yo, xo, c] = scale * sum of the 2x2 window of x [B, 2Ho, 2Wo, C]: avg_pool2d forward (scale 0.25, nn.py:86) and
 * nearest-upsample backward (scale 1) */
DL_API int dl_reduce2x2(const void* x, void* out, int64_t B, int64_t Ho, int64_t Wo, int64_t C, float scale,
                        dl_stream_t stream);
/* out[b, y, x, c] = scale * x[b, y/2, x/2, c] for out [B, 2Hi, 2Wi, C]: nearest upsample forward (scale 1, nn.py:50) and
 * avg_pool2d backward (scale 0.25) */
DL_API int dl_expand2x2(const void* x, void* out, int64_t B, int64_t Hi, int64_t Wi, int64_t C, float scale,
                        dl_stream_t stream);
/* dl_reduce2x2 (expand = 0) / dl_expand2x2 (expand != 0) over TWO tensors of the same geometry in one launch (x1 / out1 NULL: one) --
 * a resampling ResBlock moves its h and its x through the same Upsample / Downsample (unet.py:196-203,226-230; nn.py:28-88);
 * Hs x Ws = the small map.  16 bytes per lane when C % 8 == 0. */
DL_API int dl_resample2x2_pair(const void* x0, void* out0, const void* x1, void* out1, int64_t B, int64_t Hs, int64_t Ws, int64_t C,
                               float scale, int expand, dl_stream_t stream);
/* out[b, yo, xo, c] = x[b, 2yo, 2xo, c] for x [B, 2Ho, 2Wo, C], C % 8 == 0: with dl_conv3x3_nt at full resolution in front this
 * is Downsample's 3x3 stride-2 pad-1 convolution (nn.py:79) */
DL_API int dl_pick2x2(const void* x, void* out, int64_t B, int64_t Ho, int64_t Wo, int64_t C, dl_stream_t stream);
/* out[b, y, x, c] = dy[b, y/2, x/2, c] where y and x are both even, 0 elsewhere (out [B, 2Hi, 2Wi, C], C % 8 == 0): the
 * backward of dl_pick2x2; the stride-1 data / weight gradients of the conv follow */
DL_API int dl_stuff2x2(const void* dy, void* out, int64_t B, int64_t Hi, int64_t Wi, int64_t C, dl_stream_t stream);
/* additive ResBlock conditioning (use_scale_shift_norm=False, unet.py:235-237 `h = h + emb_out`):
 * out[b, p, c] = x[b, p, c] + e[b, c] over x bf16 [B, HW, C], e bf16 rows of leading dimension lde (C, lde % 8 == 0) */
DL_API int dl_rowbias_add(const void* x, const void* e, int64_t lde, void* out, int64_t B, int64_t HW, int64_t C,
                          dl_stream_t stream);
/* its backward for e: de[b, c] = sum_p dy[b, p, c] (f32 accumulation, bf16 store into rows of leading dimension lde) */
DL_API int dl_rowbias_bwd(const void* dy, void* de, int64_t lde, int64_t B, int64_t HW, int64_t C, dl_stream_t stream);
/* AttentionBlock core (unet.py:311-318: heads split the channel dim as (h d), scale = dh^-0.5): q/k/v/out are token rows
 * with head h at columns [h*dh, (h+1)*dh); n <= 64 tokens, dh % 8 == 0; probs f32 [B, H, n, n] is kept for the backward */
DL_API int dl_attn_small_fwd(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldkv, void* out,
                             int64_t ldo, float* probs, int64_t B, int64_t n, int64_t H, int64_t dh, float scale,
                             dl_stream_t stream);
DL_API int dl_attn_small_bwd(const void* q, const void* k, const void* v, int64_t ldq, int64_t ldkv, const void* dout,
                             int64_t ldo, const float* probs, void* dq, void* dk, void* dv, int64_t B, int64_t n,
                             int64_t H, int64_t dh, float scale, dl_stream_t stream);
/* out = a + b (bf16): gradient fan-in of the UNet skip connections (unet.py:846-851: hs.append / torch.cat) */
DL_API int dl_add_bf16(const void* a, const void* b, void* out, int64_t n, dl_stream_t stream);
/* dst[r, c] = src[r, c] for a [rows, cols] window of two pitched bf16 matrices: channel concat / split of the skip
 * connections (torch.cat(dim=1) in NCHW == column blocks in NHWC) and zero-padding to the GEMM alignment */
DL_API int dl_copy2d_bf16(const void* src, int64_t lds, void* dst, int64_t ldd, int64_t rows, int64_t cols,
                          dl_stream_t stream);

/* ------------------------------------------------------------------ fp32-class regime (csrc/f32.hip)
 * The reference's DEFAULT precision (`precision_type="no"`: training/trainers/common.py:76,105, configs/trainer/default.yaml:4):
 * f32 activations, f32 GEMM operands straight from the parameter arena (no bf16 shadows), f32 accumulation.  Every product of the
 * step runs on the exact-f32 matrix instruction v_mfma_f32_32x32x2_f32 through ONE strided, batched entry point: */
typedef struct dl_f32_gemm_t {
  const float* A;          /* trans_a == 0: [M, K] row-major (K contiguous, row stride lda); != 0: [K, M] (M contiguous) */
  const float* B;          /* trans_b == 0: [N, K] row-major (torch Linear weight layout); != 0: [K, N] (N contiguous) */
  float* C;                /* [M, N], row stride ldc */
  int64_t lda, ldb, ldc;
  int64_t M, N, K;
  int32_t trans_a, trans_b;
  /* two-level batch: problem (i, j), i < batch1, j < batch2, starts at X + i * stride_x1 + j * stride_x2 (elements).  The attention
   * matmuls use (sample, head) with the head stride 64 inside token-major rows: no head split / transpose pass (mmdit.py:85-100) */
  int64_t batch1, batch2;
  int64_t stride_a1, stride_a2, stride_b1, stride_b2, stride_c1, stride_c2;
  float alpha;             /* C = act(alpha * A.B + bias) [+ C when accumulate] */
  const float* bias;       /* f32 [N] or NULL */
  int32_t act;             /* DL_ACT_NONE | DL_ACT_SILU */
  float* pre_out;          /* optional: alpha * A.B + bias before the activation (layout of C), saved for the backward */
  int32_t accumulate;      /* != 0: C += (parameter gradients accumulate into the arena) */
  float* scratch;          /* optional caller-owned f32 scratch: unbatched products with few output tiles and a long contraction
                            * (weight gradients over all tokens) split K over workgroups, every split stores its partial image and
                            * a fold adds them in a fixed order (no atomics, bit-reproducible) */
  int64_t scratch_floats;
} dl_f32_gemm_t;
/* nn.Linear forward (trans 0,0: mmdit.py:81,102,260-264, nn.py:530), data gradient (0,1: dX = dY W), weight gradient (1,1 with
 * A = dY [R, out], B = X [R, in], accumulate), and F.scaled_dot_product_attention's matmuls (mmdit.py:92-100) over materialised
 * probabilities: S = alpha Q K^T (0,0), O = P V (0,1), dP = dO V^T (0,0), dQ = alpha dS K (0,1), dK = alpha dS^T Q (1,1), dV = P^T dO. */
DL_API int dl_f32_gemm(const dl_f32_gemm_t* desc, dl_stream_t stream);
/* modulate(LayerNorm(x)) in f32 (mmdit.py:299,305,547 + nn.py:539), same contract as dl_ln_modulate_fwd with f32 rows everywhere
 * (modulation rows included); t != NULL applies the previous sub-layer's gated residual first (mmdit.py:302,308) */
DL_API int dl_f32_ln_modulate_fwd(const float* x, const float* w, const float* b, const float* scale, const float* shift,
                                  int64_t ld_mod, int64_t rows_per_mod, float eps, float* out, float* mean, float* rstd,
                                  const float* t, const float* gate, int64_t ld_gate, float* x_out, int64_t M, int64_t D,
                                  dl_stream_t stream);
/* its backward (contract of dl_ln_modulate_bwd, f32): one workgroup per modulation group, so dscale / dshift / dgate rows [group, :]
 * (row stride ld_dmod) and dwb_partial [groups, 2, D] are WRITTEN by their single producer (no atomics); M % rows_per_mod == 0 */
DL_API int dl_f32_ln_modulate_bwd(const float* dout, const float* x, const float* w, const float* b, const float* scale,
                                  int64_t ld_mod, int64_t rows_per_mod, const float* mean, const float* rstd, const float* dres,
                                  float* dx, float* dscale, float* dshift, int64_t ld_dmod, float* dwb_partial, const float* gate_t,
                                  const float* gate, int64_t ld_gate, float* dt, float* dgate, int64_t M, int64_t D,
                                  dl_stream_t stream);
/* QKNorm (nn.py:427-431,473-475: RMS over the full inner dim) + N-D RoPE on interleaved pairs (nn.py:345-353) in f32: qkv [B*N, ld]
 * (q in columns [0, D), k in [D, 2D)) -> qk [B*N, 2D] token-major (heads stay column blocks: dl_f32_gemm addresses them by stride);
 * rrms f32 [B*N, 2]; cos / sin f32 [rows, rot/2]; pos int32 [B*N] = rotary table row of every token (SPRINT's kept tokens,
 * sprint.py:347-349) or NULL (row n of the table for token n of its sample) */
DL_API int dl_f32_qk_norm_rope_fwd(const float* qkv, int64_t ld, const float* scale_q, const float* scale_k, const float* cos,
                                   const float* sin, float* qk, float* rrms, int64_t B, int64_t N, int64_t H, int64_t dh,
                                   int64_t rot, float eps, const int32_t* pos, dl_stream_t stream);
/* backward: dqk [B*N, 2D] -> columns [0, 2D) of dqkv (row stride ld_d); dscale_partials f32 [B, 2, D] written (one workgroup per
 * sample), folded by dl_reduce_rows_f32 */
DL_API int dl_f32_qk_norm_rope_bwd(const float* dqk, const float* qkv, int64_t ld, const float* scale_q, const float* scale_k,
                                   const float* cos, const float* sin, const float* rrms, float* dqkv, int64_t ld_d,
                                   float* dscale_partials, int64_t B, int64_t N, int64_t H, int64_t dh, int64_t rot,
                                   const int32_t* pos, dl_stream_t stream);
/* SPRINT token routing in f32 (sprint.py:317-387; the bf16 forms: dl_scatter_tokens_add, dl_restore_tokens, dl_masked_colsum,
 * dl_gated_residual_fwd, dl_gate_bwd -- same argument meaning; the gather copies rows bytewise, so dl_gather_tokens with 2 D columns
 * IS the f32 gather).  No atomics: dl_f32_masked_colsum_partials writes partials f32 [slabs, C] (slab s = rows [s ceil(R / slabs),
 * ...)) for a fixed-order fold (dl_reduce_rows_batched_f32); dl_f32_gate_bwd WRITES dgate (one workgroup per modulation group). */
DL_API int dl_f32_scatter_tokens_add(const float* src, int64_t ld_src, const int32_t* idx, float* dst, int64_t ld_dst, int64_t B,
                                     int64_t N, int64_t k, int64_t D, dl_stream_t stream);
DL_API int dl_f32_restore_tokens(const float* xd, int64_t ld_xd, const int32_t* inv, const float* mask, float* out, int64_t ld_out,
                                 int64_t B, int64_t N, int64_t k, int64_t D, dl_stream_t stream);
DL_API int dl_f32_masked_colsum_partials(const float* x, int64_t ld, const int32_t* sel, float* partials, int64_t slabs, int64_t R,
                                         int64_t C, dl_stream_t stream);
DL_API int dl_f32_gated_residual_fwd(const float* x, const float* t, const float* gate, int64_t ld_gate, int64_t rows_per_mod,
                                     float* out, int64_t ld_out, int64_t M, int64_t D, dl_stream_t stream);
DL_API int dl_f32_gate_bwd(const float* dout, const float* t, const float* gate, int64_t ld_gate, int64_t rows_per_mod, float* dt,
                           float* dgate, int64_t ld_dgate, int64_t M, int64_t D, dl_stream_t stream);
/* DDT decoder conditioning in f32 (ddt.py:423-424, nn.py:530; the bf16 forms: dl_ddt_cond_fwd / _bwd): out = silu(silu(enc + temb[b]));
 * backward: denc written, dtemb f32 [B, ld_t] WRITTEN (one workgroup per sample, no atomics) */
DL_API int dl_f32_ddt_cond_fwd(const float* enc, int64_t ld, const float* temb, int64_t ld_t, int64_t B, int64_t N, int64_t D, float* out,
                               dl_stream_t stream);
DL_API int dl_f32_ddt_cond_bwd(const float* dsz, const float* enc, int64_t ld, const float* temb, int64_t ld_t, int64_t B, int64_t N,
                               int64_t D, float* denc, float* dtemb, dl_stream_t stream);
/* softmax over the last dimension of the scaled scores (mmdit.py:92-100), in place; backward dS = P (dP - rowsum(dP P)) over dP */
DL_API int dl_f32_softmax_fwd(float* s, int64_t rows, int64_t cols, dl_stream_t stream);
DL_API int dl_f32_softmax_bwd(const float* p, float* dp, int64_t rows, int64_t cols, dl_stream_t stream);
/* PackedSwiGLU nn.py:484-486 in f32: u [M, 2F] -> h [M, F]; backward du [M, 2F] */
DL_API int dl_f32_swiglu_fwd(const float* u, float* h, int64_t M, int64_t F, dl_stream_t stream);
DL_API int dl_f32_swiglu_bwd(const float* dh, const float* u, float* du, int64_t M, int64_t F, dl_stream_t stream);
/* out = a + b ; dx = dy * silu'(pre) */
DL_API int dl_f32_add(const float* a, const float* b, float* out, int64_t n, dl_stream_t stream);
DL_API int dl_f32_silu_bwd(const float* dy, const float* pre, float* dx, int64_t n, dl_stream_t stream);
/* dl_patchify / dl_timestep_embedding / dl_cond_combine_{fwd,bwd} with f32 outputs (mmdit.py:757-765, nn.py:106-114,
 * mmdit.py:867-868); the label-table gradient is added in batch order by one thread per column (no atomics) */
DL_API int dl_f32_patchify(const float* x, float* tok, int64_t B, int64_t C, int64_t H, int64_t W, int64_t p, int64_t ld, int order,
                           dl_stream_t stream);
DL_API int dl_f32_timestep_embedding(const float* t, float* out, int64_t B, int64_t dim, float max_period, dl_stream_t stream);
DL_API int dl_f32_cond_combine_fwd(const float* e, const float* table, const int64_t* idx, float* emb, float* act, int64_t B,
                                   int64_t E, dl_stream_t stream);
DL_API int dl_f32_cond_combine_bwd(const float* dact, const float* emb, const int64_t* idx, float* demb, float* dtable, int64_t B,
                                   int64_t E, dl_stream_t stream);

/* ---- fp32-class regime of the UNet (networks/denoisers/unet.py, networks/utils/nn.py:11-88): NHWC f32 rows [B*H*W, C].
 * A 3x3 / pad-1 convolution (unet.py:187,208,594,745) = dl_f32_im2col3x3 + dl_f32_gemm against the weight in its native [Co, Ci*9]
 * layout: cols[p, ci*9 + ky*3 + kx] = x[p + (ky-1, kx-1), ci] (zero outside the image); data gradient = dl_f32_gemm(dY, W) into dcols
 * + dl_f32_col2im3x3 (the adjoint gather, one writer per element); weight gradient = dl_f32_gemm(dY^T cols), accumulated. */
DL_API int dl_f32_im2col3x3(const float* x, int64_t ldx, float* cols, int64_t B, int64_t H, int64_t W, int64_t C, dl_stream_t stream);
DL_API int dl_f32_col2im3x3(const float* dcols, float* dx, int64_t ld_dx, int64_t B, int64_t H, int64_t W, int64_t C,
                            dl_stream_t stream);
/* GroupNorm32 (nn.py:11-13) in f32: statistics per (sample, group) f32 [B, G, 2] = mean, rstd (two passes, as the reference);
 * out = act((xhat w + b)(1 + film_scale[b, c]) + film_shift[b, c]) (unet.py:215-237; film NULL: plain GroupNorm, unet.py:296-322);
 * backward: dx = dres + gradient through the norm, dfilm_scale / dfilm_shift [B, ld_film] written, dw_partial / db_partial [B, C]
 * per-sample partials WRITTEN (the caller folds them in a fixed order: dl_reduce_rows_batched_f32) -- no atomics */
DL_API int dl_f32_gn_stats(const float* x, float* stats, int64_t B, int64_t HW, int64_t C, int64_t G, float eps, dl_stream_t stream);
DL_API int dl_f32_gn_apply_fwd(const float* x, const float* stats, const float* w, const float* b, const float* film_scale,
                               const float* film_shift, int64_t ld_film, int act_silu, float* out, int64_t B, int64_t HW, int64_t C,
                               int64_t G, dl_stream_t stream);
DL_API int dl_f32_gn_bwd(const float* dout, const float* x, const float* stats, const float* w, const float* b,
                         const float* film_scale, const float* film_shift, int64_t ld_film, int act_silu, const float* dres, float* dx,
                         float* dw_partial, float* db_partial, float* dfilm_scale, float* dfilm_shift, int64_t B, int64_t HW, int64_t C,
                         int64_t G, dl_stream_t stream);
/* 2x2 resampling (nn.py:28-88) at the SMALL resolution Hs x Ws: mode 0 reduce (scale * sum of the 2x2 window: avg-pool forward 0.25 /
 * nearest-upsample backward 1), 1 expand (scale * x[y/2, x/2]), 2 pick (x[2y, 2x]: stride-2 sampling of a stride-1 conv, nn.py:79),
 * 3 stuff (adjoint of pick) */
DL_API int dl_f32_resample2x2(const float* x, float* out, int64_t B, int64_t Hs, int64_t Ws, int64_t C, float scale, int mode,
                              dl_stream_t stream);
/* boundary layout converts (unet.py:832), strided 2-D copy (torch.cat / split of the skip connections, unet.py:846-851), additive
 * ResBlock conditioning h + emb_out (unet.py:235-237) and its per-sample pixel sum */
DL_API int dl_f32_nchw_to_nhwc(const float* x, float* out, int64_t B, int64_t C, int64_t HW, int64_t ld, dl_stream_t stream);
DL_API int dl_f32_nhwc_to_nchw(const float* x, float* out, int64_t B, int64_t C, int64_t HW, int64_t ld, dl_stream_t stream);
DL_API int dl_f32_copy2d(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int64_t rows, int64_t cols, dl_stream_t stream);
DL_API int dl_f32_rowbias_add(const float* x, const float* e, int64_t lde, float* out, int64_t B, int64_t HW, int64_t C,
                              dl_stream_t stream);
DL_API int dl_f32_rowbias_bwd(const float* dy, float* de, int64_t lde, int64_t B, int64_t HW, int64_t C, dl_stream_t stream);

/* ------------------------------------------------------------------ fused block driver */
/* One adaLN-zero DiT block (DiTBlock._forward mmdit.py:288-309, DiTAttention mmdit.py:75-104, MLP mmdit.py:260-264) as ONE call
 * per direction: the library issues the block's launch sequence itself (csrc/block.hip) -- the SURVEY section 8b "fused driver".
 * Every pointer is a caller-owned device buffer (bf16 unless noted); M = B*N token rows, dh = D/H = 64, F = mlp_ratio*D.
 * Index of p[]: */
enum {
  /* forward inputs / saved activations */
  DL_BLK_X_IN = 0,    /* [M,D] residual stream entering the block (written when PEND_X is given) */
  DL_BLK_PEND_X,      /* [M,D] or NULL: x before the previous block's MLP residual; then x_in = pend_x + pend_gate * pend_t */
  DL_BLK_PEND_T,      /* [M,D] previous block's t2 */
  DL_BLK_PEND_GATE,   /* [Bp, ld_mod] row view: previous block's gate2 */
  DL_BLK_SCALE1, DL_BLK_SHIFT1, DL_BLK_GATE1, DL_BLK_SCALE2, DL_BLK_SHIFT2, DL_BLK_GATE2, /* row views of the modulation matrix */
  DL_BLK_LN1_W, DL_BLK_LN1_B, DL_BLK_LN2_W, DL_BLK_LN2_B, DL_BLK_QN_SCALE, DL_BLK_KN_SCALE,  /* f32 parameters */
  DL_BLK_W_QKV, DL_BLK_W_PROJ, DL_BLK_W_UP, DL_BLK_W_UP_PERM, DL_BLK_W_DOWN,          /* bf16 shadows [out, in] */
  DL_BLK_WT_QKV, DL_BLK_WT_PROJ, DL_BLK_WT_UP, DL_BLK_WT_DOWN,                        /* transposed shadows [in, out] */
  DL_BLK_ROPE_COS, DL_BLK_ROPE_SIN,                                                   /* f32 [N, rot/2] */
  DL_BLK_XM1, DL_BLK_MEAN1, DL_BLK_RSTD1, DL_BLK_QKV, DL_BLK_Q, DL_BLK_K, DL_BLK_V /* NULL: V in place (N <= 256) */, DL_BLK_RRMS,
  DL_BLK_A, DL_BLK_LSE, DL_BLK_T1, DL_BLK_X1, DL_BLK_XM2, DL_BLK_MEAN2, DL_BLK_RSTD2, DL_BLK_U, DL_BLK_H, DL_BLK_T2,
  /* backward */
  DL_BLK_DT2,         /* [M,D] gradient of t2 (produced by the LayerNorm backward of the block above / the head) */
  DL_BLK_DX_IN,       /* [M,D] residual-stream gradient entering from above */
  DL_BLK_DX_MID,      /* [M,D] scratch: residual-stream gradient between the two branches */
  DL_BLK_DX_OUT,      /* [M,D] residual-stream gradient leaving the block (may alias DX_IN) */
  DL_BLK_DH, DL_BLK_DU, DL_BLK_DXM, DL_BLK_DT1, DL_BLK_DA, DL_BLK_DQ, DL_BLK_DK, DL_BLK_DV, DL_BLK_DQKV, DL_BLK_DFEAT /* or NULL */,
  DL_BLK_DSCALE1, DL_BLK_DSHIFT1, DL_BLK_DGATE1, DL_BLK_DSCALE2, DL_BLK_DSHIFT2,      /* f32 row views of the modulation gradient */
  DL_BLK_DWB1, DL_BLK_DWB2,                                                           /* f32 [B,2,D] per-sample LayerNorm-affine sums */
  DL_BLK_PREV_T2, DL_BLK_PREV_GATE2, DL_BLK_PREV_DT2, DL_BLK_PREV_DGATE2,             /* previous block's MLP residual, or NULL x4 */
  DL_BLK_G_QKV, DL_BLK_G_PROJ, DL_BLK_G_UP, DL_BLK_G_DOWN, DL_BLK_G_LN1, DL_BLK_G_LN2, DL_BLK_G_QK_SCALE,  /* f32 gradients (+=) */
  /* row_gemms mode: the LayerNorm that FOLLOWS this block (LN1 of the next block, or the final LayerNorm) runs in the epilogue of
   * this block's MLP-down GEMM: its affine parameters (NULL x2: none), modulation rows and outputs */
  DL_BLK_NEXT_LN_W, DL_BLK_NEXT_LN_B, DL_BLK_NEXT_SCALE, DL_BLK_NEXT_SHIFT,
  DL_BLK_NEXT_X,      /* [M,D] residual stream leaving the block (x1 + gate2 * t2) */
  DL_BLK_NEXT_XM, DL_BLK_NEXT_MEAN, DL_BLK_NEXT_RSTD,
  DL_BLK_TN_SLAB,     /* f32 [tn_slab_floats] scratch of dl_gemm_tn_group, or NULL: the four weight gradients as dl_gemm_tn_ex launches */
  DL_BLK_QK_PARTIALS, /* f32 [1024 * 2 * D] scratch of dl_qk_norm_rope_bwd_inplace, or NULL; with it (V in place, D <= 512) dQ / dK /
                       * dV are written token-major into DQKV (dl_attn_bwd_tok) and DQ / DK are not used */
  DL_BLK_SSQ,         /* f32 [M, 2] ZEROED by the caller before dl_dit_block_fwd, or NULL; with it (row_gemms, V in place) the forward
                       * runs dl_gemm_nt_ssq + dl_attn_fwd_qkn: no QK-norm + RoPE pass of its own */
  DL_BLK_NPTR
};
typedef struct dl_dit_block_t {
  void* p[96];                   /* indexed by DL_BLK_* */
  int64_t B, N, D, H, F;
  int64_t ld_mod, ld_dmod;       /* row strides (elements) of the bf16 modulation matrix and of its f32 gradient */
  int64_t ldw_d, ldw_f;          /* row strides of the forward shadows with in = D / in = F */
  int64_t ldwt_d, ldwt_f2, ldwt_3d; /* row strides of the transposed shadows with out = D / 2F / 3D */
  int64_t rot;                   /* rotary width per head */
  float eps;                     /* LayerNorm epsilon (1e-5 in the blocks) */
  float next_eps;                /* epsilon of the LayerNorm behind DL_BLK_NEXT_* */
  int32_t row_gemms;             /* bit 0: LayerNorm-modulate forward / backward as GEMM epilogues (dl_ln_modulate_gemm_*; needs D == 384,
                                  * N == 256): LN1 is NOT run by dl_dit_block_fwd (XM1 / MEAN1 / RSTD1 come from the previous block's
                                  * MLP-down epilogue or the patch embedding) and the MLP branch's gated residual IS applied (NEXT_X);
                                  * bit 1: QK-norm + RoPE in the qkv GEMM's epilogue; bit 2: dl_dit_block_bwd leaves the LayerNorm-affine
                                  * partials DWB1 / DWB2 unfolded (the caller folds all blocks at once: dl_reduce_rows_batched_f32) */
  int64_t tn_slab_floats;        /* size of DL_BLK_TN_SLAB; with a slab the block's four weight gradients are ONE atomics-free launch
                                  * (dl_gemm_tn_group) issued on `side` once dqkv exists */
  int32_t max_workgroups;        /* > 0: workgroup budget of the block's persistent main-chain kernels (one workgroup per CU by default).
                                  * Data parallel: CUs - r leaves r CUs to the communication library's workgroups, so a gradient
                                  * exchange that runs under the backward costs r / CUs instead of a second round of every launch
                                  * (base_trainer.py:277-279: the DDP all-reduce overlaps the backward) */
} dl_dit_block_t;
/* forward of one block; train != 0 keeps the MLP pre-activations U for the backward.  The MLP branch's gated residual
 * (x1 + gate2 * t2) is NOT applied: it is the next block's (or the final LayerNorm's) pending triple. */
DL_API int dl_dit_block_fwd(const dl_dit_block_t* blk, int train, dl_stream_t stream);
/* backward of one block on `main`; the weight-gradient GEMMs and LayerNorm-affine folds are issued on `side` behind events
 * (join `side` before the gradients are consumed; side_stream == main_stream issues everything inline on one stream);
 * side_workgroups caps the persistent wgrad workgroups (0 = one per CU) */
DL_API int dl_dit_block_bwd(const dl_dit_block_t* blk, dl_stream_t main_stream, dl_stream_t side_stream, int side_workgroups);

#ifdef __cplusplus
}
#endif
#endif /* DIFFULAB_HIP_H */
