/* libdiffulab_probe.so -- LAB / TEST instrumentation, NOT part of the product ABI (that is diffulab_hip.h + diffulab_comm.h).
 *
 * Built from diffulab_amd/csrc/lab/probe.hip by `make probe` (also part of `make all`, so the GPU tests find it in-tree).  Nothing
 * under diffulab_amd/ loads this library.  Two kinds of entry points:
 *   - instruction-semantics probes the GPU TESTS use to pin the operand layouts the product kernels rely on
 *     (dl_probe_tr16, dl_probe_mfma_f8);
 *   - tuning probes for scripts/ probe scripts (sustained MFMA rate, operand-DMA stream, store patterns).
 * Same conventions as diffulab_hip.h: plain pointers, int status (0 = ok), the caller's hipStream_t. */
#ifndef DIFFULAB_PROBE_H
#define DIFFULAB_PROBE_H
#include <stdint.h>

#include "diffulab_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

DL_API const char* dl_probe_last_error(void);
/* raw ds_read_b64_tr_b16 lane map: fills out[64*4] with what each lane receives when lane l passes address 8*l over an LDS image
 * holding the uint16 values 0..255 */
DL_API int dl_probe_tr16(uint16_t* out, dl_stream_t stream);
/* D f32 [32,32] = A[32,64] . B[32,64]^T for e4m3 bytes, through one v_mfma_scale_f32_32x32x64_f8f6f4 with the operand layout
 * attention_fp8.hip relies on (lane l: row l & 31, bytes [32 (l >> 5), +32)); pins the instruction's semantics */
DL_API int dl_probe_mfma_f8(const void* a, const void* b, float* d, dl_stream_t stream);
/* sustained MFMA 32x32x16 bf16 rate of the GEMM main-loop skeleton on 256 workgroups x 8 waves, `iters` k-steps of 24 MFMAs per
 * wave: mode 0 MFMA only, 1 + LDS fragment reads, 2 + one workgroup barrier per k-step, 3 + the 56 KiB direct-to-LDS DMA per
 * k-step from `src` (>= 256*57344 bytes), 4-7 copy-path variants.  out: f32 [256*512] (sink).  modes 8 / 9: store-pattern probe --
 * `out` is a bf16 [65536, 1152] buffer written `iters` times in the GEMM register epilogue's pattern (32 rows x 32 B per
 * instruction) / with full 128-byte lines per 8 lanes. */
DL_API int dl_probe_mfma(int mode, int iters, const void* src, float* out, dl_stream_t stream);
/* the operand DMA stream of the 256 x 384 NT GEMM tile walk alone (kb = 128: 64-deep steps / 2 ring slots, kb = 64: 32-deep steps
 * with 64-byte row segments / 4 ring slots); out: >= 256*512 floats */
DL_API int dl_probe_dma(int kb, const void* A, const void* Bw, int64_t M, int64_t K, float* out, dl_stream_t stream);

/* the same stream with a free geometry (round 4): 256 * wgs_per_cu workgroups of `threads` threads, tiles of rows_a activation rows +
 * rows_b (shared, L2-resident) weight rows per 64-deep stage, an nslot-deep LDS ring; nothing is computed.  Answers whether the
 * L2 -> LDS operand rate of a CU scales with the number of resident / issuing waves.  out: >= 256*512 floats */
DL_API int dl_probe_dma2(int wgs_per_cu, int threads, int rows_a, int rows_b, int nslot, const void* A, const void* Bw, int64_t M,
                         int64_t K, float* out, dl_stream_t stream);
/* dl_probe_dma2 with the k-steps of workgroup w of an XCD rotated by (rot * w) % (K / 64): the workgroups of an XCD then read different
 * lines of the shared weight panel at any moment (is the operand stream bound by a hot L2 channel?) */
DL_API int dl_probe_dma2_rot(int wgs_per_cu, int threads, int rows_a, int rows_b, int nslot, int rot, const void* A, const void* Bw,
                             int64_t M, int64_t K, float* out, dl_stream_t stream);
/* the 256 + 384-row tile walk of dl_probe_dma with PLAIN 16-byte loads into registers instead of direct-to-LDS loads (nothing is
 * written to the LDS): pattern 0 = the DMA's addresses (8 rows x 128 B per wave instruction), 1 = the MFMA fragment pattern (32 rows x
 * 32 B); depth = stages of loads in flight per wave (2 | 3).  Is the L2 -> CU path faster than its direct-to-LDS form? */
DL_API int dl_probe_ld(int pattern, int depth, const void* A, const void* Bw, int64_t M, int64_t K, float* out, dl_stream_t stream);
/* `n_wgs` workgroups of `threads` threads that do nothing but hold their CU slots for `usec` microseconds (wall clock): a stand-in
 * for a communication kernel (RCCL's channel workgroups) beside the compute stream, to measure what co-residency costs the
 * one-workgroup-per-CU kernels (scripts/lab/occupied_cus.py). */
DL_API int dl_probe_spin(int n_wgs, int threads, int usec, dl_stream_t stream);

/* LAB (rounds 2-3, not used by any engine: forward-only, breaks even end to end once its quantisation pre-pass is counted; the
 * product's config-5 attention is the bf16 kernel).  fp8 (OCP e4m3) forward of the general form for the long joint text-image sequences (BASELINE config 5; mmdit.py:172-190) on the
 * CDNA4 block-scaled matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4 (one MFMA contracts a whole 64-wide head dimension).
 * dl_probe_attn_fp8_quantize: q [B,H,Nq,64], k, v [B,H,Nk,64] bf16 -> q8, k8 (same row layout, 1 byte per element), v8t [B,H,64,Nk]
 * (V transposed, the keys of every 64-key block in the order the kernel's P registers hold them) and scales f32 [B,H,3]
 * (amax / 448 of q, k, v per head).  dl_probe_attn_fwd_fp8: same outputs as dl_attn_fwd_ex (out bf16 [B,Nq,H*64], lse f32 [B,H,Nq]);
 * Nq, Nk multiples of 256 up to 4096; key_bias as in dl_attn_fwd_ex.  The backward stays dl_attn_bwd_ex on the bf16 tensors. */
DL_API int dl_probe_attn_fp8_quantize(const void* q, const void* k, const void* v, void* q8, void* k8, void* v8t, float* scales,
                                int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t dh, dl_stream_t stream);
DL_API int dl_probe_attn_fwd_fp8(const void* q8, const void* k8, const void* v8t, const float* scales, void* out, float* lse,
                           int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t dh, float scale, const float* key_bias,
                           dl_stream_t stream);
/* round 6: QK^T only in fp8 (q8, k8 of dl_probe_attn_fp8_quantize), the probabilities and V stay bf16 (v bf16 [B,H,Nk,64]) */
DL_API int dl_probe_attn_fwd_fp8qk(const void* q8, const void* k8, const void* v, const float* scales, void* out, float* lse,
                                   int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t dh, float scale, const float* key_bias,
                                   dl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFULAB_PROBE_H */
