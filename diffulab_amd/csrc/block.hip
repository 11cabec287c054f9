// dl_dit_block_{fwd,bwd}: the launch sequence of ONE adaLN-zero DiT block (reference: DiTBlock._forward mmdit.py:288-309 with
// DiTAttention.forward mmdit.py:75-104 and the PackedSwiGLU MLP nn.py:478-486 / mmdit.py:260-264) issued from native code.
//
// The kernels are the library's own entry points (dl_ln_modulate_*, dl_gemm_nt*, dl_qk_norm_rope_*, dl_attn_*_sv / _ex,
// dl_swiglu_*, dl_gemm_tn_ex, dl_reduce_rows_f32); this file only orders them, exactly as diffulab_amd/engine.py does from Python:
//   forward : [x += gate * t of the previous sub-layer fused into] LN-modulate -> qkv GEMM -> QK-RMSNorm + RoPE + head split ->
//             attention -> projection GEMM -> LN-modulate (+ the attention branch's gated residual) -> MLP-up GEMM with the fused
//             SwiGLU epilogue -> MLP-down GEMM.  The MLP branch's gated residual is left pending for the next block's LayerNorm.
//   backward: the reverse chain on `main`; the four weight-gradient GEMMs and the two LayerNorm-affine folds are off the dependency
//             chain and go to `side` behind an event, capped at `side_wgs` workgroups.
// A host that is not Python (the C ABI's reason to exist) drives a whole DiT with one call per block and direction; the Python
// engine uses it too (DL_NATIVE_BLOCK, default on): ~25 ctypes calls and tensor-view constructions per block and direction
// become one.
#include <stdlib.h>

#include "common.h"

static thread_local hipEvent_t g_blk_event = nullptr;
static int fork_to_side(hipStream_t main, hipStream_t side) {
  if (main == side) return 0;
  if (!g_blk_event && hipEventCreateWithFlags(&g_blk_event, hipEventDisableTiming) != hipSuccess) return -1;
  if (hipEventRecord(g_blk_event, main) != hipSuccess) return -1;
  return hipStreamWaitEvent(side, g_blk_event, 0) == hipSuccess ? 0 : -1;
}

// the same hand-off for hosts that issue the launches themselves (the Python engines' side streams, launch plans): everything queued
// on `waited` so far happens before anything queued on `waiter` from now on.  One thread-local event, re-recorded per call (a wait
// refers to the record that precedes it).
extern "C" int dl_stream_wait_stream(dl_stream_t waiter, dl_stream_t waited) {
  if (fork_to_side((hipStream_t)waited, (hipStream_t)waiter) != 0) {
    dl_set_error("dl_stream_wait_stream: %s", hipGetErrorString(hipGetLastError()));
    return DL_ERR_LAUNCH;
  }
  return DL_OK;
}
extern "C" int dl_memset_zero(void* p, int64_t bytes, dl_stream_t stream) {
  DL_CHECK_ARG(p && bytes >= 0, "dl_memset_zero: bad args");
  if (bytes && hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream) != hipSuccess) {
    dl_set_error("dl_memset_zero: %s", hipGetErrorString(hipGetLastError()));
    return DL_ERR_LAUNCH;
  }
  return DL_OK;
}

#define P(k) (b->p[DL_BLK_##k])
#define RUN(call)               \
  do {                          \
    const int rc__ = (call);    \
    if (rc__ != DL_OK) return rc__; \
  } while (0)

extern "C" int dl_dit_block_fwd(const dl_dit_block_t* b, int train, dl_stream_t stream) {
  DL_CHECK_ARG(b, "dl_dit_block_fwd: null block");
  const int64_t B = b->B, N = b->N, D = b->D, H = b->H, F = b->F, M = B * N, dh = D / H;
  DL_CHECK_ARG(B > 0 && N > 0 && D > 0 && H > 0 && F > 0 && dh == 64, "dl_dit_block_fwd: bad dims (head_dim must be 64)");
  const DlWgCapScope cap_scope(b->max_workgroups);
  const float sm = 0.125f;  // 64^-0.5
  if (b->row_gemms & 1) {
    // Row-complete GEMMs (csrc/gemm_ln.hip): every LayerNorm-modulate is the epilogue of the GEMM in front of it.  XM1 / MEAN1 /
    // RSTD1 were written by the previous block's MLP-down GEMM (or the patch embedding); the projection carries LN2, the MLP-down
    // GEMM the gated residual of the MLP branch and the LayerNorm that follows the block.
    DL_CHECK_ARG(P(V) == nullptr && P(NEXT_X) && P(NEXT_XM) && P(NEXT_SCALE) && P(NEXT_SHIFT) && P(NEXT_MEAN) && P(NEXT_RSTD),
                 "dl_dit_block_fwd: row_gemms needs V in place and the DL_BLK_NEXT_* slots");
    if (P(SSQ)) {  // QK-norm statistics leave with the qkv GEMM, the norm + RoPE are applied as the attention stages q and k
      RUN(dl_gemm_nt_ssq(P(XM1), D, P(W_QKV), b->ldw_d, P(QKV), 3 * D, M, 3 * D, D, (float*)P(SSQ), 2, stream));
      // (inference: the normalised q, k and rrms feed only the backward -- not written)
      RUN(dl_attn_fwd_qkn(P(QKV), (const float*)P(SSQ), (const float*)P(QN_SCALE), (const float*)P(KN_SCALE), (const float*)P(ROPE_COS),
                          (const float*)P(ROPE_SIN), 1e-6f, b->rot, train ? P(Q) : nullptr, train ? P(K) : nullptr,
                          train ? (float*)P(RRMS) : nullptr, P(A), (float*)P(LSE), B, H, N, dh, sm, stream));
    } else if (b->row_gemms & 2) {
      RUN(dl_gemm_nt_qk_norm_rope(P(XM1), D, P(W_QKV), b->ldw_d, B, N, H, dh, b->rot, 1e-6f, (const float*)P(QN_SCALE),
                                  (const float*)P(KN_SCALE), (const float*)P(ROPE_COS), (const float*)P(ROPE_SIN), P(QKV), P(Q), P(K),
                                  (float*)P(RRMS), N, 0, stream));
    } else {
      RUN(dl_gemm_nt(P(XM1), D, P(W_QKV), b->ldw_d, P(QKV), 3 * D, M, 3 * D, D, nullptr, DL_ACT_NONE, DL_BF16, nullptr, nullptr, 0, nullptr,
                     0, 1, stream));
      RUN(dl_qk_norm_rope_fwd(P(QKV), (const float*)P(QN_SCALE), (const float*)P(KN_SCALE), (const float*)P(ROPE_COS),
                              (const float*)P(ROPE_SIN), P(Q), P(K), nullptr, (float*)P(RRMS), B, N, H, dh, b->rot, 1e-6f, stream));
    }
    if (!P(SSQ))
      RUN(dl_attn_fwd_sv(P(Q), P(K), (const char*)P(QKV) + 2 * D * 2, N * 3 * D, dh, 3 * D, P(A), (float*)P(LSE), B, H, N, dh, sm, stream));
    RUN(dl_ln_modulate_gemm_fwd(P(A), D, P(W_PROJ), b->ldw_d, M, D, P(X_IN), P(GATE1), b->ld_mod, (const float*)P(LN2_W),
                                (const float*)P(LN2_B), P(SCALE2), P(SHIFT2), b->ld_mod, N, b->eps, P(T1), P(X1), P(XM2), (float*)P(MEAN2),
                                (float*)P(RSTD2), D, stream));
    RUN(dl_gemm_nt_swiglu(P(XM2), D, P(W_UP_PERM), b->ldw_d, train ? P(U) : nullptr, 2 * F, P(H), F, M, F, D, stream));
    RUN(dl_ln_modulate_gemm_fwd(P(H), F, P(W_DOWN), b->ldw_f, M, F, P(X1), P(GATE2), b->ld_mod, (const float*)P(NEXT_LN_W),
                                (const float*)P(NEXT_LN_B), P(NEXT_SCALE), P(NEXT_SHIFT), b->ld_mod, N, b->next_eps, P(T2), P(NEXT_X),
                                P(NEXT_XM), (float*)P(NEXT_MEAN), (float*)P(NEXT_RSTD), D, stream));
    return DL_OK;
  }
  // LayerNorm 1 (+ the pending gated residual of the previous block's MLP branch: x_in <- pend_x + pend_gate * pend_t)
  RUN(dl_ln_modulate_fwd(P(PEND_X) ? P(PEND_X) : P(X_IN), (const float*)P(LN1_W), (const float*)P(LN1_B), P(SCALE1), P(SHIFT1),
                         b->ld_mod, N, b->eps, P(XM1), (float*)P(MEAN1), (float*)P(RSTD1), P(PEND_X) ? P(PEND_T) : nullptr,
                         P(PEND_X) ? P(PEND_GATE) : nullptr, b->ld_mod, P(PEND_X) ? P(X_IN) : nullptr, M, D, stream));
  RUN(dl_gemm_nt(P(XM1), D, P(W_QKV), b->ldw_d, P(QKV), 3 * D, M, 3 * D, D, nullptr, DL_ACT_NONE, DL_BF16, nullptr, nullptr, 0, nullptr, 0,
                 1, stream));
  const bool v_in_place = P(V) == nullptr;  // N <= 256: the attention addresses V inside the token-major qkv rows
  RUN(dl_qk_norm_rope_fwd(P(QKV), (const float*)P(QN_SCALE), (const float*)P(KN_SCALE), (const float*)P(ROPE_COS),
                          (const float*)P(ROPE_SIN), P(Q), P(K), P(V), (float*)P(RRMS), B, N, H, dh, b->rot, 1e-6f, stream));
  if (v_in_place)
    RUN(dl_attn_fwd_sv(P(Q), P(K), (const char*)P(QKV) + 2 * D * 2, N * 3 * D, dh, 3 * D, P(A), (float*)P(LSE), B, H, N, dh, sm, stream));
  else
    RUN(dl_attn_fwd(P(Q), P(K), P(V), P(A), (float*)P(LSE), B, H, N, dh, sm, stream));
  RUN(dl_gemm_nt(P(A), D, P(W_PROJ), b->ldw_d, P(T1), D, M, D, D, nullptr, DL_ACT_NONE, DL_BF16, nullptr, nullptr, 0, nullptr, 0, 1, stream));
  // LayerNorm 2 (+ the attention branch's gated residual: x1 = x_in + gate1 * t1)
  RUN(dl_ln_modulate_fwd(P(X_IN), (const float*)P(LN2_W), (const float*)P(LN2_B), P(SCALE2), P(SHIFT2), b->ld_mod, N, b->eps, P(XM2),
                         (float*)P(MEAN2), (float*)P(RSTD2), P(T1), P(GATE1), b->ld_mod, P(X1), M, D, stream));
  // (U == NULL in a training block: the backward recomputes the pre-activations, dl_mlp_dswiglu_recompute)
  int rc = dl_gemm_nt_swiglu(P(XM2), D, P(W_UP_PERM), b->ldw_d, train ? P(U) : nullptr, 2 * F, P(H), F, M, F, D, stream);
  if (rc == DL_ERR_UNSUPPORTED) {  // small / ragged shapes: the unfused pair
    DL_CHECK_ARG(P(U), "dl_dit_block_fwd: this shape has no fused MLP-up kernel and needs the U buffer");
    RUN(dl_gemm_nt(P(XM2), D, P(W_UP), b->ldw_d, P(U), 2 * F, M, 2 * F, D, nullptr, DL_ACT_NONE, DL_BF16, nullptr, nullptr, 0, nullptr, 0, 1,
                   stream));
    RUN(dl_swiglu_fwd(P(U), P(H), M, F, stream));
  } else if (rc != DL_OK) {
    return rc;
  }
  RUN(dl_gemm_nt(P(H), F, P(W_DOWN), b->ldw_f, P(T2), D, M, D, F, nullptr, DL_ACT_NONE, DL_BF16, nullptr, nullptr, 0, nullptr, 0, 1, stream));
  return DL_OK;
}

extern "C" int dl_dit_block_bwd(const dl_dit_block_t* b, dl_stream_t main_, dl_stream_t side_, int side_wgs) {
  DL_CHECK_ARG(b, "dl_dit_block_bwd: null block");
  hipStream_t main = (hipStream_t)main_, side = (hipStream_t)side_;  // (side == main: everything inline on one stream)
  const DlWgCapScope cap_scope(b->max_workgroups);
  const int64_t B = b->B, N = b->N, D = b->D, H = b->H, F = b->F, M = B * N, dh = D / H;
  const float sm = 0.125f;
  // With a slab the four weight gradients of the block are ONE atomics-free launch (dl_gemm_tn_group) behind the last of their
  // operands (dqkv); without one (or for shapes the 384 x 192 tile does not divide) they are four dl_gemm_tn_ex launches, each
  // issued as soon as its operands exist.
  dl_wgrad_t wg[4];
  int nwg = 0;
  bool grouped = P(TN_SLAB) != nullptr && ((D % 384 == 0 && F % 192 == 0) || (D % 256 == 0 && F % 256 == 0)) && M % 32 == 0 && M >= 2048;
  auto wgrad = [&](const void* dy, int64_t ldy, const void* x, int64_t ldx, void* g, int64_t Mo, int64_t No) -> int {
    if (grouped) {
      wg[nwg++] = dl_wgrad_t{dy, ldy, x, ldx, (float*)g, Mo, No};
      return DL_OK;
    }
    if (fork_to_side(main, side)) return DL_ERR_LAUNCH;
    return dl_gemm_tn_ex(dy, ldy, x, ldx, (float*)g, No, Mo, No, M, side_wgs, side);
  };
  auto fold = [&](void* partial, void* g) -> int {  // [B, 2, D] per-sample sums -> [w; b] gradients, cleared while read
    if (fork_to_side(main, side)) return DL_ERR_LAUNCH;
    return dl_reduce_rows_f32((float*)partial, (float*)g, B, 2 * D, 1, side);
  };
  // ---- MLP branch (dt2 / its dgate were produced by the LayerNorm backward that ran before this block)
  RUN(wgrad(P(DT2), D, P(H), F, P(G_DOWN), D, F));
  if (P(U)) {
    RUN(dl_gemm_nt(P(DT2), D, P(WT_DOWN), b->ldwt_d, P(DH), F, M, F, D, nullptr, DL_ACT_NONE, DL_BF16, nullptr, nullptr, 0, nullptr, 0, 1, main));
    RUN(dl_swiglu_bwd(P(DH), P(U), P(DU), M, F, main));
  } else {  // no saved pre-activations: recomputed per tile next to dH, neither written (csrc/mlp_bwd.hip)
    RUN(dl_mlp_dswiglu_recompute(P(XM2), D, P(W_UP_PERM), b->ldw_d, P(DT2), D, P(WT_DOWN), b->ldwt_d, P(DU), 2 * F, M, F, D, D, main));
  }
  RUN(wgrad(P(DU), 2 * F, P(XM2), D, P(G_UP), 2 * F, D));
  const bool rows = (b->row_gemms & 1) != 0, fold_here = (b->row_gemms & 4) == 0;
  if (rows) {  // LayerNorm-2 backward (+ the attention branch's gate backward) in the epilogue of the MLP-up data-gradient GEMM
    RUN(dl_ln_modulate_gemm_bwd(P(DU), 2 * F, P(WT_UP), b->ldwt_f2, M, 2 * F, P(X1), (const float*)P(LN2_W), (const float*)P(LN2_B),
                                P(SCALE2), b->ld_mod, N, (const float*)P(MEAN2), (const float*)P(RSTD2), P(DX_IN), P(DX_MID),
                                (float*)P(DSCALE2), (float*)P(DSHIFT2), b->ld_dmod, (float*)P(DWB2), P(T1), P(GATE1), b->ld_mod, P(DT1),
                                (float*)P(DGATE1), D, main));
  } else {
    RUN(dl_gemm_nt(P(DU), 2 * F, P(WT_UP), b->ldwt_f2, P(DXM), D, M, D, 2 * F, nullptr, DL_ACT_NONE, DL_BF16, nullptr, nullptr, 0, nullptr,
                   0, 1, main));
    RUN(dl_ln_modulate_bwd(P(DXM), P(X1), (const float*)P(LN2_W), (const float*)P(LN2_B), P(SCALE2), b->ld_mod, N,
                           (const float*)P(MEAN2), (const float*)P(RSTD2), P(DX_IN), P(DX_MID), (float*)P(DSCALE2), (float*)P(DSHIFT2),
                           b->ld_dmod, (float*)P(DWB2), P(T1), P(GATE1), b->ld_mod, P(DT1), (float*)P(DGATE1), M, D, main));
  }
  if (fold_here) RUN(fold(P(DWB2), P(G_LN2)));
  // ---- attention branch
  RUN(wgrad(P(DT1), D, P(A), D, P(G_PROJ), D, D));
  RUN(dl_gemm_nt(P(DT1), D, P(WT_PROJ), b->ldwt_d, P(DA), D, M, D, D, nullptr, DL_ACT_NONE, DL_BF16, nullptr, nullptr, 0, nullptr, 0, 1, main));
  const bool v_in_place = P(V) == nullptr;
  if (v_in_place && P(QK_PARTIALS) && D <= 512) {
    // every gradient of the attention leaves token-major inside the dqkv rows; the QK-norm backward transforms them in place
    RUN(dl_attn_bwd_tok(P(Q), P(K), P(QKV), P(A), P(DA), (const float*)P(LSE), P(DQKV), B, H, N, dh, sm, main));
    RUN(dl_qk_norm_rope_bwd_inplace(P(QKV), (const float*)P(QN_SCALE), (const float*)P(KN_SCALE), (const float*)P(ROPE_COS),
                                    (const float*)P(ROPE_SIN), (const float*)P(RRMS), P(DQKV), (float*)P(G_QK_SCALE),
                                    (float*)P(QK_PARTIALS), B, N, H, dh, b->rot, nullptr, main));
  } else {
    if (v_in_place)
      RUN(dl_attn_bwd_sv(P(Q), P(K), (const char*)P(QKV) + 2 * D * 2, N * 3 * D, dh, 3 * D, P(A), P(DA), (const float*)P(LSE), P(DQ), P(DK),
                         (char*)P(DQKV) + 2 * D * 2, N * 3 * D, dh, 3 * D, B, H, N, dh, sm, main));
    else
      RUN(dl_attn_bwd(P(Q), P(K), P(V), P(A), P(DA), (const float*)P(LSE), P(DQ), P(DK), P(DV), B, H, N, dh, sm, main));
    RUN(dl_qk_norm_rope_bwd(P(DQ), P(DK), v_in_place ? nullptr : P(DV), P(QKV), (const float*)P(QN_SCALE), (const float*)P(KN_SCALE),
                            (const float*)P(ROPE_COS), (const float*)P(ROPE_SIN), (const float*)P(RRMS), P(DQKV), (float*)P(G_QK_SCALE), B, N,
                            H, dh, b->rot, main));
  }
  RUN(wgrad(P(DQKV), 3 * D, P(XM1), D, P(G_QKV), 3 * D, D));
  if (grouped) {
    if (fork_to_side(main, side)) return DL_ERR_LAUNCH;
    RUN(dl_gemm_tn_group(wg, nwg, M, (float*)P(TN_SLAB), b->tn_slab_floats, side_wgs, side));
  }
  if (!rows)
    RUN(dl_gemm_nt(P(DQKV), 3 * D, P(WT_QKV), b->ldwt_3d, P(DXM), D, M, D, 3 * D, nullptr, DL_ACT_NONE, DL_BF16, nullptr, nullptr, 0,
                   nullptr, 0, 1, main));
  if (P(DFEAT)) RUN(dl_add_bf16(P(DX_MID), P(DFEAT), P(DX_MID), M * D, main));  // auxiliary-loss gradient on this block's input
  // LayerNorm 1 backward + the gated residual of the PREVIOUS block's MLP branch (absent for the first block)
  if (rows) {
    RUN(dl_ln_modulate_gemm_bwd(P(DQKV), 3 * D, P(WT_QKV), b->ldwt_3d, M, 3 * D, P(X_IN), (const float*)P(LN1_W), (const float*)P(LN1_B),
                                P(SCALE1), b->ld_mod, N, (const float*)P(MEAN1), (const float*)P(RSTD1), P(DX_MID), P(DX_OUT),
                                (float*)P(DSCALE1), (float*)P(DSHIFT1), b->ld_dmod, (float*)P(DWB1), P(PREV_T2), P(PREV_GATE2), b->ld_mod,
                                P(PREV_DT2), (float*)P(PREV_DGATE2), D, main));
  } else {
    RUN(dl_ln_modulate_bwd(P(DXM), P(X_IN), (const float*)P(LN1_W), (const float*)P(LN1_B), P(SCALE1), b->ld_mod, N,
                           (const float*)P(MEAN1), (const float*)P(RSTD1), P(DX_MID), P(DX_OUT), (float*)P(DSCALE1), (float*)P(DSHIFT1),
                           b->ld_dmod, (float*)P(DWB1), P(PREV_T2), P(PREV_GATE2), b->ld_mod, P(PREV_DT2), (float*)P(PREV_DGATE2), M, D,
                           main));
  }
  if (fold_here) RUN(fold(P(DWB1), P(G_LN1)));
  return DL_OK;
}
