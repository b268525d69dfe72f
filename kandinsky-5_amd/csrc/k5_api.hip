// k5_api.hip — kernel-level entry points of the C ABI (include/k5.h): thin extern "C" shims over the
// C++ launchers so that parity tests and host glue can call every kernel with raw device pointers.
#include "k5_kernels.h"

void k5_set_error(const char* fmt, ...);

static int ret(int st, const char* what) {
  if (st != K5_OK) {
    const hipError_t e = hipGetLastError();
    k5_set_error("%s: status %d (%s)", what, st, e == hipSuccess ? "argument/alignment" : hipGetErrorString(e));
  }
  return st;
}

extern "C" {

int k5_gemm_bf16(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lda, int ldw,
                 int ldc, int epilogue, const void* resid, int ldr, const float* gate, void* stream) {
  return ret(k5_launch_gemm_bf16(A, W, bias, C, M, N, K, lda, ldw, ldc, epilogue, resid, ldr, gate, (hipStream_t)stream),
             "k5_gemm_bf16");
}

int k5_gemm_bf16_variant(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lda, int ldw,
                         int ldc, int epilogue, const void* resid, int ldr, const float* gate, void* stream, int kernel, int token_tile) {
  if (token_tile != 0 && token_tile != 128 && token_tile != 192 && token_tile != 256) return ret(K5_ERR_ARG, "k5_gemm_bf16_variant");
  return ret(k5_launch_gemm_bf16(A, W, bias, C, M, N, K, lda, ldw, ldc, epilogue, resid, ldr, gate, (hipStream_t)stream, kernel, token_tile / 32),
             "k5_gemm_bf16_variant");
}

int k5_causal_softmax_bf16(const float* scores, void* P, int S, int hw, int lds, int ldp, void* stream) {
  if (!scores || !P || lds < S) return K5_ERR_ARG;
  return ret(k5_launch_causal_softmax(scores, P, S, hw, lds, ldp, (hipStream_t)stream), "k5_causal_softmax_bf16");
}

int k5_vae_attention512_bf16(const void* q, const void* k, const void* vt, void* o, int S, int hw, int ldqk, int ldvt, int ldo,
                             float scale, void* stream) {
  return ret(k5_launch_vae_attention512(q, k, vt, o, S, hw, ldqk, ldvt, ldo, scale, (hipStream_t)stream), "k5_vae_attention512_bf16");
}

int k5_gemm_bf16_f32out(const void* A, const void* W, float* C, int M, int N, int K, int lda, int ldw, int ldc, float alpha,
                        int causal_hw, void* stream) {
  return ret(k5_launch_gemm_bf16_f32out(A, W, C, M, N, K, lda, ldw, ldc, alpha, causal_hw, (hipStream_t)stream), "k5_gemm_bf16_f32out");
}

int k5_attention_bf16(const void* Q, const void* K, const void* Vt, void* O, int H, int q_len, int kv_len, int ldq,
                      int ldk, int ldvt, int ldo, void* stream) {
  return ret(k5_launch_attention_bf16_bounded(Q, K, Vt, O, H, q_len, kv_len, ldq, ldk, ldvt, ldo, 0.f, (hipStream_t)stream),
             "k5_attention_bf16");
}

int64_t k5_attention_state_size(int H, int q_len) { return (int64_t)k5_attention_state_bytes(H, q_len); }

int k5_attention_bf16_range(const void* Q, const void* K, const void* Vt, void* O, int H, int q_len, int kv_len, int ldq,
                            int ldk, int ldvt, int ldo, float score_bound, int tile_off0, int tile_cnt, int tile_skip_at,
                            int tile_skip_n, void* state, int flags, void* stream) {
  return ret(k5_launch_attention_bf16_range(Q, K, Vt, O, H, q_len, kv_len, ldq, ldk, ldvt, ldo, score_bound, 0, 0, tile_off0,
                                            tile_cnt, tile_skip_at, tile_skip_n, (float*)state, flags, (hipStream_t)stream),
             "k5_attention_bf16_range");
}

int k5_gemm_fp8(const void* A8, const void* W8, const float* w_scale, void* C, int M, int N, int K, int lda, int ldw, int ldc, int epi,
                const void* resid, int ldr, const float* gate, void* stream) {
  return ret(k5_launch_gemm_fp8(A8, W8, w_scale, C, M, N, K, lda, ldw, ldc, epi, resid, ldr, gate, (hipStream_t)stream), "k5_gemm_fp8");
}

int k5_quant_rows_fp8(const void* x, void* out, float* scale, int rows, int K, int ldx, int ldo, void* stream) {
  return ret(k5_launch_quant_rows_fp8(x, out, scale, rows, K, ldx, ldo, (hipStream_t)stream), "k5_quant_rows_fp8");
}

int k5_attention_bf16_prescaled(const void* Q, const void* Kc, const void* Vt, void* O, int H, int q_len, int kv_len, int ldq,
                                int ldk, int ldvt, int ldo, float score_bound, void* stream) {
  return ret(k5_launch_attention_bf16_range(Q, Kc, Vt, O, H, q_len, kv_len, ldq, ldk, ldvt, ldo, score_bound, 0, 0, 0, -1, 0x7fffffff, 0,
                                            nullptr, 0, (hipStream_t)stream, nullptr, true), "k5_attention_bf16_prescaled");
}

int k5_attention_bf16_prescaled_auto(const void* Q, const void* Kc, const void* Vt, void* O, int H, int q_len, int kv_len, int ldq,
                                     int ldk, int ldvt, int ldo, const int* head_flags, int variant, void* workspace, void* stream) {
  return ret(k5_launch_attention_bf16_range(Q, Kc, Vt, O, H, q_len, kv_len, ldq, ldk, ldvt, ldo, 0.f, 0, 0, 0, -1, 0x7fffffff, 0,
                                            nullptr, 0, (hipStream_t)stream, (float*)workspace, true, head_flags, variant),
             "k5_attention_bf16_prescaled_auto");
}

int k5_attention_flags(float* qstat, float* kstat, int nk, int kstride, int H, int force_online, int* flags, void* stream) {
  return ret(k5_launch_attn_flags(qstat, kstat, nk, kstride, H, force_online, flags, nullptr, (hipStream_t)stream), "k5_attention_flags");
}

// the same two with per-row offsets: k5_attention_flags_rows also writes kmax[H] and keeps heads up to a bound of 190 on the fixed
// form; k5_attention_bf16_prescaled_rows runs them with it (head_flags is read AND, on a late fallback, written)
int k5_attention_flags_rows(float* qstat, float* kstat, int nk, int kstride, int H, int force_online, int* flags, float* kmax, void* stream) {
  if (!kmax) return ret(K5_ERR_ARG, "k5_attention_flags_rows");
  return ret(k5_launch_attn_flags(qstat, kstat, nk, kstride, H, force_online, flags, nullptr, (hipStream_t)stream, kmax), "k5_attention_flags_rows");
}
// centred per-row offsets (K5KeyCentre): rstat = squared radii of the keys around `centre` (consumed), krad out
int k5_attention_flags_rows_centred(float* qstat, float* kstat, int nk, int kstride, int H, int force_online, int* flags, float* kmax, float* rstat,
                                    float* krad, void* stream) {
  if (!kmax || !rstat || !krad) return ret(K5_ERR_ARG, "k5_attention_flags_rows_centred");
  return ret(k5_launch_attn_flags(qstat, kstat, nk, kstride, H, force_online, flags, nullptr, (hipStream_t)stream, kmax, nullptr, rstat, krad),
             "k5_attention_flags_rows_centred");
}
int k5_attention_bf16_prescaled_rows_centred(const void* Q, const void* Kc, const void* Vt, void* O, int H, int q_len, int kv_len, int ldq,
                                             int ldk, int ldvt, int ldo, int* head_flags, const float* kmax, const float* centre, const float* krad,
                                             void* workspace, void* stream) {
  if (!head_flags || !kmax || !centre || !krad) return ret(K5_ERR_ARG, "k5_attention_bf16_prescaled_rows_centred");
  const K5KeyCentre kc{centre, krad};
  return ret(k5_launch_attention_bf16_range(Q, Kc, Vt, O, H, q_len, kv_len, ldq, ldk, ldvt, ldo, 0.f, 0, 0, 0, -1, 0x7fffffff, 0,
                                            nullptr, 0, (hipStream_t)stream, (float*)workspace, true, head_flags, K5_ATTN_AUTO, nullptr, kmax, 0, nullptr, &kc),
             "k5_attention_bf16_prescaled_rows_centred");
}
// anchored offsets for the heads beyond the window (include/k5.h): flags with the marker, the offsets, the attention that reads them
int k5_attention_flags_rows_anchored(float* qstat, float* kstat, int nk, int kstride, int H, int force_online, int* flags, float* kmax, float* rstat,
                                     float* krad, const int* prefer_online, void* stream) {
  if (!kmax || !rstat || !krad) return ret(K5_ERR_ARG, "k5_attention_flags_rows_anchored");
  return ret(k5_launch_attn_flags(qstat, kstat, nk, kstride, H, force_online, flags, nullptr, (hipStream_t)stream, kmax, prefer_online, rstat, krad, 1, 0, true),
             "k5_attention_flags_rows_anchored");
}
int k5_attention_row_anchor(const void* Q, const void* Kc, int H, int q_len, int kv_len, int ldq, int ldk, int key0, int kv_total, const float* kmax,
                            float* anchor, void* stream) {
  return ret(k5_launch_attn_row_anchor(Q, Kc, H, q_len, kv_len, ldq, ldk, key0, kv_total, kmax, anchor, (hipStream_t)stream), "k5_attention_row_anchor");
}
int k5_attention_bf16_prescaled_rows_anchored(const void* Q, const void* Kc, const void* Vt, void* O, int H, int q_len, int kv_len, int ldq,
                                              int ldk, int ldvt, int ldo, int* head_flags, const float* kmax, const float* centre, const float* krad,
                                              const float* anchor, void* workspace, void* stream) {
  if (!head_flags || !kmax || !centre || !krad || !anchor) return ret(K5_ERR_ARG, "k5_attention_bf16_prescaled_rows_anchored");
  const K5KeyCentre kc{centre, krad, anchor};
  return ret(k5_launch_attention_bf16_range(Q, Kc, Vt, O, H, q_len, kv_len, ldq, ldk, ldvt, ldo, 0.f, 0, 0, 0, -1, 0x7fffffff, 0,
                                            nullptr, 0, (hipStream_t)stream, (float*)workspace, true, head_flags, K5_ATTN_AUTO, nullptr, kmax, 0, nullptr, &kc),
             "k5_attention_bf16_prescaled_rows_anchored");
}
int k5_attention_bf16_prescaled_rows(const void* Q, const void* Kc, const void* Vt, void* O, int H, int q_len, int kv_len, int ldq,
                                     int ldk, int ldvt, int ldo, int* head_flags, const float* kmax, void* workspace, void* stream) {
  if (!head_flags || !kmax) return ret(K5_ERR_ARG, "k5_attention_bf16_prescaled_rows");
  return ret(k5_launch_attention_bf16_range(Q, Kc, Vt, O, H, q_len, kv_len, ldq, ldk, ldvt, ldo, 0.f, 0, 0, 0, -1, 0x7fffffff, 0,
                                            nullptr, 0, (hipStream_t)stream, (float*)workspace, true, head_flags, K5_ATTN_AUTO, nullptr, kmax),
             "k5_attention_bf16_prescaled_rows");
}

// one pass of a multi-pass schedule with per-row offsets (what the sequence-parallel engine runs): key tiles tile_off0 .. +tile_cnt,
// flags & 1 resume / & 2 leave the state, late_pass 1 = not the last pass, 2 = the last (see include/k5.h)
int k5_attention_bf16_prescaled_rows_pass(const void* Q, const void* Kc, const void* Vt, void* O, int H, int q_len, int kv_len, int ldq,
                                          int ldk, int ldvt, int ldo, int* head_flags, const float* kmax, int tile_off0, int tile_cnt,
                                          float* state, int flags, int late_pass, void* workspace, void* stream) {
  if (!head_flags || !kmax) return ret(K5_ERR_ARG, "k5_attention_bf16_prescaled_rows_pass");
  return ret(k5_launch_attention_bf16_range(Q, Kc, Vt, O, H, q_len, kv_len, ldq, ldk, ldvt, ldo, 0.f, 0, 0, tile_off0, tile_cnt, 0x7fffffff, 0,
                                            state, flags, (hipStream_t)stream, (float*)workspace, true, head_flags, K5_ATTN_AUTO, nullptr, kmax,
                                            late_pass), "k5_attention_bf16_prescaled_rows_pass");
}

// the same pass with norm_qk + apply_rotary of the QUERIES fused into the kernel's Q load (K5QueryNorm): Q holds the raw projection.
// head_flags + kmax (k5_attention_flags_rows with a ZERO query statistic: every head starts on the fixed form) -> the fixed-offset
// workgroups decide per head (a row bound above 190 flips the flag to 0); both null -> online max everywhere.
int k5_attention_bf16_prescaled_qnorm_pass(const void* Q, const void* Kc, const void* Vt, void* O, int H, int q_len, int kv_len, int ldq,
                                           int ldk, int ldvt, int ldo, const float* q_norm_w, const float* q_cos, const float* q_sin,
                                           int* head_flags, const float* kmax, int tile_off0, int tile_cnt, float* state, int flags,
                                           int late_pass, void* workspace, void* stream) {
  if (!q_norm_w || !q_cos || !q_sin || (!head_flags != !kmax)) return ret(K5_ERR_ARG, "k5_attention_bf16_prescaled_qnorm_pass");
  const K5QueryNorm qn{q_norm_w, q_cos, q_sin, nullptr};
  return ret(k5_launch_attention_bf16_range(Q, Kc, Vt, O, H, q_len, kv_len, ldq, ldk, ldvt, ldo, 0.f, 0, 0, tile_off0, tile_cnt, 0x7fffffff, 0,
                                            state, flags, (hipStream_t)stream, (float*)workspace, true, head_flags,
                                            head_flags ? K5_ATTN_AUTO : K5_ATTN_ONLINE, nullptr, kmax, late_pass, &qn),
             "k5_attention_bf16_prescaled_qnorm_pass");
}

int64_t k5_attention_balance_size(int H, int q_len) { return (int64_t)k5_attention_balance_bytes(H, q_len); }

int k5_attention_bf16_balanced(const void* Q, const void* K, const void* Vt, void* O, int H, int q_len, int kv_len, int ldq,
                               int ldk, int ldvt, int ldo, float score_bound, void* workspace, void* stream) {
  if (!workspace) return ret(K5_ERR_ARG, "k5_attention_bf16_balanced");
  return ret(k5_launch_attention_bf16_range(Q, K, Vt, O, H, q_len, kv_len, ldq, ldk, ldvt, ldo, score_bound, 0, 0, 0, -1, 0x7fffffff, 0,
                                            nullptr, 0, (hipStream_t)stream, (float*)workspace), "k5_attention_bf16_balanced");
}

int64_t k5_nabla_workspace_size(int H, int num_blocks) { return (int64_t)k5_nabla_workspace_bytes(H, num_blocks); }

int k5_nabla_select_bf16(const void* q, const void* k, int ldq, int ldk, int H, int N, int T, int Hb, int Wb, int wT, int wH,
                         int wW, float P, void* workspace, void* stream) {
  return ret(k5_launch_nabla_select(q, k, ldq, ldk, H, N, T, Hb, Wb, wT, wH, wW, P, workspace, (hipStream_t)stream),
             "k5_nabla_select_bf16");
}

int k5_attention_nabla_bf16(const void* Q, const void* K, const void* Vt, void* O, int H, int N, int ldq, int ldk, int ldvt,
                            int ldo, float score_bound, const void* workspace, void* stream) {
  if (!workspace || N <= 0 || (N % 64)) return ret(K5_ERR_ARG, "k5_attention_nabla_bf16");
  const int *list, *cnt;
  k5_nabla_workspace_views(const_cast<void*>(workspace), H, N / 64, nullptr, nullptr, &list, &cnt);
  return ret(k5_launch_attention_bf16_sparse(Q, K, Vt, O, H, N, N, ldq, ldk, ldvt, ldo, score_bound, list, cnt, N / 64, 0, 0,
                                             (hipStream_t)stream), "k5_attention_nabla_bf16");
}

int k5_nabla_mask_u8(const void* workspace, int H, int num_blocks, void* out_u8, void* stream) {
  return ret(k5_launch_nabla_mask_u8(workspace, H, num_blocks, num_blocks, out_u8, (hipStream_t)stream), "k5_nabla_mask_u8");
}

int k5_nabla_select_rect_bf16(const void* q, const void* k, int ldq, int ldk, int H, int Nq, int q_block0, int N, int T, int Hb,
                              int Wb, int wT, int wH, int wW, float P, void* workspace, void* stream) {
  return ret(k5_launch_nabla_select_rect(q, k, ldq, ldk, H, Nq, q_block0, N, T, Hb, Wb, wT, wH, wW, P, workspace,
                                         (hipStream_t)stream), "k5_nabla_select_rect_bf16");
}

int k5_attention_nabla_rect_bf16(const void* Q, const void* K, const void* Vt, void* O, int H, int Nq, int N, int ldq, int ldk,
                                 int ldvt, int ldo, float score_bound, const void* workspace, int vt_chunk_keys,
                                 int64_t vt_chunk_stride, void* stream) {
  if (!workspace || N <= 0 || (N % 64) || Nq <= 0 || (Nq % 64)) return ret(K5_ERR_ARG, "k5_attention_nabla_rect_bf16");
  const int *list, *cnt;
  k5_nabla_workspace_views(const_cast<void*>(workspace), H, N / 64, nullptr, nullptr, &list, &cnt, nullptr, Nq / 64);   // as k5_nabla_select_rect_bf16 laid it out
  return ret(k5_launch_attention_bf16_sparse(Q, K, Vt, O, H, Nq, N, ldq, ldk, ldvt, ldo, score_bound, list, cnt, N / 64,
                                             vt_chunk_keys, (long long)vt_chunk_stride, (hipStream_t)stream),
             "k5_attention_nabla_rect_bf16");
}

// what the sequence-parallel engine runs: the map with the rank's own key blocks leading every list, and the list-driven attention on
// pre-scaled keys with per-head flags / per-row offsets — in one pass (pass = 0) or two (1: the leading local entries, state out;
// 2: the rest, state in, normalise; a head whose row underflows in pass 1 goes late and is recomputed by the online form of pass 2)
int k5_nabla_select_rect_local_bf16(const void* q, const void* k, int ldq, int ldk, int H, int Nq, int q_block0, int N, int T, int Hb,
                                    int Wb, int wT, int wH, int wW, float P, void* workspace, int local_block0, int local_blocks,
                                    void* stream) {
  return ret(k5_launch_nabla_select_rect(q, k, ldq, ldk, H, Nq, q_block0, N, T, Hb, Wb, wT, wH, wW, P, workspace,
                                         (hipStream_t)stream, local_block0, local_blocks), "k5_nabla_select_rect_local_bf16");
}
int k5_attention_nabla_rect_prescaled_pass(const void* Q, const void* Kc, const void* Vt, void* O, int H, int Nq, int N, int ldq, int ldk,
                                           int ldvt, int ldo, const void* workspace, int vt_chunk_keys, int64_t vt_chunk_stride,
                                           int* head_flags, const float* kmax, int pass, float* state, void* stream) {
  if (!workspace || N <= 0 || (N % 64) || pass < 0 || pass > 2 || (pass && !state) || !head_flags || !kmax)
    return ret(K5_ERR_ARG, "k5_attention_nabla_rect_prescaled_pass");
  const int *list, *cnt, *cnt_local;
  if (Nq <= 0 || (Nq % 64)) return ret(K5_ERR_ARG, "k5_attention_nabla_rect_prescaled_pass");
  k5_nabla_workspace_views(const_cast<void*>(workspace), H, N / 64, nullptr, nullptr, &list, &cnt, &cnt_local, Nq / 64);
  const K5SparsePass p1{nullptr, state, 2, 1}, p2{cnt_local, state, 1, 2};
  return ret(k5_launch_attention_bf16_sparse(Q, Kc, Vt, O, H, Nq, N, ldq, ldk, ldvt, ldo, 0.f, list, pass == 1 ? cnt_local : cnt, N / 64,
                                             vt_chunk_keys, (long long)vt_chunk_stride, (hipStream_t)stream, true, head_flags, K5_ATTN_AUTO,
                                             kmax, pass == 0 ? nullptr : (pass == 1 ? &p1 : &p2)),
             "k5_attention_nabla_rect_prescaled_pass");
}

int k5_nabla_mask_rect_u8(const void* workspace, int H, int q_blocks, int num_blocks, void* out_u8, void* stream) {
  return ret(k5_launch_nabla_mask_u8(workspace, H, q_blocks, num_blocks, out_u8, (hipStream_t)stream), "k5_nabla_mask_rect_u8");
}

int k5_attention_bf16_bounded(const void* Q, const void* K, const void* Vt, void* O, int H, int q_len, int kv_len,
                              int ldq, int ldk, int ldvt, int ldo, float score_bound, void* stream) {
  return ret(k5_launch_attention_bf16_bounded(Q, K, Vt, O, H, q_len, kv_len, ldq, ldk, ldvt, ldo, score_bound,
                                              (hipStream_t)stream), "k5_attention_bf16_bounded");
}

int k5_ln_modulate_bf16(const void* x, const float* scale, const float* shift, void* out, int rows, int D, int ldx,
                        int ldo, void* stream) {
  return ret(k5_launch_ln_modulate(x, scale, shift, out, rows, D, ldx, ldo, (hipStream_t)stream), "k5_ln_modulate_bf16");
}

int k5_rmsnorm_rope_bf16(void* x, const float* weight, const float* cos_tab, const float* sin_tab, int rows, int H,
                         int ld, int heads_per_weight, int rope_heads, void* stream) {
  const int32_t hc[2] = {heads_per_weight, rope_heads};
  return ret(k5_launch_rmsnorm_rope(x, weight, cos_tab, sin_tab, rows, H, ld, hc, (hipStream_t)stream),
             "k5_rmsnorm_rope_bf16");
}

int k5_rmsnorm_rope_stats_bf16(void* x, const float* weight, const float* cos_tab, const float* sin_tab, int rows, int H, int ld,
                               int heads_per_weight, int rope_heads, float out_scale, int scale_from_head, float* stats, void* stream) {
  const int32_t hc[2] = {heads_per_weight, rope_heads};
  static float* ws = nullptr; static size_t ws_bytes = 0;   // scratch of the kernel-level entry (the engine owns its own)
  const size_t need = k5_rmsnorm_stats_workspace_bytes(H);
  if (stats && need > ws_bytes) {
    if (ws) (void)hipFree(ws);
    ws = nullptr; ws_bytes = 0;
    if (hipMalloc((void**)&ws, need) != hipSuccess) return ret(K5_ERR_HIP, "k5_rmsnorm_rope_stats_bf16");
    ws_bytes = need;
  }
  return ret(k5_launch_rmsnorm_rope(x, weight, cos_tab, sin_tab, rows, H, ld, hc, (hipStream_t)stream, out_scale, scale_from_head,
                                    nullptr, 0, stats, ws), "k5_rmsnorm_rope_stats_bf16");
}

// the same with the centred statistics: centre [H - scale_from_head][64] out, stats = H squared norms then H - scale_from_head squared radii
int k5_rmsnorm_rope_centre_bf16(void* x, const float* weight, const float* cos_tab, const float* sin_tab, int rows, int H, int ld,
                                int heads_per_weight, int rope_heads, float out_scale, int scale_from_head, float* stats, float* centre, void* stream) {
  if (!stats || !centre || scale_from_head < 0 || scale_from_head >= H) return ret(K5_ERR_ARG, "k5_rmsnorm_rope_centre_bf16");
  const int32_t hc[2] = {heads_per_weight, rope_heads};
  static float* ws = nullptr; static size_t ws_bytes = 0;
  const size_t need = k5_rmsnorm_stats_workspace_bytes(H);
  if (need > ws_bytes) {
    if (ws) (void)hipFree(ws);
    ws = nullptr; ws_bytes = 0;
    if (hipMalloc((void**)&ws, need) != hipSuccess) return ret(K5_ERR_HIP, "k5_rmsnorm_rope_centre_bf16");
    ws_bytes = need;
  }
  return ret(k5_launch_rmsnorm_rope(x, weight, cos_tab, sin_tab, rows, H, ld, hc, (hipStream_t)stream, out_scale, scale_from_head,
                                    nullptr, 0, stats, ws, centre), "k5_rmsnorm_rope_centre_bf16");
}

int k5_gate_sum_bf16(const void* x, const void* y, const float* gate, void* out, int rows, int D, void* stream) {
  return ret(k5_launch_gate_sum(x, y, gate, out, rows, D, (hipStream_t)stream), "k5_gate_sum_bf16");
}

int k5_gemv_f32(const float* x, const float* W, const float* b, float* y, int N, int K, int silu_in, const float* add,
                void* stream) {
  return ret(k5_launch_gemv_f32(x, W, b, y, N, K, silu_in, add, (hipStream_t)stream), "k5_gemv_f32");
}

int k5_time_features_f32(float t, float* out, int D, void* stream) {
  return ret(k5_launch_time_features(t, out, D, (hipStream_t)stream), "k5_time_features_f32");
}

int k5_ln_affine_bf16(const void* x, const float* w, const float* b, void* out_bf16, float* out_f32, int rows, int D,
                      void* stream) {
  return ret(k5_launch_ln_affine(x, w, b, out_bf16, out_f32, rows, D, (hipStream_t)stream), "k5_ln_affine_bf16");
}

int k5_rope_table_f32(float* cos_tab, float* sin_tab, const int32_t* pos_t, const int32_t* pos_h, const int32_t* pos_w,
                      int T, int H, int W, int n0, int n1, int n2, float s0, float s1, float s2,
                      const int32_t* tok_perm, void* stream) {
  return ret(k5_launch_rope_table(cos_tab, sin_tab, pos_t, pos_h, pos_w, T, H, W, n0, n1, n2, s0, s1, s2, tok_perm,
                                  (hipStream_t)stream), "k5_rope_table_f32");
}

int k5_patchify_bf16(const float* x, void* out, int T, int H, int W, int x_channels, int Cin_total, int Kpad,
                     const int32_t* tok_perm, void* stream) {
  return ret(k5_launch_patchify(x, out, T, H, W, x_channels, Cin_total, Kpad, tok_perm, (hipStream_t)stream),
             "k5_patchify_bf16");
}

int k5_unpatchify_bf16(const void* x, void* out, int T, int Hp, int Wp, int C, int ldx, const int32_t* tok_perm,
                       void* stream) {
  return ret(k5_launch_unpatchify(x, out, T, Hp, Wp, C, ldx, tok_perm, (hipStream_t)stream), "k5_unpatchify_bf16");
}

int k5_cfg_euler(float* img, const void* v_cond, const void* v_uncond, float w, float dt, int64_t n, void* stream) {
  return ret(k5_launch_cfg_euler(img, v_cond, v_uncond, w, dt, n, (hipStream_t)stream), "k5_cfg_euler");
}

int k5_conv3d_bf16(const void* X, const void* W, const float* bias, void* out, int Ts, int Hs, int Ws, int Cin, int Cout,
                   int up_t, int up_s, int ldc, const void* resid, int ldr, void* stream) {
  return ret(k5_launch_conv3d_bf16(X, W, bias, out, Ts, Hs, Ws, Cin, Cout, up_t, up_s, ldc, resid, ldr, (hipStream_t)stream),
             "k5_conv3d_bf16");
}

int k5_conv3d_strided_bf16(const void* X, const void* W, const float* bias, void* out, int Ts, int Hs, int Ws, int Cin, int Cout,
                           int st_t, int st_s, int ldc, void* stream) {
  return ret(k5_launch_conv3d_bf16_strided(X, W, bias, out, Ts, Hs, Ws, Cin, Cout, 1, 1, st_t, st_s, ldc, nullptr, 0, (hipStream_t)stream),
             "k5_conv3d_strided_bf16");
}

int64_t k5_conv3d_stats_size(int M, int Cout) { return (int64_t)2 * ((M + 255) / 256) * (Cout / 4) * 2 * (int64_t)sizeof(float); }

int k5_conv3d_bf16_stats(const void* X, const void* W, const float* bias, void* out, int Ts, int Hs, int Ws, int Cin, int Cout,
                         int up_t, int up_s, int ldc, const void* resid, int ldr, float* quad_stats, void* stream) {
  if (!quad_stats || (Cout & 3)) return K5_ERR_ARG;
  const int st = k5_launch_conv3d_w4(X, W, bias, out, Ts, Hs, Ws, Cin, Cout, up_t, up_s, ldc, resid, ldr, quad_stats, (hipStream_t)stream);
  return st == K5_ERR_UNSUPPORTED ? st : ret(st, "k5_conv3d_bf16_stats");   // outside the 4-wave kernel's range: no statistics, caller's business
}

int k5_groupnorm_bf16_quads(const void* x, const float* gamma, const float* beta, void* out, int M, int C, int G, float eps, int silu,
                            const float* quad_stats, void* workspace, void* stream) {
  if (!quad_stats || !workspace) return K5_ERR_ARG;
  return ret(k5_launch_groupnorm_bf16_quads(x, gamma, beta, out, M, C, G, eps, silu, C, C, quad_stats, 2 * ((M + 255) / 256), (float*)workspace,
                                            (hipStream_t)stream), "k5_groupnorm_bf16_quads");
}

int64_t k5_groupnorm_workspace_size(int M, int G) { return (int64_t)k5_groupnorm_workspace_bytes(M, G); }

int k5_groupnorm_bf16(const void* x, const float* gamma, const float* beta, void* out, int M, int C, int G, float eps,
                      int silu, void* workspace, void* stream) {
  return ret(k5_launch_groupnorm_bf16(x, gamma, beta, out, M, C, G, eps, silu, C, C, workspace, (hipStream_t)stream),
             "k5_groupnorm_bf16");
}

}  // extern "C"
