// k5_kernels.h — internal C++ launchers shared by the C-ABI layer (k5_api.hip) and the engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "k5.h"  // status codes, epilogue ids (public C ABI)

int k5_launch_gemm_bf16(const void* A, const void* W, const float* bias, void* C, int M, int N, int K,
                        int lda, int ldw, int ldc, int epi, const void* resid, int ldr,
                        const float* gate, hipStream_t stream,
                        int force_kernel = 0,    // K5_GEMM_V1's numbering: 2 = the 128 x 128 direct-to-LDS kernel whatever the shape (callers that must keep a summation order)
                        int force_mt = 0);       // four-wave kernel: 16-row token tiles per wave, 8 / 6 / 4 = 256- / 192- / 128-row workgroup tiles (0 = by cost)

// split-K tail of the four-wave GEMM (round 6): 0 = whole tiles everywhere (one K order: bit-identical across kernels and tile heights — what force_kernel
// 2 / 4 / 5 / 8 always give), 1 = the ragged last round of a launch is cut along K where at most half of the CUs would be busy, 2 = wherever a
// tile can be cut (default 0: measured neutral, csrc/gemm_bf16.hip); -1 = back to K5_GEMM_SK / the built-in default.  Process-wide (the GEMM launcher has no handle).
void k5_gemm_set_stream_k_default(int mode);
void k5_gemm_set_stream_k_thread(int mode);
int k5_gemm_stream_k_policy();

// Attention: O[q][h*64+d] = softmax(Q K^T / 8) V, bf16, head_dim 64, non-causal.
//   Q  [q_len][ldq]  (head h at columns h*64..), K [kv_len][ldk], Vt [H*64][ldvt] = V transposed, O [q_len][ldo].
// Dense attention with a caller-proved bound |q.k| <= score_bound (0 = unknown -> online running max).
int k5_launch_attention_bf16_bounded(const void* Q, const void* K, const void* Vt, void* O, int H, int q_len,
                                     int kv_len, int ldq, int ldk, int ldvt, int ldo, float score_bound,
                                     hipStream_t stream);

// Same, with V^T stored as per-rank chunks (sequence parallelism): keys [c*chunk_keys, (c+1)*chunk_keys) live at
// Vt + c*chunk_stride, row stride ldvt.
int k5_launch_attention_bf16_chunked(const void* Q, const void* K, const void* Vt, void* O, int H, int q_len,
                                     int kv_len, int ldq, int ldk, int ldvt, int ldo, float score_bound,
                                     int vt_chunk_keys, long long vt_chunk_stride, hipStream_t stream);

// General form: process key tiles  e -> e + tile_off0 (+ tile_skip_n once >= tile_skip_at), e < tile_cnt (-1 = to the end);
// flags & 1: resume from `state`, flags & 2: write `state` instead of O (k5_attention_state_bytes floats-as-bytes).
size_t k5_attention_state_bytes(int H, int q_len);
// segmented walk instead of the (offset, skip) one: position e -> tile_off0 + seg(e / len) * stride + e % len, where segments
// >= skip shift up by one (the sequence-parallel schedule's "slice s of every rank's slot except mine"); tile_cnt positions
struct K5TileSegments { int len, stride, skip; };
// one pass of a two-pass walk of the NABLA lists (pre-scaled keys): list positions [begin[w], cnt[w]) of workgroup w (begin null: 0);
// state (k5_attention_state_bytes) is resumed with flags & 1 and left — no output — with flags & 2; late_pass as in the range launcher
struct K5SparsePass { const int* begin; float* state; int flags; int late_pass; };
// norm_qk (+ apply_rotary) of the QUERY rows fused into the attention kernel's Q-fragment load: Q then holds the raw projection.
// w: 64 RMSNorm weights.  cos / sin null: cross-attention (unscaled keys; K5_ERR_UNSUPPORTED unless score_bound selects the fixed-offset
// kernel).  cos / sin [row][32] fp32: visual self-attention (pre-scaled keys); with row_offset_kmax the fixed-offset workgroups then
// decide per head themselves (a row bound above 190 flips the head's flag to the online form; counters [fixed, online] follow).
struct K5QueryNorm { const float* w; const float* cos; const float* sin; unsigned long long* counters; };
// Centred form of the per-row offsets (AttnP::kcentre): centre [H][64] = a convex combination of each head's keys (k5_launch_rmsnorm_rope
// key_centre), radius [H] = max |k' - centre| with margin (k5_launch_attn_flags krad_out).  Rows whose plain bound |q| kmax exceeds 90 run
// with the offset q.c + |q| R - 90: no overflow, and no row-sum underflow while |q| R <= 190 whatever common component the scores carry.
// row_anchor (nullable, [H][q_len]): anchored offsets of the heads k5_launch_attn_flags(anchored) marked with a negative kmax entry
// (k5_launch_attn_row_anchor) — AttnP::row_anchor
struct K5KeyCentre { const float* centre; const float* radius; const float* row_anchor = nullptr; };
int k5_launch_attention_bf16_range(const void* Q, const void* K, const void* Vt, void* O, int H, int q_len,
                                   int kv_len, int ldq, int ldk, int ldvt, int ldo, float score_bound,
                                   int vt_chunk_keys, long long vt_chunk_stride, int tile_off0, int tile_cnt, int tile_skip_at,
                                   int tile_skip_n, float* state, int flags, hipStream_t stream, float* balance_ws = nullptr, bool k_prescaled = false,
                                   const int* head_flags = nullptr, int variant = 0, const K5TileSegments* segments = nullptr,
                                   const float* row_offset_kmax = nullptr, int late_pass = 0,   // late_pass: AttnP::late_pass (multi-pass + per-row offsets)
                                   const K5QueryNorm* query_norm = nullptr,   // fused norm_qk (+ RoPE) of the query rows, see K5QueryNorm
                                   const K5KeyCentre* key_centre = nullptr);   // centred per-row offsets (with row_offset_kmax), see K5KeyCentre
size_t k5_attention_balance_bytes(int H, int q_len);
// softmax form of the pre-scaled-key launches: AUTO = fixed offset where the bound (score_bound, or the per-head device
// flags) allows it and the lazy online max elsewhere; ONLINE = the online max everywhere
enum { K5_ATTN_AUTO = 0, K5_ATTN_ONLINE = 1 };
// per-head flags from the |q|^2 / |k'|^2 maxima that k5_launch_rmsnorm_rope(stats) left (consumed: reset to 0); kstat holds
// nk partial maxima at stride kstride floats; counters (optional, device u64[2]) += heads sent to {fixed, online}
// kmax_out (nullable, H floats): max |k'_h| with margin for the per-row offsets of the fixed-offset form (row_offset_kmax of the
// attention launchers); with it heads up to a bound of 190 (instead of 90) keep that form — a row whose sum underflows sends its
// head to the online form late (the flag is rewritten by the attention kernel)
// prefer_online (nullable, H ints, with kmax_out only): heads to send to the online form although their bound is within the per-row-offset
// window — k5_launch_attn_pref_update sets the entry of a head more than a quarter of whose (head, query block) jobs fell back the
// last time (job flags at the end of the balance workspace that run's attention launches used; group_rows 2: 128-query jobs)
int k5_launch_attn_flags(float* qstat, float* kstat, int nk, int kstride, int H, int force_online, int* flags,
                         unsigned long long* counters, hipStream_t stream, float* kmax_out = nullptr, const int* prefer_online = nullptr,
                         float* rstat = nullptr, float* krad_out = nullptr,   // squared radii in (consumed) / radii with margin out: K5KeyCentre::radius
                         int nq = 1, int qstride = 0,                          // qstat as nq partial maxima at stride qstride (Ulysses)
                         bool anchored = false,    // heads beyond the window: fixed form on anchored offsets (kmax_out entry < 0) instead of the online form
                         unsigned int* leave_sig = nullptr);   // nullable: set to 1 when any head leaves the plain fixed-offset form (online or anchored)
int k5_launch_attn_row_anchor(const void* Q, const void* Kc, int H, int q_len, int kv_len, int ldq, int ldk, int key0, int kv_total,
                              const float* kmax, float* out, hipStream_t stream);
int k5_launch_attn_pref_update(float* balance_ws, int H, int q_len, int group_rows, int* prefer_online, hipStream_t stream,
                               unsigned int* leave_sig = nullptr);   // nullable: set to 1 when a head is marked

// ---- fp8 (e4m3) feed-forward path, opt-in (gemm_fp8.hip) ----
int k5_launch_gemm_fp8(const void* A8, const void* W8, const float* w_scale, void* C, int M, int N, int K, int lda, int ldw, int ldc,
                       int epi, const void* resid, int ldr, const float* gate, hipStream_t stream, const float* bias = nullptr, int scale_m = 0);
int k5_launch_quant_rows_fp8(const void* x, void* out, float* scale, int rows, int K, int ldx, int ldo, hipStream_t stream);

// ---- NABLA (block-sparse) ----
size_t k5_nabla_workspace_bytes(int H, int nb, int nqb = 0, int list_rows = 0);   // nqb: query-block rows selected here (0 = all nb); list_rows: key-tile lists per head (0 = one per row)
int k5_launch_nabla_select(const void* q, const void* k, int ldq, int ldk, int H, int N, int T, int Hb, int Wb, int wT, int wH,
                           int wW, float P, void* workspace, hipStream_t s);
void k5_nabla_workspace_views(void* workspace, int H, int nb, const unsigned long long** bits, const int** kv_nb, const int** list,
                              const int** cnt, const int** cnt_local = nullptr, int nqb = 0);   // nqb as given to k5_nabla_workspace_bytes (the lists sit behind that many rows of logits)
int k5_launch_nabla_select_rect(const void* q, const void* k, int ldq, int ldk, int H, int Nq, int q_block0, int N, int T, int Hb,
                                int Wb, int wT, int wH, int wW, float P, void* workspace, hipStream_t s,
                                int local_block0 = 0, int local_blocks = 0,   // > 0: these key blocks lead every list (cnt_local of them)
                                int group_rows = 4,                           // 64-query rows per list (4, or 2 for the 128-query workgroups)
                                int pair_stride = 0);                         // group_rows 2: which two rows share a list (k5_pair_row; 0 = adjacent)
// sequence parallelism: a rank's key-block means (k5_launch_nabla_block_means into its slot of a [P][H][slot_blocks][64] buffer),
// gathered, re-laid into the workspace (k5_launch_nabla_key_means_from_slots); k5_launch_nabla_select_rect(k = nullptr) selects from them
int k5_launch_nabla_block_means(const void* x, int ld, int H, int nblocks, int stride_blocks, void* out, hipStream_t s);
void k5_nabla_workspace_means(void* workspace, int H, int nb, void** qa, void** ka);   // the means' regions (k5_launch_rmsnorm_rope mean_q / mean_k write them directly)
int k5_launch_nabla_key_means_from_slots(const void* gathered, int H, int nb, int slot_blocks, void* workspace, hipStream_t s);
int k5_launch_nabla_mask_u8(const void* workspace, int H, int nqb, int nb, void* out, hipStream_t s);
// *acc += number of kept (query block, key block) pairs of the map in `workspace` (H x nqb rows)
int k5_launch_nabla_count(const void* workspace, int H, int nqb, int nb, unsigned long long* acc, hipStream_t s);
int k5_launch_nabla_count_lists(const void* workspace, int H, int nqb, int nb, int group_rows, unsigned long long* acc, hipStream_t s);
int k5_launch_attention_bf16_sparse(const void* Q, const void* K, const void* Vt, void* O, int H, int q_len, int kv_len, int ldq,
                                    int ldk, int ldvt, int ldo, float score_bound, const int* list, const int* cnt, int list_stride,
                                    int vt_chunk_keys, long long vt_chunk_stride, hipStream_t stream, bool k_prescaled = false,
                                    const int* head_flags = nullptr, int variant = 0, const float* row_offset_kmax = nullptr,
                                    const K5SparsePass* pass = nullptr, float* balance_ws = nullptr,   // k5_attention_balance_bytes; pre-scaled keys
                                    int group_rows = 4,    // 2: lists per TWO 64-query rows (k5_launch_nabla_select_rect group_rows = 2), 128-query workgroups
                                    bool balance = true,   // false: balance_ws only carries the per-job fallback flags of the per-row-offset form
                                    const K5KeyCentre* key_centre = nullptr,
                                    int pair_stride = 0);  // group_rows 2: rows of a group per k5_pair_row (the stride the lists were built with)

// K1: out = bf16( LayerNorm(x; eps 1e-5, no affine) * (scale + 1) + shift )
// out_e4m3 (nullable, [rows][D] bytes): also — or, with out == nullptr, only — e4m3(bf16(.)) at the static scale 1: the activation operand of
// the fp8 GEMMs without a separate quantisation pass (what k5_launch_quant_rows_fp8(scale = nullptr) makes of the bf16 rows, bit for bit)
int k5_launch_ln_modulate(const void* x, const float* scale, const float* shift, void* out, int rows,
                          int D, int ldx, int ldo, hipStream_t stream, void* out_e4m3 = nullptr);
// K5 + K3: in place over [rows][ld] (H heads of 64): RMSNorm(eps, weight) -> bf16 -> RoPE (optional)
//   x holds H heads per row; head h uses weight[(h / heads_per_weight)*64 ..]; RoPE (cos/sin [rows][32]) on heads
//   < rope_heads.  heads_cfg = host pointer to {heads_per_weight, rope_heads} or null (= {H, H}).
int k5_launch_rmsnorm_rope(void* x, const float* weight, const float* cos, const float* sin, int rows,
                           int H, int ld, const int32_t* heads_cfg, hipStream_t stream,
                           float out_scale = 1.f, int scale_from_head = 0x7fffffff, void* scaled_out = nullptr, int ld_scaled = 0,
                           float* stats = nullptr, float* stats_ws = nullptr,
                           float* key_centre = nullptr,    // OUT [H - scale_from_head][64]: sample-mean key per scaled head; stats then also gets H - scale_from_head squared radii |k' - c|^2
                           // NABLA (rows % 64 == 0): 64-token block means of the UNSCALED normalised + rotated heads, taken in this pass (what
                           // k5_launch_nabla_block_means computes from the stored tensor): heads < scale_from_head -> mean_q [head][mq_stride][64] bf16,
                           // the others -> mean_k [head - scale_from_head][mk_stride][64]
                           void* mean_q = nullptr, int mq_stride = 0, void* mean_k = nullptr, int mk_stride = 0);
size_t k5_rmsnorm_stats_workspace_bytes(int H);
// Ulysses sequence parallelism: (q | k) rows [rows][2 D] -> per-destination blocks [P][slot_rows][2 D / P]; outputs [P][slot_rows][D / P] -> [rows][D]
int k5_launch_ulysses_pack_qk(const void* x, void* out, int rows, int slot_rows, int D, int P, hipStream_t s);
int k5_launch_ulysses_unpack_o(const void* in, void* out, int rows, int slot_rows, int D, int P, hipStream_t s);   // stats_ws: scratch of this size whenever stats is given
// heads >= scale_from_head are multiplied by out_scale before the bf16 rounding — in place, or (scaled_out != null) into
// scaled_out[row][(head - scale_from_head) * 64 ...] while the unscaled values stay in place.
// stats (device, [H] floats, zeroed by the consumer): stats[h] = max(stats[h], |x_row,h|^2) over the rows of the call, of the
// values the softmax reads (the scaled ones for heads >= scale_from_head) -> data-derived softmax bound (k5_launch_attn_flags)
// K15: cos/sin tables [T*H*W][n0+n1+n2] for RoPE3D (RoPE1D: H=W=1, n1=n2=0), optional token permutation
int k5_launch_rope_table(float* cosT, float* sinT, const int32_t* p0, const int32_t* p1, const int32_t* p2, int T,
                         int H, int W, int n0, int n1, int n2, float s0, float s1, float s2, const int32_t* tok_perm,
                         hipStream_t stream);
// K2 (standalone form): out = bf16(x + gate * y)
int k5_launch_gate_sum(const void* x, const void* y, const float* gate, void* out, int rows, int D,
                       hipStream_t stream);
// fp32 GEMV with optional SiLU on the input: y[n] = sum_k act(x[k]) * W[n][k] + b[n]   (K11, K12)
int k5_launch_gemv_f32(const float* x, const float* W, const float* b, float* y, int N, int K, int silu_in,
                       const float* add, hipStream_t stream);
// sinusoidal time features (K12): out[0..D/2) = cos(t f_i), out[D/2..D) = sin(t f_i)
// tvec/step (both or neither): read the time from tvec[*step] on the device (hipGraph-replayable sampler step)
int k5_launch_time_features(float t, float* out, int D, hipStream_t stream, const float* tvec = nullptr, const int* step = nullptr);
int k5_launch_step_inc(int* step, hipStream_t stream);
// LayerNorm with affine over bf16 rows (K13): out = bf16(LN(x) * w + b); also fp32 output option
int k5_launch_ln_affine(const void* x, const float* w, const float* b, void* out_bf16, float* out_f32, int rows,
                        int D, hipStream_t stream);
// patchify (K14) fp32 latent (T,H,W,C) [+ zero visual_cond channels] -> bf16 [Ntok][Kpad], optional
// token permutation (fractal order, K16)
int k5_launch_patchify(const float* x, void* out, int T, int H, int W, int C, int Cin_total, int Kpad,
                       const int32_t* tok_perm, hipStream_t stream);
// un-patchify (K17 tail): [Ntok][C*4] bf16 (feature order c,ph,pw) -> (T,H,W,C) bf16, token perm optional
int k5_launch_unpatchify(const void* x, void* out, int T, int Hp, int Wp, int C, int ldx,
                         const int32_t* tok_perm, hipStream_t stream);
// K18: CFG combine + Euler.  v = cond (bf16) or bf16(u + bf16(w * bf16(c-u))); img += float(bf16(dt*v))
int k5_launch_cfg_euler(float* img, const void* v_cond, const void* v_uncond, float w, float dt, int64_t n,
                        hipStream_t stream, const float* dtvec = nullptr, const int* step = nullptr);
// fp32 -> bf16 cast, bf16 -> fp32
int k5_launch_cast_f32_bf16(const float* x, void* out, int64_t n, hipStream_t stream);
// weight packing: src [rows][cols] (K5_F32 / K5_BF16 / K5_F16, device) -> dst [rows][ld] bf16 (RNE) or fp32, pad columns zeroed
int k5_launch_pack_matrix(const void* src, int src_dtype, void* dst, int dst_bf16, int64_t rows, int cols, int ld, hipStream_t stream);

// causal_hw > 0: frame-causal scores, tiles wholly at columns >= (row / causal_hw + 1) * causal_hw are skipped (left unwritten)
int k5_launch_gemm_bf16_f32out(const void* A, const void* W, float* C, int M, int N, int K, int lda, int ldw, int ldc,
                               float alpha, int causal_hw, hipStream_t stream);

// ---- VAE decoder kernels (channels-last bf16 activations) ----
// 4-wave 256-row variant (conv3d_w4.hip); K5_ERR_UNSUPPORTED outside its range (Cin % 128, Cout = 128 or % 256, >= one round of tiles)
// quad_stats (nullable): [2 ceil(M / 256)][Cout / 4][2] fp32 partial GroupNorm sums of the stored outputs (see conv3d_w4.hip)
int k5_launch_conv3d_w4(const void* X, const void* W, const float* bias, void* out, int Ts, int Hs, int Ws, int Cin, int Cout,
                        int up_t, int up_s, int ldc, const void* resid, int ldr, float* quad_stats, hipStream_t stream);
int k5_launch_conv3d_bf16(const void* X, const void* W, const float* bias, void* out, int Ts, int Hs, int Ws, int Cin,
                          int Cout, int up_t, int up_s, int ldc, const void* resid, int ldr, hipStream_t stream);
int k5_launch_conv3d_bf16_strided(const void* X, const void* W, const float* bias, void* out, int Ts, int Hs, int Ws, int Cin,
                                  int Cout, int up_t, int up_s, int st_t, int st_s, int ldc, const void* resid, int ldr, hipStream_t stream);
// which kernel the LAST conv launcher call of this host thread took (diagnostics: the VAE engine counts them per handle so that a
// parity test can assert the production kernels ran, tests/test_gpu_vae.py)
enum { K5_CONV_KIND_TILE128 = 0, K5_CONV_KIND_W4 = 1, K5_CONV_KIND_W4_STATS = 2, K5_CONV_KIND_OUT3 = 3 };
int k5_conv3d_last_kind();
void k5_conv3d_set_last_kind(int kind);
size_t k5_groupnorm_workspace_bytes(int M, int G);
int k5_launch_groupnorm_bf16(const void* x, const float* gamma, const float* beta, void* out, int M, int C, int G, float eps,
                             int silu, int ldx, int ldo, void* workspace, hipStream_t s);
// GroupNorm whose statistics were emitted by the producing conv: quad_stats [nblk][C / 4][2] (k5_launch_conv3d_w4); stats_ws: 2 G floats
int k5_launch_groupnorm_bf16_quads(const void* x, const float* gamma, const float* beta, void* out, int M, int C, int G, float eps,
                                   int silu, int ldx, int ldo, const float* quad_stats, int nblk, float* stats_ws, hipStream_t s);
int k5_launch_causal_softmax(const float* scores, void* P, int S, int hw, int lds, int ldp, hipStream_t s);
// the mid block's attention in one kernel for C = 512 (vae_attn.hip): q, k [S][ldqk], vt [512][ldvt >= ceil(S/32)*32], o [S][ldo]
int k5_launch_vae_attention512(const void* q, const void* k, const void* vt, void* o, int S, int hw, int ldqk, int ldvt, int ldo,
                               float scale, hipStream_t stream);
int k5_launch_nchw_to_mc(const float* z, void* out, int C, int64_t M, int Cpad, hipStream_t s, int64_t cstride = 0);   // cstride: elements between the channels of z (0 = M)
int k5_launch_blend_place_bf16(const void* a, int64_t a_stride, int len_a, const void* b, int64_t b_stride, void* dst, int64_t dst_stride, int64_t outer,
                               int64_t inner, int extent, int keep, hipStream_t s);
int k5_launch_frames_to_uint8(const void* x, void* out, int64_t n, hipStream_t s);
int k5_launch_mc_to_nchw(const void* x, void* out, int C, int64_t M, int ldx, hipStream_t s);
int k5_launch_blend_bf16(const void* a, void* b, int64_t outer, int len_a, int len_b, int64_t inner, int extent, hipStream_t s);
