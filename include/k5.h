/* k5.h — C ABI of libk5.so, the MI355X (gfx950) denoising engine for Kandinsky-5 T2V Lite.
 *
 * The reference (ai-forever/Kandinsky-5) is pure Python/PyTorch and has NO FFI/plugin layer
 * (SURVEY.md §8b); its seams are Python call signatures.  Each entry point below therefore
 * cites the reference *Python interface* it stands behind (paths relative to the reference
 * repo).  The Python host mirror (kandinsky-5_amd/kandinsky/) binds these with ctypes and keeps
 * the reference's names/arguments/errors; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch types.  `stream` is a hipStream_t passed as void*
 *     (NULL = the null stream).  All device work is enqueued asynchronously on that stream.
 *   - device pointers are BORROWED from the caller (torch owns activations); the engine OWNS its
 *     packed weights and workspaces (hipMalloc).
 *   - every function returns a k5_status; k5_last_error() gives a thread-local message.
 *   - a handle is not thread-safe; one handle per GPU per process (one process per GPU).
 *   - bf16 tensors are row-major with explicit leading dimensions in ELEMENTS.
 */
#ifndef K5_H
#define K5_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  K5_OK = 0, K5_ERR_ARG = 1, K5_ERR_ALIGN = 2, K5_ERR_HIP = 3, K5_ERR_STATE = 4, K5_ERR_KEY = 5,
  K5_ERR_UNSUPPORTED = 6
} k5_status;
typedef enum { K5_F32 = 0, K5_BF16 = 1, K5_F16 = 2 } k5_dtype;
typedef enum { K5_EPI_BIAS = 0, K5_EPI_BIAS_M = 1, K5_EPI_GELU = 2, K5_EPI_GATE = 3, K5_EPI_F32 = 4 /* internal */ } k5_epilogue;

/* bumped whenever an entry point is added or changes meaning; the host binding checks it BEFORE binding symbols, so that a stale
 * libk5.so fails with a clear message instead of a missing-symbol lookup (round 3: 4) */
#define K5_ABI_VERSION 8
int k5_abi_version(void);
const char* k5_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Kernel-level entry points (used by the parity tests and by torch-side glue).
 * ---------------------------------------------------------------------------------------- */

/* C[M][N] = A[M][K] . W[N][K]^T (+bias) with fused epilogue; bf16 in/out, fp32 accumulate.
 * Replaces autocast nn.Linear: kandinsky/models/nn.py:180-191 (get_qkv), :204-206 (out_l),
 * :352-361 (FeedForward incl. nn.GELU), :71 (TextEmbeddings), :96 (VisualEmbeddings), :382 (OutLayer);
 * K5_EPI_GATE additionally fuses apply_gate_sum nn.py:30-33: C = bf16(resid + gate[n]*bf16(acc+bias)).
 * K5_EPI_BIAS_M adds bias[m] (used to emit V^T = W_v . X^T directly). bias/gate are fp32. */
int k5_gemm_bf16(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lda,
                 int ldw, int ldc, int epilogue, const void* resid, int ldr, const float* gate, void* stream);
/* The same GEMM on a NAMED kernel (tests and A/B tools; k5_gemm_bf16 picks by shape and cost): kernel 0 = as k5_gemm_bf16, 2 = 128 x 128 tiles,
 * 4 = the four-wave persistent kernel, 5 = 128 x 128 quadrants with deep prefetch, 8 = the eight-wave ping-pong kernel (a kernel whose shape
 * conditions do not hold falls through to the next one, as in the automatic choice); token_tile 0 = by cost, 128 / 192 / 256 = rows of the
 * four-wave kernel's workgroup tile (ABI 7).  Every kernel sums a K column in the same order: results are bit-identical across them. */
int k5_gemm_bf16_variant(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lda, int ldw, int ldc,
                         int epilogue, const void* resid, int ldr, const float* gate, void* stream, int kernel, int token_tile);

/* S_f32[M][N] = alpha * A[M][K] . W[N][K]^T — the fp32 attention scores of the VAE mid block (diffusers Attention called from
 * HunyuanVideoMidBlock3D, kandinsky/models/vae.py:341-362, with prepare_causal_attention_mask vae.py:110-122).  causal_hw > 0:
 * row i is only ever read at columns < (i / causal_hw + 1) * causal_hw (its own and earlier frames); output tiles wholly beyond
 * that limit are not computed and left unwritten.  C is fp32 [M][ldc]. */
int k5_gemm_bf16_f32out(const void* A, const void* W, float* C, int M, int N, int K, int lda, int ldw, int ldc,
                        float alpha, int causal_hw, void* stream);

/* P_bf16[i][j] = softmax_j(scores[i][j]) over the frame-causal columns j < (i / hw + 1) * hw, 0 elsewhere (j < ldp) — the masked
 * softmax of the same mid-block attention (mask: prepare_causal_attention_mask, vae.py:110-122).  scores fp32 [S][lds]. */
int k5_causal_softmax_bf16(const float* scores, void* P, int S, int hw, int lds, int ldp, void* stream);
/* The whole mid-block attention in one kernel for C = 512 (one head; the scores are never materialised):
 * O[i] = softmax_j<lim(i) (scale * q_i . k_j) V[j], lim(i) = (i / hw + 1) * hw — diffusers `Attention` + the frame-causal mask,
 * kandinsky/models/vae.py:110-122,341-362.  q, k: [S][ldqk] bf16 (512 columns each; k may be q + 512 of one [S][1024] buffer),
 * vt: V transposed [512][ldvt] bf16 with ldvt >= ceil(S / 32) * 32 (columns S .. ldvt must hold FINITE values — they are multiplied
 * by a probability of exactly 0; the engine zero-fills them), o: [S][ldo]. */
int k5_vae_attention512_bf16(const void* q, const void* k, const void* vt, void* o, int S, int hw, int ldqk, int ldvt, int ldo,
                             float scale, void* stream);

/* O[q][h*64+d] = softmax(Q K^T / 8) V per head (head_dim 64, non-causal, fp32 softmax).
 * Replaces FA(q,k,v): nn.py:201 (text self-attn), :254 (visual self-attn), :336 (cross-attn).
 * Q [q_len][ldq], K [kv_len][ldk] with head h at columns h*64..; Vt [H*64][ldvt] is V transposed
 * (row h*64+d, column key). */
int k5_attention_bf16(const void* Q, const void* K, const void* Vt, void* O, int H, int q_len, int kv_len,
                      int ldq, int ldk, int ldvt, int ldo, void* stream);

/* Split-key form of the dense attention (sequence-parallel overlap): process 64-key tiles e -> e + tile_off0
 * (+ tile_skip_n once the index reaches tile_skip_at) for e < tile_cnt (-1: to the last tile); flags & 1 resumes from
 * `state` (fp32 accumulators of an earlier call over other tiles), flags & 2 writes `state` instead of O.
 * state: k5_attention_state_size(H, q_len) bytes.  Two calls covering disjoint tile sets == one k5_attention_bf16 call. */
int64_t k5_attention_state_size(int H, int q_len);
int k5_attention_bf16_range(const void* Q, const void* K, const void* Vt, void* O, int H, int q_len, int kv_len,
                            int ldq, int ldk, int ldvt, int ldo, float score_bound, int tile_off0, int tile_cnt,
                            int tile_skip_at, int tile_skip_n, void* state, int flags, void* stream);
/* The same attention with the keys ALREADY multiplied by the softmax scale in the exp2 domain: Kc = bf16(log2(e)/8 * k)
 * (one rounding, done by the producer: the engine's rmsnorm/RoPE kernel).  The scores are then the exp2 arguments: no
 * per-score multiply-add.  Whole key tiles only (kv_len % 64 == 0: no ragged-tile masking; K5_ERR_ARG otherwise).
 * Softmax form: with score_bound > 0 and score_bound * log2(e)/8 <= 90 (score_bound bounds |q.k| in RAW units) the
 * constant offset 0 (exp2 can neither overflow nor flush a row); otherwise the lazy online max (the running offset rides
 * in the MFMA accumulator's initial value and is only revised when a tile's maximum leaves a 2^60 window). */
int k5_attention_bf16_prescaled(const void* Q, const void* Kc, const void* Vt, void* O, int H, int q_len, int kv_len,
                                int ldq, int ldk, int ldvt, int ldo, float score_bound, void* stream);
/* What the engine runs for FA(q,k,v) of the visual self-attention (nn.py:254): the form is chosen PER HEAD on the device.
 * k5_rmsnorm_rope_stats_bf16 (below) leaves max |q_h|^2 / max |k'_h|^2 over the rows; k5_attention_flags turns them into
 * head_flags[h] = 1 (|q|max |k'|max <= 90: fixed offset) or 0 (online max) and resets the statistics (kstat: nk partial
 * maxima at stride kstride floats — one per sequence-parallel rank).  k5_attention_bf16_prescaled_auto launches both forms
 * over the same grid; a workgroup exits at once unless its head is its form's.  head_flags NULL: variant 1 = online max
 * everywhere.  workspace: NULL or k5_attention_balance_size bytes (balanced launch). */
int k5_attention_flags(float* qstat, float* kstat, int nk, int kstride, int H, int force_online, int* head_flags, void* stream);
/* The same pair with PER-ROW softmax offsets (what the engine's single-GPU path runs): k5_attention_flags_rows also writes
 * kmax[h] = max |k'_h| and keeps heads up to |q|max |k'|max <= 190 on the fixed-offset form; there query row q of head h runs on
 * the constant offset max(0, |q| kmax[h] - 90) — exp2 cannot overflow whatever the data, and the result is exact unless the row's
 * whole sum underflows (< 2^-60: its largest score lies more than 150 below its Cauchy-Schwarz bound).  A workgroup that meets
 * such a row sets head_flags[h] = 0 and the online-max launch that follows in the same call redoes that head: head_flags is
 * read and written. */
int k5_attention_flags_rows(float* qstat, float* kstat, int nk, int kstride, int H, int force_online, int* head_flags, float* kmax,
                            void* stream);
int k5_attention_bf16_prescaled_rows(const void* Q, const void* Kc, const void* Vt, void* O, int H, int q_len, int kv_len, int ldq,
                                     int ldk, int ldvt, int ldo, int* head_flags, const float* kmax, void* workspace, void* stream);
/* CENTRED per-row offsets (round 3; what the engine's single-GPU path runs).  The plain offset |q| max|k'| - 90 is built from a bound the
 * scores may sit far below: keys of a head that share a large mean direction put EVERY score of a row near +-0.4 of that bound, and the
 * row sums underflow wholesale.  With a centre c_h (any convex combination of the head's keys; k5_rmsnorm_rope_centre_bf16 takes the mean
 * of a strided sample in the pass that writes the keys) and the radius R_h = max |k' - c_h|:  q.k' <= q.c + |q| R  and  max_j q.k'_j >= q.c,
 * so a row whose plain bound exceeds 90 runs with the offset q.c + |q| R - 90 — exp2 arguments <= 90 and a row sum >= 2^(90 - |q| R): no
 * underflow at all while |q| R <= 190.  Same softmax (any offset gives the same softmax); rows with a plain bound <= 90 keep offset 0.
 * k5_rmsnorm_rope_centre_bf16: as k5_rmsnorm_rope_stats_bf16, plus centre [H - scale_from_head][64] out and H - scale_from_head squared radii
 * appended to stats.  k5_attention_flags_rows_centred: rstat (consumed) -> krad; a head keeps the fixed form while min(|q|max kmax,
 * |q|max R) <= 190.  k5_attention_bf16_prescaled_rows_centred: k5_attention_bf16_prescaled_rows with the centred offsets. */
int k5_rmsnorm_rope_centre_bf16(void* x, const float* weight, const float* cos_tab, const float* sin_tab, int rows, int H, int ld,
                                int heads_per_weight, int rope_heads, float out_scale, int scale_from_head, float* stats, float* centre,
                                void* stream);
int k5_attention_flags_rows_centred(float* qstat, float* kstat, int nk, int kstride, int H, int force_online, int* head_flags, float* kmax,
                                    float* rstat, float* krad, void* stream);
int k5_attention_bf16_prescaled_rows_centred(const void* Q, const void* Kc, const void* Vt, void* O, int H, int q_len, int kv_len, int ldq,
                                             int ldk, int ldvt, int ldo, int* head_flags, const float* kmax, const float* centre,
                                             const float* krad, void* workspace, void* stream);
/* ANCHORED offsets for the heads BEYOND that window (ABI 5; the engine's one-GPU path, option "attn_anchor").  The Cauchy-Schwarz offsets
 * end where min(|q| kmax, |q| R) exceeds 190 and the online-max form (0.39-0.41 of peak instead of 0.52) used to take over.  Instead the offset
 * of a row is anchored at a score the row ACHIEVES: the maximum s over a sample of keys (the four 16-key tiles of the row's own 64-token
 * block and 28 tiles strided over the kv_len keys given), plus e = min(60, spread (s - sample mean)) — where the maximum over ALL
 * kv_total keys is expected when the scores scatter: spread = 1.4 (sqrt(2 ln kv_total) - c) / c, c = sqrt(2 ln 512): 0.44 at 47 616 keys — rounded
 * up, + 20.  Dense attention: the row's true maximum is >= s, so its term is >= 2^-80 and the row sum cannot underflow whatever the
 * norms; the form is exact while the true maximum lies below s + e + 132.  (Dense attention only: under NABLA the largest KEPT score
 * may lie far below a sample of all keys — the engine leaves such heads on the online form there.)  Beyond that the row sum exceeds 2^112 (or is
 * inf / NaN): the job falls back to the online form exactly like an underflowing one, and below 2^112 the sum bounds every accumulator,
 * so nothing overflows unnoticed.
 * k5_attention_flags_rows_anchored: as _centred, but a head beyond the window keeps flag 1 and gets kmax[h] = -1 (the marker) unless
 * prefer_online[h] (nullable) is set.  k5_attention_row_anchor: anchor [H][q_len] for the marked heads (other rows untouched); key0 = the
 * index among the given keys of query row 0's token, kv_total >= kv_len (a rank of a sharded schedule samples its own keys).  k5_attention_bf16_prescaled_rows_anchored: _rows_centred reading the anchors of the marked heads. */
int k5_attention_flags_rows_anchored(float* qstat, float* kstat, int nk, int kstride, int H, int force_online, int* head_flags, float* kmax,
                                     float* rstat, float* krad, const int* prefer_online, void* stream);
int k5_attention_row_anchor(const void* Q, const void* Kc, int H, int q_len, int kv_len, int ldq, int ldk, int key0, int kv_total,
                            const float* kmax, float* anchor, void* stream);
int k5_attention_bf16_prescaled_rows_anchored(const void* Q, const void* Kc, const void* Vt, void* O, int H, int q_len, int kv_len, int ldq,
                                              int ldk, int ldvt, int ldo, int* head_flags, const float* kmax, const float* centre,
                                              const float* krad, const float* anchor, void* workspace, void* stream);
/* One pass of a multi-pass schedule (sequence parallelism: local keys first, gathered keys later) with per-row offsets: key tiles
 * [tile_off0, tile_off0 + tile_cnt), flags & 1 = resume `state`, & 2 = leave it instead of writing O (k5_attention_state_size
 * bytes).  late_pass 1 = not the last pass, 2 = the last: a row that underflows in any pass writes head_flags[h] = 2, every later
 * fixed-offset launch and the online launches of the non-final passes then skip the head, and the online launch of the last pass
 * recomputes it from scratch over all kv_len keys. */
int k5_attention_bf16_prescaled_rows_pass(const void* Q, const void* Kc, const void* Vt, void* O, int H, int q_len, int kv_len, int ldq,
                                          int ldk, int ldvt, int ldo, int* head_flags, const float* kmax, int tile_off0, int tile_cnt,
                                          float* state, int flags, int late_pass, void* workspace, void* stream);
/* The same pass with the reference's norm_qk + apply_rotary of the QUERIES (nn.py:193-197, 239-243) done inside the kernel's Q load:
 * Q holds the raw query projection, q_norm_w the 64 RMSNorm weights, q_cos / q_sin [q_len][32] fp32 the rotary table
 * (k5_rope_table).  There is no max|q|^2 statistic then: call k5_attention_flags_rows with a zero query statistic (every head
 * starts on the fixed form) and the fixed-offset workgroups decide per head — a row whose bound |q| max|k'| exceeds 190 sets
 * head_flags[h] = 0 and the online launch of the same call owns the head.  head_flags and kmax both null: online max everywhere.
 * tile_cnt -1 = all key tiles from tile_off0 (single pass: state null, flags 0, late_pass 0). */
int k5_attention_bf16_prescaled_qnorm_pass(const void* Q, const void* Kc, const void* Vt, void* O, int H, int q_len, int kv_len, int ldq,
                                           int ldk, int ldvt, int ldo, const float* q_norm_w, const float* q_cos, const float* q_sin,
                                           int* head_flags, const float* kmax, int tile_off0, int tile_cnt, float* state, int flags,
                                           int late_pass, void* workspace, void* stream);
int k5_attention_bf16_prescaled_auto(const void* Q, const void* Kc, const void* Vt, void* O, int H, int q_len, int kv_len,
                                     int ldq, int ldk, int ldvt, int ldo, const int* head_flags, int variant, void* workspace,
                                     void* stream);
/* k5_attention_bf16[_bounded] with load balancing: the (head, 256-query) jobs that do not fill a whole round of the
 * device's resident workgroups are split 2-4 ways along the keys and merged (same result up to fp32 summation order).
 * workspace: k5_attention_balance_size(H, q_len) bytes.  The engine uses this for every large self-attention. */
int64_t k5_attention_balance_size(int H, int q_len);
int k5_attention_bf16_balanced(const void* Q, const void* K, const void* Vt, void* O, int H, int q_len, int kv_len,
                               int ldq, int ldk, int ldvt, int ldo, float score_bound, void* workspace, void* stream);

/* NABLA block-sparse attention = nablaT_v2 (kandinsky/models/utils.py:136-163, incl. the STA window of
 * fast_sta_nabla :108-133) + flex_attention(q,k,v,block_mask) (nn.py:257-280), tokens in fractal order
 * (utils.py:31-41): N = T*Hb*Wb*64.  k5_nabla_select_bf16 fills `workspace` (k5_nabla_workspace_size bytes) with the
 * block map of every (head, 64-query block); k5_attention_nabla_bf16 consumes it; k5_nabla_mask_u8 expands the map to
 * uint8 [H][N/64][N/64] (tests / diagnostics) and k5_nabla_counts copies the kept-block count per row. */
int64_t k5_nabla_workspace_size(int H, int num_blocks);
int k5_nabla_select_bf16(const void* q, const void* k, int ldq, int ldk, int H, int N, int T, int Hb, int Wb, int wT,
                         int wH, int wW, float P, void* workspace, void* stream);
int k5_attention_nabla_bf16(const void* Q, const void* K, const void* Vt, void* O, int H, int N, int ldq, int ldk,
                            int ldvt, int ldo, float score_bound, const void* workspace, void* stream);
int k5_nabla_mask_u8(const void* workspace, int H, int num_blocks, void* out_u8, void* stream);
/* Sequence-parallel form (SURVEY.md §8e "NABLA under SP"): this rank holds the query rows of global blocks
 * [q_block0, q_block0 + Nq/64) and, after the K / V^T all-gather, all N keys; V^T optionally in per-rank chunks
 * [chunk][H*64][vt_chunk_keys] (vt_chunk_keys = 0: plain [H*64][ldvt]).  Map rows are indexed by the LOCAL query block;
 * the workspace is sized by k5_nabla_workspace_size(H, N/64) as above.  Row i of the map equals row q_block0 + i of the
 * square map. mask: uint8 [H][Nq/64][N/64].
 * COUPLING (since the round-5 workspace layout): the key-tile lists sit BEHIND the logits of the Nq/64 selected rows, so their
 * offset depends on Nq.  The attention / mask / count calls that follow a select on a workspace must be given the SAME Nq (and H, N)
 * as that select — k5_nabla_select_bf16 counts as Nq = N; nothing in the workspace records it, a different Nq reads another region
 * without an error. */
int k5_nabla_select_rect_bf16(const void* q, const void* k, int ldq, int ldk, int H, int Nq, int q_block0, int N, int T,
                              int Hb, int Wb, int wT, int wH, int wW, float P, void* workspace, void* stream);
int k5_attention_nabla_rect_bf16(const void* Q, const void* K, const void* Vt, void* O, int H, int Nq, int N, int ldq,
                                 int ldk, int ldvt, int ldo, float score_bound, const void* workspace, int vt_chunk_keys,
                                 int64_t vt_chunk_stride, void* stream);
/* Sequence-parallel NABLA as the engine runs it.  k5_nabla_select_rect_local_bf16: k5_nabla_select_rect_bf16 with the key blocks
 * [local_block0, + local_blocks) — the rank's own, in place before the gather — leading every (head, 256-query group) list.
 * k5_attention_nabla_rect_prescaled_pass: the list-driven attention on keys pre-multiplied by log2(e)/8, per-head flags and per-row
 * offsets as in k5_attention_bf16_prescaled_rows; pass 0 = the whole lists in one launch group; 1 = only the leading local entries,
 * leaving the fp32 state (k5_attention_state_size bytes) and no output; 2 = the remaining entries, resuming the state, writing O.
 * A row that underflows on its per-row offset in pass 1 sets head_flags[h] = 2 and the online launch of pass 2 recomputes the head
 * over its whole lists.  Replaces nablaT_v2 + flex_attention (kandinsky/models/utils.py:136-163, nn.py:257-280) on a token shard. */
int k5_nabla_select_rect_local_bf16(const void* q, const void* k, int ldq, int ldk, int H, int Nq, int q_block0, int N, int T, int Hb,
                                    int Wb, int wT, int wH, int wW, float P, void* workspace, int local_block0, int local_blocks,
                                    void* stream);
int k5_attention_nabla_rect_prescaled_pass(const void* Q, const void* Kc, const void* Vt, void* O, int H, int Nq, int N, int ldq, int ldk,
                                           int ldvt, int ldo, const void* workspace, int vt_chunk_keys, int64_t vt_chunk_stride,
                                           int* head_flags, const float* kmax, int pass, float* state, void* stream);
int k5_nabla_mask_rect_u8(const void* workspace, int H, int q_blocks, int num_blocks, void* out_u8, void* stream);

/* Dense k5_attention_bf16 with a caller-proved bound |q.k| <= score_bound for every (query, key) pair
 * (after norm_qk nn.py:193-197 every head vector has |x| <= 8*max|weight|).  When bound*log2(e)/8 <= 90 the
 * softmax runs with the constant offset 0 instead of the online running max (same softmax, fewer VALU ops);
 * otherwise, or with score_bound <= 0, the online-max kernel runs. */
int k5_attention_bf16_bounded(const void* Q, const void* K, const void* Vt, void* O, int H, int q_len,
                              int kv_len, int ldq, int ldk, int ldvt, int ldo, float score_bound, void* stream);

/* apply_scale_shift_norm nn.py:25-28: out = bf16(LayerNorm(x, eps 1e-5, no affine)*(scale+1)+shift). */
int k5_ln_modulate_bf16(const void* x, const float* scale, const float* shift, void* out, int rows, int D,
                        int ldx, int ldo, void* stream);
/* norm_qk nn.py:193-197 + apply_rotary nn.py:35-40, in place on [rows][ld] holding H heads of 64:
 * RMSNorm(eps=2^-23, weight[(h/heads_per_weight)*64..]) -> bf16 -> adjacent-pair rotation by
 * cos/sin[rows][32] (heads < rope_heads; cos==NULL: no rotation) -> bf16. */
int k5_rmsnorm_rope_bf16(void* x, const float* weight, const float* cos_tab, const float* sin_tab, int rows,
                         int H, int ld, int heads_per_weight, int rope_heads, void* stream);
/* The engine's form of the same op: heads >= scale_from_head are multiplied by out_scale before their (single) bf16
 * rounding (keys for the pre-scaled attention), and stats[h] (device fp32 [H], zero before the first use) receives
 * max(stats[h], |x_row,h|^2) over the rows, of the bf16 values written. */
int k5_rmsnorm_rope_stats_bf16(void* x, const float* weight, const float* cos_tab, const float* sin_tab, int rows, int H,
                               int ld, int heads_per_weight, int rope_heads, float out_scale, int scale_from_head,
                               float* stats, void* stream);
/* apply_gate_sum nn.py:30-33 (standalone form). */
int k5_gate_sum_bf16(const void* x, const void* y, const float* gate, void* out, int rows, int D, void* stream);
/* fp32-island GEMV (Modulation nn.py:161-164, TimeEmbeddings nn.py:56-61): y = W.act(x) + b (+add). */
int k5_gemv_f32(const float* x, const float* W, const float* b, float* y, int N, int K, int silu_in,
                const float* add, void* stream);
/* TimeEmbeddings sinusoid nn.py:57-58; t = 1000*sigma by value. */
int k5_time_features_f32(float t, float* out, int D, void* stream);
/* TextEmbeddings.norm nn.py:67,72 (LayerNorm affine on bf16 rows, bf16 and/or bf16-rounded fp32 out). */
int k5_ln_affine_bf16(const void* x, const float* w, const float* b, void* out_bf16, float* out_f32, int rows,
                      int D, void* stream);
/* RoPE3D/RoPE1D nn.py:99-150 cos/sin tables [T*H*W][n0+n1+n2]; positions are device int32. */
int k5_rope_table_f32(float* cos_tab, float* sin_tab, const int32_t* pos_t, const int32_t* pos_h,
                      const int32_t* pos_w, int T, int H, int W, int n0, int n1, int n2, float s0, float s1,
                      float s2, const int32_t* tok_perm, void* stream);
/* VisualEmbeddings patchify nn.py:81-95 (patch (1,2,2)), fp32 (T,H,W,x_channels) -> bf16 [Ntok][Kpad];
 * channels in [x_channels, Cin_total) read as zero (generation_utils.py:107-112); tok_perm = fractal order. */
int k5_patchify_bf16(const float* x, void* out, int T, int H, int W, int x_channels, int Cin_total, int Kpad,
                     const int32_t* tok_perm, void* stream);
/* OutLayer un-patchify nn.py:384-399: [Ntok][4C] (c,ph,pw) -> (T,2Hp,2Wp,C) bf16. */
int k5_unpatchify_bf16(const void* x, void* out, int T, int Hp, int Wp, int C, int ldx,
                       const int32_t* tok_perm, void* stream);
/* CFG combine + Euler, generation_utils.py:74-76,128.  v_uncond NULL => no guidance. */
int k5_cfg_euler(float* img, const void* v_cond, const void* v_uncond, float w, float dt, int64_t n,
                 void* stream);

/* ------------------------------------------------------------------------------------------
 * Engine: DiffusionTransformer3D (kandinsky/models/dit.py:82-181) + sampler loop
 * (kandinsky/generation_utils.py:39-129).
 * ---------------------------------------------------------------------------------------- */
typedef struct k5_dit k5_dit;

/* ctor kwargs of DiffusionTransformer3D, dit.py:83-97 */
typedef struct k5_dit_config {
  int in_visual_dim, in_text_dim, in_text_dim2, time_dim, out_visual_dim;
  int patch_size[3];
  int model_dim, ff_dim, num_text_blocks, num_visual_blocks;
  int axes_dims[3];
  int visual_cond;
} k5_dit_config;

/* arguments of DiffusionTransformer3D.forward, dit.py:155-165 (+ sparse_params of
 * generation_utils.py:10-36) */
typedef struct k5_text_cond {
  const void* text_embed;   /* device, [text_len][in_text_dim], dtype text_dtype */
  const void* pooled_embed; /* device, [1][in_text_dim2], dtype text_dtype */
  int text_dtype;           /* k5_dtype */
  int text_len;
  const int32_t* text_rope_pos; /* HOST, [text_len] */
} k5_text_cond;

typedef struct k5_forward_args {
  const float* x;           /* device fp32 (T,H,W,x_channels) */
  int T, H, W, x_channels;  /* x_channels = visual_embed_dim, or in_visual_dim (cond channels implied zero) */
  k5_text_cond cond;
  float time;               /* 1000*sigma (generation_utils.py:57) */
  const int32_t* pos_t;     /* HOST visual_rope_pos[0..2] (generation_utils.py:173-177) */
  const int32_t* pos_h;
  const int32_t* pos_w;
  float scale_factor[3];    /* conf.metrics.scale_factor */
  int attention_type;       /* 0 = flash (dense), 1 = nabla */
  float nabla_P; int nabla_wT, nabla_wH, nabla_wW;
} k5_forward_args;

typedef struct k5_sample_args {
  k5_forward_args fwd;      /* fwd.x ignored; fwd.time ignored */
  k5_text_cond null_cond;   /* used when |guidance_weight-1| > 1e-6 */
  float* latent;            /* device fp32 (T,H,W,in_visual_dim), in: noise, out: final latent */
  int num_steps;
  const float* sigmas;      /* HOST [num_steps+1] (generation_utils.py:102-103) */
  float guidance_weight;
} k5_sample_args;

int k5_dit_create(const k5_dit_config* cfg, k5_dit** out);
void k5_dit_destroy(k5_dit* dit);
/* state_dict entry (checkpoint layout SURVEY.md App. D): host pointer, any of f32/bf16/f16. */
int k5_dit_load_tensor(k5_dit* dit, const char* key, const void* host_ptr, int dtype, const int64_t* shape,
                       int rank);
/* all tensors loaded -> pack (concat Wq|Wk, pad, cast) ; must precede forward */
int k5_dit_finalize(k5_dit* dit);
/* number of state_dict keys still missing (0 after a complete load); names via k5_last_error() */
int k5_dit_missing_keys(k5_dit* dit);
/* velocity (T,H,W,out_visual_dim) bf16, device */
int k5_dit_forward(k5_dit* dit, const k5_forward_args* args, void* out_velocity, void* stream);
/* whole Euler loop on device (generate, generation_utils.py:80-129) */
int k5_sample(k5_dit* dit, const k5_sample_args* args, void* stream);

/* Sequence parallelism (replaces the reference's DTensor head-parallel plan, kandinsky/models/parallelize.py:11-102,
 * keeps its launch contract LOCAL_RANK/WORLD_SIZE, kandinsky/utils.py:40-55): one process per GPU, rank r owns the
 * visual-token rows [r*N/world, (r+1)*N/world) (N a multiple of 64*world); K and V^T of every block are
 * all-gathered over RCCL/xGMI.  rccl_lib_path: library to dlopen (NULL/"" = default names).  Rank 0 obtains a
 * 128-byte ncclUniqueId with k5_comm_unique_id, the host broadcasts it, every rank calls k5_dit_comm_init. */
int k5_comm_unique_id(const char* rccl_lib_path, void* out_unique_id_128);
int k5_dit_comm_init(k5_dit* dit, const char* rccl_lib_path, int rank, int world, const void* unique_id_128);
/* Token counts that do not divide: whole 64-token blocks, ceil(blocks / world) per rank, the last rank takes the rest
 * (3660 blocks over 8 ranks = 7 x 458 + 454); K5_ERR_UNSUPPORTED if a rank would be left without a block.
 *
 * Loopback group (tests): `world` handles of ONE process on ONE GPU act as the ranks of a sequence-parallel run.  Every rank
 * must be driven by its own host thread (a collective is a rendezvous of the threads around device-to-device copies), and
 * every rank runs exactly the code path / offsets / launch sequence of a real multi-GPU run.  Not with k5_dit_set_graph. */
/* CFG-parallel (reference semantics: the two forwards of get_velocity, generation_utils.py:53-76): the handle runs ONE
 * branch of classifier-free guidance inside k5_sample — branch 0 the conditional forward, 1 the unconditional one — and exchanges the
 * bf16 velocity with the paired handle (the same token shard of the other rank group) once per step: a 2-rank communicator of its own,
 * initialised AFTER the sequence-parallel one (both are collective: same order on every rank).  Both handles of a pair then apply the
 * identical bf16 combine + fp32 Euler update, so all ranks of both groups hold bit-identical latents.  The exchange runs on a side
 * stream between two events and is part of the captured step under k5_dit_set_graph.  With guidance_weight == 1 the pair is idle.
 * unique_id_128: from k5_comm_unique_id on branch 0, carried to branch 1 by the host.  k5_dit_cfg_branch: 0 / 1, or -1 without a pair. */
int k5_dit_cfg_pair_init(k5_dit* dit, const char* rccl_lib_path, int branch, const void* unique_id_128);
int k5_dit_cfg_branch(k5_dit* dit);
typedef struct k5_loopback k5_loopback;
int k5_loopback_create(int world, k5_loopback** out);
void k5_loopback_destroy(k5_loopback* group);
int k5_dit_comm_init_loopback(k5_dit* dit, k5_loopback* group, int rank);
/* the CFG pair as a loopback group of world 2 (tests: 2 x P handles on one GPU = two sequence-parallel groups + P pairs) */
int k5_dit_cfg_pair_init_loopback(k5_dit* dit, k5_loopback* group, int branch);
/* IPC group (ABI 8): the second transport of the sharded path, the one SURVEY.md §8(e) names beside RCCL ("direct peer writes over xGMI into
 * IPC-mapped buffers"); same launch contract (one process per rank, kandinsky/utils.py:40-55; README.md:269-276), same schedules, same bits
 * as a loopback group of the same size.  Every rank exports its K / V^T slots, velocity and statistics buffers with hipIpcGetMemHandle on
 * first use, peers map them and READ the slices they need with a copy kernel; ordering is by epoch flags in device memory (system-scope
 * release stores / bounded polling kernels), not by host rendezvous.  Unlike RCCL it accepts several ranks on ONE device — which is how the
 * process boundary of the sharded path is exercised on a one-GPU box (bench.py --gpus P --oversubscribe; tests/test_gpu_ipc_ranks.py) —
 * and on an xGMI node it needs no library at all.  shm_name: name of a POSIX shared-memory control block, unique per group and per run
 * (the host makes one up on the group's rank 0 and broadcasts it over torch.distributed); collective over the `world` processes of the
 * group (<= 16).  Works under k5_dit_set_graph: the collectives' epochs live on the device, so the captured step replays.  Options (k5_dit_get_option): "ipc_ranks", "ipc_pair_ranks"
 * (0 = another transport), "ipc_collectives", "ipc_pulled_mb", "ipc_errors" (synchronises; first flag wait that hit K5_IPC_TIMEOUT_S). */
int k5_dit_comm_init_ipc(k5_dit* dit, const char* shm_name, int rank, int world);
int k5_dit_cfg_pair_init_ipc(k5_dit* dit, const char* shm_name, int branch);
/* Engine options by name (all default 0): "attn_mode" 0 = softmax form per head from the data, 1 = online max everywhere;
 * "sp_pass1_tiles" local key tiles attended before the K / V^T gather has landed (0 = all); "emulate_world" P = TIMING
 * ONLY: rank 0's share of a P-rank run on a world = 1 communicator — collectives move nothing, results are garbage and
 * k5_dit_get_option("emulated") reads 1 so that a bench can refuse the number. */
int k5_dit_set_option(k5_dit* dit, const char* name, int value);
int k5_dit_get_option(k5_dit* dit, const char* name, int* value);
/* Self-tuning sequence-parallel schedule (replaces the fixed plan of parallelize_dit, kandinsky/models/parallelize.py:11-102, and the
 * launcher's world-size bookkeeping, kandinsky/utils.py:40-55).  Which exchange is fastest — one in-place K / V^T all-gather per block,
 * the same exchange in 2 slices, the Ulysses all-to-all (heads % ranks == 0); for NABLA one or two passes over the lists — depends on
 * the node.  The first sharded forward of a handle with more than one rank therefore times one block's self-attention section under
 * every admissible candidate on its own shapes; the ranks exchange their times, each candidate costs its slowest rank, the cheapest is
 * kept for the life of the handle (every rank decides alike from the same table).  Options set explicitly through k5_dit_set_option
 * ("sp_mode", "sp_slices", "sp_nabla_passes") are left alone, and a tuning run only assigns the options it varied (dense: "sp_mode" /
 * "sp_slices"; NABLA: "sp_nabla_passes").  The tuner is OPT-IN ("sp_autotune" 1; K5_SP_AUTOTUNE=1): its choice comes from timings and the
 * schedules sum in different orders, so with it on the same seed may give different bits on two nodes; off (the default) a handle runs the
 * all-gather schedule.  2 makes the next sharded forward tune again.  k5_dit_sp_schedule: JSON text of what was measured and chosen, incl. the bare gather's
 * bytes and GB/s ("{}" before the first tuning run); returns the text length, copies at most len - 1 characters + NUL (buf may be null).
 * k5_sp_pick_schedule: the selection rule alone (host arithmetic, no GPU): times[rank * ncand + cand] in ms, <= 0 / non-finite = did not
 * run on that rank; valid (nullable) masks candidates; returns the chosen candidate or -1, max-over-ranks per candidate in cost_out. */
int k5_dit_sp_schedule(k5_dit* dit, char* buf, int len);
int k5_sp_pick_schedule(const float* times, int ncand, int world, const int* valid, float* cost_out);
/* (block, head) self-attention launches that took the fixed-offset / the online-max softmax since the last reset. */
int k5_dit_attn_variant_counts(k5_dit* dit, long long* fixed_heads, long long* online_heads, int reset);
/* kept / possible 64x64 blocks of the NABLA maps computed while profiling was on, since that reset (realised density). */
int k5_dit_nabla_block_counts(k5_dit* dit, long long* kept, long long* possible);
/* Diagnostics (ABI 7): while a tap is set, every NABLA map the handle computes on the one-GPU path — one per visual block and forward, in execution
 * order — is expanded to uint8 [H][N/64][N/64] (1 = kept; row = query block, fractal order) behind the ones already taken in the caller's device
 * buffer, as long as it fits; k5_dit_nabla_tap_count: maps computed since the tap was set (a count beyond capacity / map size = the buffer was
 * too small).  dev_u8 = NULL removes the tap.  Lets a test put the engine's map of block b at step s next to the reference's own
 * (nablaT_v2, kandinsky/models/utils.py:136-163, as nn.py:257-298 calls it inside every block): tests/test_gpu_fulldepth.py. */
int k5_dit_set_nabla_tap(k5_dit* dit, void* dev_u8, long long capacity_bytes);
int k5_dit_nabla_tap_count(k5_dit* dit, long long* maps);
/* ... and the blocks the list-driven attention executed for those maps (one-GPU path): the rows of a workgroup share ONE key-tile list, the
 * union of what they selected (replaces flex_attention's per-row block walk, nn.py:257-280), so a row also steps over tiles only its
 * neighbours wanted; kept / executed is the launch's union efficiency. */
int k5_dit_nabla_executed_blocks(k5_dit* dit, long long* executed);

/* ------------------------------------------------------------------------------------------
 * HunyuanVideo 3D-VAE decoder (kandinsky/models/vae.py): post_quant_conv + HunyuanVideoDecoder3D.forward
 * (vae.py:684-696, 870) on one latent tile; the tiling policy (vae.py:847-1204) stays on the host mirror.
 * ---------------------------------------------------------------------------------------- */
/* HunyuanVideoCausalConv3d (vae.py:125-163, k=3, replicate pad W(1,1) H(1,1) T(2,0)) with the nearest upsample of
 * HunyuanVideoUpsampleCausal3D (vae.py:187-205) optionally folded in (up_t, up_s in {1,2}); channels-last bf16:
 * X [Ts][Hs][Ws][Cin] (Cin % 64 == 0), W [Cout][27][Cin] (tap = (dt*3+dh)*3+dw), bias fp32, out [To*Ho*Wo][ldc];
 * resid (optional) [To*Ho*Wo][ldr]: out = bf16(bf16(conv+bias) + resid) (resnet skip, vae.py:274). */
int k5_conv3d_bf16(const void* X, const void* W, const float* bias, void* out, int Ts, int Hs, int Ws, int Cin, int Cout,
                   int up_t, int up_s, int ldc, const void* resid, int ldr, void* stream);
/* HunyuanVideoDownsampleCausal3D (vae.py:208-227; encoder): the same causal conv with an output stride st_t, st_s in {1, 2}
 * and padding 0: out [To*Ho*Wo][ldc], To = (Ts-1)/st_t + 1, Ho = (Hs-1)/st_s + 1, Wo = (Ws-1)/st_s + 1. */
int k5_conv3d_strided_bf16(const void* X, const void* W, const float* bias, void* out, int Ts, int Hs, int Ws, int Cin, int Cout,
                           int st_t, int st_s, int ldc, void* stream);
/* nn.GroupNorm(G, C, eps) (+SiLU) on channels-last bf16 rows [M][C] (vae.py:246-263,672-673): fp32 statistics,
 * bf16 out.  workspace: device scratch of k5_groupnorm_workspace_size(M, G) bytes. */
int64_t k5_groupnorm_workspace_size(int M, int G);
int k5_groupnorm_bf16(const void* x, const float* gamma, const float* beta, void* out, int M, int C, int G, float eps,
                      int silu, void* workspace, void* stream);

/* The production pairing of the two (vae.py:246-263: GroupNorm -> SiLU -> conv, the norm reading what the previous conv wrote): the
 * conv also emits the GroupNorm statistics of the outputs it STORES — per 128 rows and 4 consecutive channels (sum, sum of squares),
 * quad_stats: k5_conv3d_stats_size(M, Cout) bytes, M = To*Ho*Wo — and k5_groupnorm_bf16_quads normalises from them without a
 * statistics pass of its own (workspace as for k5_groupnorm_bf16).  Only the 4-wave implicit-GEMM kernel emits statistics:
 * K5_ERR_UNSUPPORTED (nothing launched) outside its range (Cin % 128, Cout = 128 or % 256, >= 5/8 of a round of 256-row tiles). */
int64_t k5_conv3d_stats_size(int M, int Cout);
int k5_conv3d_bf16_stats(const void* X, const void* W, const float* bias, void* out, int Ts, int Hs, int Ws, int Cin, int Cout,
                         int up_t, int up_s, int ldc, const void* resid, int ldr, float* quad_stats, void* stream);
int k5_groupnorm_bf16_quads(const void* x, const float* gamma, const float* beta, void* out, int M, int C, int G, float eps,
                            int silu, const float* quad_stats, void* workspace, void* stream);

typedef struct k5_vae k5_vae;
typedef struct k5_vae_config {   /* AutoencoderKLHunyuanVideo.__init__ kwargs, vae.py:709-731 */
  int latent_channels, out_channels;
  int block_out_channels[4];
  int layers_per_block, norm_num_groups;
} k5_vae_config;
int k5_vae_create(const k5_vae_config* cfg, k5_vae** out);
void k5_vae_destroy(k5_vae* vae);
/* checkpoint tensors by state_dict key (decoder.*, post_quant_conv.*; encoder.*, quant_conv.* enable k5_vae_encode_tile;
 * other keys are ignored), host or device ptr */
int k5_vae_load_tensor(k5_vae* vae, const char* key, const void* ptr, int dtype, const int64_t* shape, int rank);
int k5_vae_finalize(k5_vae* vae);
/* z: device fp32 (latent_channels,T,H,W) (already divided by scaling_factor, generation_utils.py:220) ->
 * out: device bf16 (out_channels, 4(T-1)+1, 8H, 8W) = self.decoder(self.post_quant_conv(z)) */
int k5_vae_decode_tile(k5_vae* vae, const float* z, int T, int H, int W, void* out, void* stream);
/* (ABI 8) the same with z read IN PLACE out of a longer latent: z_channel_stride = elements between the latent channels of z (0 = T*H*W), i.e. the
 * temporal slice z[:, :, t0 : t0 + T] of the tiling loop (vae.py:1144-1204 `_temporal_tiled_decode`) without a copy */
int k5_vae_decode_tile_strided(k5_vae* vae, const float* z, int64_t z_channel_stride, int T, int H, int W, void* out, void* stream);
/* Encoder half (SURVEY.md §8 f4 — image / video conditioning): quant_conv(HunyuanVideoEncoder3D.forward(x)) on ONE tile
 * (vae.py:574-586, 808-809); the tiling policy (tiled_encode :938-1010, _temporal_tiled_encode :1096-1142) stays on the
 * host mirror.  Needs the encoder.* / quant_conv.* tensors (k5_vae_has_encoder = 1; a decode-only load leaves them out).
 * x: device fp32 (3, T, H, W), T = 4k+1 frames, H, W multiples of 8 -> out: device bf16 (2*latent_channels, k+1, H/8, W/8)
 * = the moments [mean | logvar] the reference hands to DiagonalGaussianDistribution. */
int k5_vae_encode_tile(k5_vae* vae, const float* x, int T, int H, int W, void* out, void* stream);
int k5_vae_has_encoder(k5_vae* vae);
/* Diagnostics: launches per kernel route since the last reset — out8 = {conv on 128x128 tiles, conv 4-wave, conv 4-wave + GroupNorm
 * statistics, conv_out3, GroupNorm from the conv's statistics, GroupNorm with its own statistics pass, mid attention in one kernel
 * (C = 512), mid attention as GEMM-softmax-GEMM}.  Lets a parity test assert that the kernels the timed decode runs are the ones it
 * checked (tests/test_gpu_vae.py::test_production_tile_vs_reference_golden). */
int k5_vae_path_counts(k5_vae* vae, long long* out8, int reset);
/* blend_t / blend_v / blend_h (vae.py:908-936) on contiguous bf16 tensors viewed as [outer][len][inner]:
 * b[:, y, :] = a[:, len_a-extent+y, :]*(1-y/extent) + b[:, y, :]*(y/extent), y < extent (eager bf16 rounding) */
int k5_blend_bf16(const void* a, void* b, int64_t outer, int len_a, int len_b, int64_t inner, int extent, void* stream);
/* (ABI 8) blend_t + `tile[:, :, :keep]` + the `torch.cat` of `_temporal_tiled_decode` (vae.py:1185-1204) as ONE pass: views [outer][len][inner] with
 * explicit outer strides in elements (a decoded tile minus its first frame; the output video); dst[:, y] = y < extent ? the cross-fade of k5_blend_bf16
 * (a[:, len_a - extent + y], b[:, y]) : b[:, y], y < keep.  a = NULL: no cross-fade (the first tile).  inner a multiple of 8, 16-byte aligned pointers.
 * Neither a nor b is written: the previous tile's tail is read as the decoder left it, which is what the reference blends with while tiles are at
 * least 2 * extent long (the host mirror falls back to k5_blend_bf16 otherwise). */
int k5_blend_place_bf16(const void* a, int64_t a_stride, int len_a, const void* b, int64_t b_stride, void* dst, int64_t dst_stride, int64_t outer,
                        int64_t inner, int extent, int keep, void* stream);
/* (ABI 8) the pipeline's uint8 frames, ((x.clamp(-1, 1) + 1) * 127.5).to(torch.uint8) (reference generation_utils.py:222-224) in one pass with torch's
 * bf16 rounding after every elementwise op and the truncating conversion; n a multiple of 8 */
int k5_frames_to_uint8(const void* x_bf16, void* out_u8, int64_t n, void* stream);

/* fp8 (OCP e4m3) W8A8 GEMM, opt-in (BASELINE config 5 "fp8 MFMA weights"): C = epilogue(w_scale[n] * A8 . W8^T) with fp32
 * accumulation on v_mfma_scale_f32_16x16x128_f8f6f4.  A8 [M][lda] and W8 [N][ldw] are fp8 bytes (K a multiple of 128, >= 256,
 * rows 16-byte aligned).  epi: K5_EPI_BIAS -> C bf16 (no bias term: the feed-forward layers have none, nn.py:352-361);
 * K5_EPI_GELU -> C fp8 = e4m3(GELU(bf16(.))) (the input of the second feed-forward GEMM); K5_EPI_GATE -> gated residual as
 * k5_gemm_bf16.  k5_quant_rows_fp8: x bf16 [rows][K] -> e4m3 with a per-row scale max|x|/448 written to `scale`, or with the
 * static scale 1 when scale == NULL (activations).  This is a LOSSY mode (3 mantissa bits per operand); the engine uses it
 * only after k5_dit_set_fp8(dit, 1). */
int k5_gemm_fp8(const void* A8, const void* W8, const float* w_scale, void* C, int M, int N, int K, int lda, int ldw, int ldc,
                int epi, const void* resid, int ldr, const float* gate, void* stream);
int k5_quant_rows_fp8(const void* x_bf16, void* out_fp8, float* scale, int rows, int K, int ldx, int ldo, void* stream);

/* Run linear layers of the visual blocks in W8A8 e4m3 (weights quantised per output channel on first enable, activations with the
 * static scale 1).  `enabled` is a bit mask: 1 = the feed-forward GEMMs (nn.py:352-361), 2 = the q | k | V^T projections and 4 = the out
 * projection of the visual self-attention (nn.py:233-244, 282-284); 0 = off.  LOSSY and off by default: the distance to the bf16 path per
 * layer class is stated with the parity test tests/test_gpu_dit.py::test_fp8_feed_forward_mode and in DESIGN.md §4.2; from 256 tiles up the
 * GEMMs run on the four-wave e4m3 kernel (gemm_fp8.hip, round 4).  The sequence-parallel schedules keep the out projection in bf16 and the Ulysses
 * schedule also the q | k | V^T projections: k5_dit_get_option(dit, "fp8_effective") returns the mask that is in effect on the handle's path. */
int k5_dit_set_fp8(k5_dit* dit, int enabled);

/* k5_sample replays ONE hipGraph-captured sampler step (forwards + CFG/Euler, per-step scalars read from device tables at a
 * device-side step counter) instead of launching ~500 kernels per step from the host; results are bit-identical.  Ignored
 * while MagCache or profiling is on, and for fewer than 3 steps. */
int k5_dit_set_graph(k5_dit* dit, int enabled);

/* MagCache — replaces `set_magcache_params` + `magcache_forward` (reference kandinsky/magcache_utils.py:16-101).
 * `ratio_table` holds 2*num_steps float64 ratios (cond/uncond interleaved, already extended by the two leading 1.0 and
 * nearest-interpolated as :29-39 does — the host mirror kandinsky/magcache_utils.py does that); table_len == 0
 * disables.  Afterwards every k5_dit_forward / k5_sample forward runs the reference's decision: after the first
 * `retention_ratio` of the calls, skip the visual blocks and re-apply the cached bf16 residual of the call's slot
 * (call parity) while the accumulated |1 - prod ratio| < thresh for at most K consecutive calls.  no_cfg: the call
 * counter advances by 2 (slot 0 only), :91-94.  Defaults of the reference: thresh 0.12, K 2, retention_ratio 0.2. */
int k5_dit_set_magcache(k5_dit* dit, const double* ratio_table, int table_len, int no_cfg, double thresh, int K,
                        double retention_ratio);
/* Which calls of the reference's call sequence (cond, uncond, cond, ...) this handle executes: call indices
 * first_call, first_call + stride, ...  set_magcache installs (0, 1), or (0, 2) with no_cfg.  CFG-parallel rank groups
 *  run one branch each: conditional group (0, 2), unconditional group (1, 2). */
int k5_dit_magcache_calls(k5_dit* dit, int first_call, int stride);
/* introspection: current call counter and how many forwards ran / skipped the visual blocks since set_magcache */
int k5_dit_magcache_state(k5_dit* dit, int* cnt, long long* n_ran, long long* n_skipped);

/* per-kernel-family accumulated GPU time of the last forward(s), measured with hipEvents on the
 * engine's stream when profiling is enabled.  names: "attn_self","attn_cross","gemm","elementwise",...
 * level 0 = off, 1 = every family (an event pair per family switch: ~15 per block, the stream drains at each one),
 * 2 = only "attn_self", the roofline kernel (one pair per block) — what bench.py keeps on inside its timed region.
 * The pseudo-family "self_blocks" returns (0 ms, the number of visual blocks whose self-attention ran while profiling was on):
 * a block's attention is 1 timed scope on one GPU, 2 under sequence parallelism, 1 + S with a sliced exchange — FLOPs are per block. */
int k5_dit_set_profiling(k5_dit* dit, int level);
int k5_dit_get_profile(k5_dit* dit, const char* family, double* total_ms, int64_t* launches);
int k5_dit_reset_profile(k5_dit* dit);

#ifdef __cplusplus
}
#endif
#endif /* K5_H */
