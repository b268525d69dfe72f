// small_ops.hip — HBM-bound elementwise / normalisation kernels and the fp32 "island" ops of the
// Kandinsky-5 DiT for gfx950.  All bf16 traffic is 16 B per lane (8 x bf16), coalesced over the
// feature axis; row statistics are wave-level (64-lane) shuffles.  Reference call sites:
//   apply_scale_shift_norm  kandinsky/models/nn.py:25-28      -> ln_modulate_kernel     (K1)
//   apply_gate_sum          nn.py:30-33                        -> gate_sum_kernel        (K2)
//   norm_qk + apply_rotary  nn.py:193-197, 35-40               -> rmsnorm_rope_kernel    (K5+K3)
//   Modulation / TimeEmbeddings linears (fp32 islands) nn.py:56-61,161-164 -> gemv_f32_kernel
//   TimeEmbeddings sinusoid nn.py:57-58                        -> time_features_kernel
//   TextEmbeddings.norm     nn.py:67,72                        -> ln_affine_kernel       (K13)
//   VisualEmbeddings patchify nn.py:81-95 (+ fractal_flatten models/utils.py:31-41) -> patchify_kernel
//   OutLayer un-patchify    nn.py:384-399 (+ fractal_unflatten :44-51)             -> unpatchify_kernel
//   CFG combine + Euler     generation_utils.py:74-76,128      -> cfg_euler_kernel       (K18)
//   RoPE1D/RoPE3D tables    nn.py:99-150                        -> rope_table_kernel      (K15)
#include <stdlib.h>

#include "k5_common.h"
#include "k5_kernels.h"

namespace {

constexpr int MAXC = 4;  // 16-B chunks per lane per row: D <= 64*8*4 = 2048

// ---------------------------------------------------------------------------------------------
// K1: one wave per row.  Round 6: the per-column vectors (scale + 1 | shift, or weight | bias) are staged ONCE per workgroup in LDS — until
// round 5 every row re-read both vectors from L1 (14 KB of fp32 per 3.5-KB row at D = 1792: two thirds of the bytes the texture path moved).
// 76 -> 64 us per call at 47 616 x 1792 (4.5 -> 5.4 TB/s, profiles/r06_ln_rows_ab.log).  A wave may walk several rows (`4 gridDim` apart;
// K5_LN_WG caps the grid) but one row per wave measured best: 5.36 TB/s against 4.7-5.2 with 512-4096 workgroups.  Same operations in the same
// order per row: the bits do not change.  (All four 16-B loads of a row are issued in one burst ahead of the arithmetic: the same kernel with the
// loads inside the chunk loop ran at 4.1 TB/s, profiles/r06_ln_rows_ab.log second table.)
template <bool AFFINE>
__global__ __launch_bounds__(256) void ln_kernel(const bf16_t* __restrict__ x, const float* __restrict__ a,
                                                 const float* __restrict__ b, bf16_t* __restrict__ out,
                                                 float* __restrict__ out_f32, int rows, int D, int ldx, int ldo,
                                                 uint8_t* __restrict__ out8 = nullptr) {   // e4m3(bf16(.)), static scale 1, row stride D (fp8 modes of the engine)
  // [a | b][half 0 | half 1][chunk][4]: a lane's two 16-B halves of a chunk sit in two planes, so that consecutive lanes read consecutive
  // 16 B (conflict-free ds_read_b128)
  __shared__ __attribute__((aligned(16))) float vec[2][2][64 * MAXC][4];
  const int lane = threadIdx.x & 63;
  const int nch = D >> 3;
  for (int c = threadIdx.x; c < 2 * nch; c += 256) {       // c = 2 chunk + half
    const f32x4 av = *reinterpret_cast<const f32x4*>(a + 4 * c), bv = *reinterpret_cast<const f32x4*>(b + 4 * c);
    *reinterpret_cast<f32x4*>(vec[0][c & 1][c >> 1]) = AFFINE ? av : f32x4{av[0] + 1.0f, av[1] + 1.0f, av[2] + 1.0f, av[3] + 1.0f};   // K1: scale + 1
    *reinterpret_cast<f32x4*>(vec[1][c & 1][c >> 1]) = bv;
  }
  __syncthreads();
  const int stride = 4 * gridDim.x;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  u32x4 raw[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
    if (lane + 64 * i < nch) raw[i] = *reinterpret_cast<const u32x4*>(x + (size_t)row * ldx + 8 * (lane + 64 * i));
  for (; row < rows; row += stride) {
    float v[MAXC][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      if (lane + 64 * i < nch) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[i][2 * j] = __uint_as_float(raw[i][j] << 16);
          v[i][2 * j + 1] = __uint_as_float(raw[i][j] & 0xffff0000u);
          sum += v[i][2 * j] + v[i][2 * j + 1];
        }
      }
    }
    const int nrow = row + stride;
    if (nrow < rows) {   // (a capped grid) the next row of this wave: in flight under this row's arithmetic and stores
#pragma unroll
      for (int i = 0; i < MAXC; ++i)
        if (lane + 64 * i < nch) raw[i] = *reinterpret_cast<const u32x4*>(x + (size_t)nrow * ldx + 8 * (lane + 64 * i));
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      if (lane + 64 * i < nch) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; sq += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + 1e-5f);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
        float o[8];
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(vec[0][0][ch]), a1 = *reinterpret_cast<const f32x4*>(vec[0][1][ch]);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(vec[1][0][ch]), b1 = *reinterpret_cast<const f32x4*>(vec[1][1][ch]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float n = (v[i][j] - mean) * rstd;
          const float aj = j < 4 ? a0[j] : a1[j - 4], bj = j < 4 ? b0[j] : b1[j - 4];
          // K1: n * (scale + 1) + shift ; K13: n * weight + bias
          o[j] = __fadd_rn(__fmul_rn(n, aj), bj);
        }
        if (out) {
          u32x4 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
          *reinterpret_cast<u32x4*>(out + (size_t)row * ldo + 8 * ch) = pk;
        }
        if (out_f32) {
#pragma unroll
          for (int j = 0; j < 8; ++j) out_f32[(size_t)row * D + 8 * ch + j] = bf_round(o[j]);
        }
        if (out8) {   // what k5_launch_quant_rows_fp8 (static scale) makes of the bf16 row: the bf16 rounding first, then the saturating e4m3 one
          uint2 q8;
          q8.x = pack_fp8x4(bf_round(o[0]), bf_round(o[1]), bf_round(o[2]), bf_round(o[3]));
          q8.y = pack_fp8x4(bf_round(o[4]), bf_round(o[5]), bf_round(o[6]), bf_round(o[7]));
          *reinterpret_cast<uint2*>(out8 + (size_t)row * D + 8 * ch) = q8;
        }
      }
    }
  }
}

// workgroups of a launch: one wave per row unless K5_LN_WG caps the grid (A/B: a wave then walks several rows)
inline int ln_grid(int rows) {
  static const int cap = getenv("K5_LN_WG") ? atoi(getenv("K5_LN_WG")) : 0x7fffffff;
  const int wg = (rows + 3) / 4;
  return wg < cap ? wg : cap;
}

// ---------------------------------------------------------------------------------------------
// K5 + K3: one thread per 16-B chunk (8 of the 64 head elements); 8-lane groups own one head.  Grid-stride over rows with a
// stride that is a multiple of H*8 threads, so a thread keeps its (head, chunk) and can carry the head's running
// max |x|^2 in a register: one atomic per 8-lane group per launch instead of one per head vector.
// MEANS (round 4, NABLA): the pass also leaves the 64-token BLOCK MEANS of the heads' unscaled, normalised + rotated values — what
// block_mean_kernel (nabla.hip) computed by reading q | k once more: fp32 sum of the bf16-rounded values in row order, x 1/64, one bf16 rounding
// (the reference's bf16 `.mean(-2)`, utils.py:140-143).  A thread then walks whole blocks of 64 CONSECUTIVE rows (block bb = slot, slot + S, ...)
// instead of rows `stride` apart, so the sums stay in its registers; heads < scale_from_head go to mean_q [head][mq_stride blocks][64], the others
// to mean_k [head - scale_from_head][mk_stride][64].  With the means taken here the unscaled keys need not be stored: scaled in place.
template <bool MEANS>
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(bf16_t* __restrict__ x, const float* __restrict__ weight,
                                                           const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                           int rows, int H, int heads_per_weight, int ld, int rope_heads,
                                                           float out_scale, int scale_from_head, bf16_t* __restrict__ scaled_out,
                                                           int ld_scaled, float* __restrict__ stats, const float* __restrict__ centre,
                                                           bf16_t* __restrict__ mean_q, int mq_stride, bf16_t* __restrict__ mean_k, int mk_stride) {
  // stats: per-head maxima of this block: |x_h|^2 for the H heads of a row, then (centre given) |k'_h - c_h|^2 for the scaled heads —
  // the radius of a head's keys around the centre key_centre_kernel estimated; H + (H - scale_from_head) <= 256 entries
  __shared__ unsigned int smax[256];
  if (stats) { smax[threadIdx.x] = 0u; __syncthreads(); }
  const int64_t total = (int64_t)rows * H * 8;
  const int64_t stride = (int64_t)gridDim.x * 256;        // multiple of H * 8 (launcher)
  const int64_t g0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)(g0 & 7);
  const int head = (int)((g0 >> 3) % H);
  const float* w = weight + (head / heads_per_weight) * 64 + 8 * c;
  float wv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) wv[j] = w[j];
  const bool rope = cosT && head < rope_heads;
  const bool scaled = head >= scale_from_head;
  float n2max = 0.f, r2max = 0.f;
  float msum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float cv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool centred = stats && centre && scaled;
  if (centred) {
#pragma unroll
    for (int j = 0; j < 8; ++j) cv[j] = centre[(head - scale_from_head) * 64 + 8 * c + j];
  }
  // every lane of a wave runs the same number of iterations except in the last one (shuffles need the whole 8-lane group:
  // groups never straddle the end because total is a multiple of 8)
  const int slot0 = (int)(g0 / (H * 8)), nslot = (int)(stride / (H * 8)), nblk = rows >> 6;
  // MEANS: iteration i = (block, t): block = slot0 + (i / 64) * nslot, row = 64 block + t; otherwise row i of this thread's stride
  // MEANS: the NEXT row's 16 bytes are requested before this row is worked on and stored (the compiler cannot move a load of x above a store
  // to x by itself): two loads in flight per thread — with whole 64-row blocks per thread the launch is a little over one round of resident
  // workgroups, and one load per thread did not keep HBM busy through its second round
  // (Round 6: the same prefetch in the dense form — rows `stride` apart, ~20 per thread at 47 616 tokens — measured SLOWER: 154 us against 142-144,
  // profiles/r06_elem_ab.log; seven waves per SIMD with one load each already cover the latency, the second load only adds to the queue.)
  u32x4 raw_pf = {0u, 0u, 0u, 0u};
  if (MEANS && slot0 < nblk) raw_pf = *reinterpret_cast<const u32x4*>(x + (size_t)(64 * slot0) * ld + head * 64 + 8 * c);
  for (int64_t g = g0, it = 0; MEANS ? (slot0 + (int)(it >> 6) * nslot < nblk) : (g < total); g += stride, ++it) {
    const int mblk = MEANS ? slot0 + (int)(it >> 6) * nslot : 0, mt = (int)(it & 63);
    const int row = MEANS ? 64 * mblk + mt : (int)((g >> 3) / H);
    bf16_t* px = x + (size_t)row * ld + head * 64 + 8 * c;
    u32x4 raw;
    if (MEANS) {
      raw = raw_pf;
      const int nblk_next = slot0 + (int)((it + 1) >> 6) * nslot;
      if (nblk_next < nblk) raw_pf = *reinterpret_cast<const u32x4*>(x + (size_t)(64 * nblk_next + (int)((it + 1) & 63)) * ld + head * 64 + 8 * c);
    } else {
      raw = *reinterpret_cast<const u32x4*>(px);
    }
    float v[8];
    float sq = 0.f;
    // The sum of squares with its operations SPELLED OUT (round 6) — what the compiler made of `sq += lo * lo + hi * hi` in both instantiations of this
    // kernel (a pair is fma(hi, hi, lo * lo); the pairs added in order; the eight chunks of a head as a tree; fma(sq, 1/64, eps)) — because the
    // attention kernel's fused query norm (attn_fwd.hip, K5QueryNorm) restates it and contraction had picked fma(lo, lo, hi * hi) there: 1 (row, head) in
    // 10^4 then got another last bit of 1 / rms and, now and then, another bf16 rounding of a query element.  Same bits as before here.
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[2 * j] = __uint_as_float(raw[j] << 16);
      v[2 * j + 1] = __uint_as_float(raw[j] & 0xffff0000u);
      sq = __fadd_rn(sq, fmaf(v[2 * j + 1], v[2 * j + 1], __fmul_rn(v[2 * j], v[2 * j])));
    }
    sq = __fadd_rn(sq, __shfl_xor(sq, 1, 64));
    sq = __fadd_rn(sq, __shfl_xor(sq, 2, 64));
    sq = __fadd_rn(sq, __shfl_xor(sq, 4, 64));
    const float rs = rsqrtf(fmaf(sq, 1.0f / 64.0f, 1.1920928955078125e-07f));  // eps = finfo(fp32).eps
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = bf_round(__fmul_rn(__fmul_rn(v[j], rs), wv[j]));  // .type_as(q)
    if (rope) {
      const f32x4 cs = *reinterpret_cast<const f32x4*>(cosT + (size_t)row * 32 + 4 * c);
      const f32x4 sn = *reinterpret_cast<const f32x4*>(sinT + (size_t)row * 32 + 4 * c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // __fmul_rn / __fadd_rn are plain * and + to this compiler (no OCML_BASIC_ROUNDED_OPERATIONS): under -ffp-contract=fast ONE of the two products
        // of a rotation is fused into the add, and which one is the compiler's choice per kernel.  Spelled out (round 6) as what both instantiations of
        // this kernel had: the product with the ODD element rounded, the one with the even element fused — the form the attention kernel's fused query
        // norm restates (it had the other one: 7 (row, head) pairs in 10^4 got another bf16 rounding of one query element).  Same bits as before here.
        const float x0 = y[2 * j], x1 = y[2 * j + 1];
        y[2 * j] = fmaf(cs[j], x0, -__fmul_rn(sn[j], x1));
        y[2 * j + 1] = fmaf(sn[j], x0, __fmul_rn(cs[j], x1));
      }
    }
    if (MEANS) {
      if (mt == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) msum[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) msum[j] += bf_round(y[j]);       // the bf16 values the unscaled tensor holds (held), in row order
      if (mt == 63) {
        const u32x4 mk = {pack_bf16x2(msum[0] * (1.f / 64), msum[1] * (1.f / 64)), pack_bf16x2(msum[2] * (1.f / 64), msum[3] * (1.f / 64)),
                          pack_bf16x2(msum[4] * (1.f / 64), msum[5] * (1.f / 64)), pack_bf16x2(msum[6] * (1.f / 64), msum[7] * (1.f / 64))};
        bf16_t* mo = head < scale_from_head ? (mean_q ? mean_q + ((size_t)head * mq_stride + mblk) * 64 + 8 * c : nullptr)
                                            : (mean_k ? mean_k + ((size_t)(head - scale_from_head) * mk_stride + mblk) * 64 + 8 * c : nullptr);
        if (mo) *reinterpret_cast<u32x4*>(mo) = mk;
      }
    }
    u32x4 pks = {0u, 0u, 0u, 0u};
    if (scaled) {   // keys handed to the pre-scaled softmax: k' = bf16(log2(e)/8 * k), one rounding
      if (scaled_out) {   // NABLA: the unscaled keys stay in place (block map), the scaled copy goes to its own buffer
        pks = u32x4{pack_bf16x2(__fmul_rn(y[0], out_scale), __fmul_rn(y[1], out_scale)), pack_bf16x2(__fmul_rn(y[2], out_scale), __fmul_rn(y[3], out_scale)),
                    pack_bf16x2(__fmul_rn(y[4], out_scale), __fmul_rn(y[5], out_scale)), pack_bf16x2(__fmul_rn(y[6], out_scale), __fmul_rn(y[7], out_scale))};
        *reinterpret_cast<u32x4*>(scaled_out + (size_t)row * ld_scaled + (head - scale_from_head) * 64 + 8 * c) = pks;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = __fmul_rn(y[j], out_scale);
      }
    }
    const u32x4 pk = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7])};
    *reinterpret_cast<u32x4*>(px) = pk;
    if (stats) {   // |x|^2 of the bf16 values the softmax will consume (the scaled copy where there is one)
      const u32x4 src = (scaled && scaled_out) ? pks : pk;
      float n2 = 0.f, r2 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float lo = __uint_as_float(src[j] << 16), hi = __uint_as_float(src[j] & 0xffff0000u);
        n2 += lo * lo + hi * hi;
        if (centred) { const float dl = lo - cv[2 * j], dh = hi - cv[2 * j + 1]; r2 += dl * dl + dh * dh; }
      }
      n2 += __shfl_xor(n2, 1, 64);   // the head vector's 8 lanes
      n2 += __shfl_xor(n2, 2, 64);
      n2 += __shfl_xor(n2, 4, 64);
      n2max = n2 == n2 ? fmaxf(n2max, n2) : __uint_as_float(0x7f800000u);   // NaN -> +inf: forces the online-max path
      if (centred) {
        r2 += __shfl_xor(r2, 1, 64);
        r2 += __shfl_xor(r2, 2, 64);
        r2 += __shfl_xor(r2, 4, 64);
        r2max = r2 == r2 ? fmaxf(r2max, r2) : __uint_as_float(0x7f800000u);
      }
    }
  }
  // per-head max over the rows of this call: |q.k'| <= |q| |k'| then bounds every exp2 argument of the head
  // (attn_flags_kernel).  Non-negative floats order like their bit patterns.  No global atomics (a hundred thousand of them on
  // 28 addresses cost more than the kernel): LDS max per block, one partial row per block, reduced by stats_reduce_kernel.
  if (stats) {
    const int Hs = centre ? H + (H - scale_from_head) : H;
    if (c == 0 && (MEANS ? slot0 < nblk : g0 < total)) {
      atomicMax(&smax[head], __float_as_uint(n2max));
      if (centred) atomicMax(&smax[H + head - scale_from_head], __float_as_uint(r2max));
    }
    __syncthreads();
    if ((int)threadIdx.x < Hs) stats[(size_t)blockIdx.x * Hs + threadIdx.x] = __uint_as_float(smax[threadIdx.x]);
  }
}

// Centre of a head's keys for the centred Cauchy-Schwarz bound of the attention's per-row softmax offsets (attn_fwd.hip, AttnP::kcentre):
// c_h = mean of k'_j over a strided SAMPLE of the rows — any vector is a valid centre for the upper bound q.k' <= q.c + |q| max|k' - c|, and
// a convex combination of actual keys also gives the lower bound max_j q.k'_j >= q.c that rules the row-sum underflow out.  Runs BEFORE
// rmsnorm_rope_kernel on the raw projection, with that kernel's arithmetic (RMSNorm, weight, bf16, RoPE, scale, bf16), so that the main
// kernel can take max|k' - c|^2 in its one pass.  One block per key head: thread = (sample lane t >> 3, 16-B chunk t & 7); fixed-order sums.
// Sample size: ANY convex combination of keys is a valid centre; what the sample has to catch is a component the head's keys share, and
// 256 rows spread over the sequence do.  The kernel is one workgroup per head with a serial chain of row loads per thread — 1024 rows
// measured 56 us per block (+1.8 ms per step, same-box A/B against attn_row_offsets = 0: 544.3 vs 542.0 ms), 8 iterations instead of 32.
constexpr int K5_CENTRE_SAMPLE = 256;
__global__ __launch_bounds__(256) void key_centre_kernel(const bf16_t* __restrict__ x, const float* __restrict__ weight,
                                                         const float* __restrict__ cosT, const float* __restrict__ sinT, int rows, int ld,
                                                         int head0, int heads_per_weight, int rope_heads, float out_scale, int nsample,
                                                         float* __restrict__ centre) {
  __shared__ float part[32][64];
  const int hk = blockIdx.x, head = head0 + hk, c = threadIdx.x & 7, sl = threadIdx.x >> 3;
  const float* w = weight + (head / heads_per_weight) * 64 + 8 * c;
  float wv[8], acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { wv[j] = w[j]; acc[j] = 0.f; }
  const bool rope = cosT && head < rope_heads;
  // every row this thread samples is requested BEFORE any of them is worked on (round 5: the serial chain of 8 dependent row loads was 10.9 us per
  // block at 3328 tokens — a launch that costs one memory round trip instead of eight); same values summed in the same order
  constexpr int NIT = K5_CENTRE_SAMPLE / 32;
  u32x4 raws[NIT]; f32x4 csv[NIT], snv[NIT]; int rowv[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = it * 32 + sl;
    const bool live = i < nsample;
    rowv[it] = (int)(((long long)(live ? i : 0) * rows) / nsample);
    raws[it] = *reinterpret_cast<const u32x4*>(x + (size_t)rowv[it] * ld + head * 64 + 8 * c);
    if (rope) {
      csv[it] = *reinterpret_cast<const f32x4*>(cosT + (size_t)rowv[it] * 32 + 4 * c);
      snv[it] = *reinterpret_cast<const f32x4*>(sinT + (size_t)rowv[it] * 32 + 4 * c);
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = it * 32 + sl;
    const bool live = i < nsample;
    const u32x4 raw = raws[it];
    float v[8], sq = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[2 * j] = __uint_as_float(raw[j] << 16);
      v[2 * j + 1] = __uint_as_float(raw[j] & 0xffff0000u);
      sq += v[2 * j] * v[2 * j] + v[2 * j + 1] * v[2 * j + 1];
    }
    sq += __shfl_xor(sq, 1, 64);
    sq += __shfl_xor(sq, 2, 64);
    sq += __shfl_xor(sq, 4, 64);
    const float rs = rsqrtf(sq * (1.0f / 64.0f) + 1.1920928955078125e-07f);
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = bf_round(__fmul_rn(__fmul_rn(v[j], rs), wv[j]));
    if (rope) {
      const f32x4 cs = csv[it], sn = snv[it];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float x0 = y[2 * j], x1 = y[2 * j + 1];
        y[2 * j] = __fadd_rn(__fmul_rn(cs[j], x0), __fmul_rn(-sn[j], x1));
        y[2 * j + 1] = __fadd_rn(__fmul_rn(sn[j], x0), __fmul_rn(cs[j], x1));
      }
    }
    if (live)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += bf_round(__fmul_rn(y[j], out_scale));
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[sl][8 * c + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
    for (int l = 0; l < 32; ++l) s += part[l][threadIdx.x];
    centre[hk * 64 + threadIdx.x] = s / (float)nsample;
  }
}

// out[h] = max(out[h], max over the nblk partial rows): a few blocks, each over a slice of the rows (thread t: head t % H, row
// lane t / H), then one atomic per head and block — a handful, not one per head vector
__global__ __launch_bounds__(256) void stats_reduce_kernel(const float* __restrict__ part, int nblk, int H, float* __restrict__ out) {
  __shared__ float red[256];
  const int per = 256 / H;                  // row lanes per head
  const int h = threadIdx.x % H, lane = threadIdx.x / H;
  const int b0 = (int)((long long)nblk * blockIdx.x / gridDim.x), b1 = (int)((long long)nblk * (blockIdx.x + 1) / gridDim.x);
  float m = 0.f;
  if (lane < per)
    for (int b = b0 + lane; b < b1; b += per) { const float v = part[(size_t)b * H + h]; m = v == v ? fmaxf(m, v) : __uint_as_float(0x7f800000u); }
  red[threadIdx.x] = lane < per ? m : 0.f;
  __syncthreads();
  if ((int)threadIdx.x < H) {
    float r = 0.f;
    for (int l = 0; l < per; ++l) r = fmaxf(r, red[l * H + threadIdx.x]);   // NaNs were mapped to +inf above
    atomicMax(reinterpret_cast<unsigned int*>(out) + threadIdx.x, __float_as_uint(r));
  }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_sum_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ y,
                                                       const float* __restrict__ gate, bf16_t* __restrict__ out,
                                                       int64_t nchunks, int chunks_per_row) {
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < nchunks; g += (int64_t)gridDim.x * 256) {
    const int ch = (int)(g % chunks_per_row);
    const u32x4 rx = *reinterpret_cast<const u32x4*>(x + g * 8);
    const u32x4 ry = *reinterpret_cast<const u32x4*>(y + g * 8);
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gate + 8 * ch), g1 = *reinterpret_cast<const f32x4*>(gate + 8 * ch + 4);
    float o[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gl = j < 2 ? g0[2 * j] : g1[2 * j - 4], gh = j < 2 ? g0[2 * j + 1] : g1[2 * j - 3];
      o[2 * j] = __fadd_rn(__uint_as_float(rx[j] << 16), __fmul_rn(gl, __uint_as_float(ry[j] << 16)));
      o[2 * j + 1] = __fadd_rn(__uint_as_float(rx[j] & 0xffff0000u), __fmul_rn(gh, __uint_as_float(ry[j] & 0xffff0000u)));
    }
    u32x4 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
    *reinterpret_cast<u32x4*>(out + g * 8) = pk;
  }
}

// ---------------------------------------------------------------------------------------------
// fp32 GEMV, one wave per output row (weights streamed once, 16 B per lane)
__global__ __launch_bounds__(256) void gemv_f32_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                       const float* __restrict__ b, float* __restrict__ y, int N, int K,
                                                       int silu_in, const float* __restrict__ add) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float* w = W + (size_t)n * K;
  float acc = 0.f;
  for (int k = 4 * lane; k < K; k += 256) {
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
    f32x4 xv = *reinterpret_cast<const f32x4*>(x + k);
    if (silu_in) {
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[j] = xv[j] / (1.0f + expf(-xv[j]));
    }
    acc += wv[0] * xv[0] + wv[1] * xv[1] + wv[2] * xv[2] + wv[3] * xv[3];
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    float r = acc + (b ? b[n] : 0.f);
    if (add) r += add[n];
    y[n] = r;
  }
}

// tvec / step: hipGraph-replayable form — the time is read from a device table at the device-side step counter
__global__ void time_features_kernel(float t, const float* __restrict__ tvec, const int* __restrict__ step, float* __restrict__ out, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = D / 2;
  if (i >= half) return;
  if (tvec) t = tvec[*step];
  // get_freqs(models/utils.py:21-28): exp(-ln(1e4) * i / half) in fp32
  const float f = expf(__fdiv_rn(__fmul_rn(-9.210340371976184f, (float)i), (float)half));
  const float a = __fmul_rn(t, f);
  out[i] = cosf(a);
  out[half + i] = sinf(a);
}

// cos/sin tables [ntok][32]: column j in [0,n0) uses axis 0 ... ; angle = pos * freq / scale
__global__ void rope_table_kernel(float* __restrict__ cosT, float* __restrict__ sinT, const int32_t* __restrict__ p0,
                                  const int32_t* __restrict__ p1, const int32_t* __restrict__ p2, int T, int H, int W,
                                  int n0, int n1, int n2, float s0, float s1, float s2, const int32_t* __restrict__ tok_perm) {
  const int np = n0 + n1 + n2;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)T * H * W * np;
  if (gid >= total) return;
  const int j = (int)(gid % np);
  const int64_t row = gid / np;
  const int64_t tok = tok_perm ? tok_perm[row] : row;
  const int w = (int)(tok % W), h = (int)((tok / W) % H), t = (int)(tok / ((int64_t)W * H));
  int pos, i, n;
  float sc;
  if (j < n0) { pos = p0[t]; i = j; n = n0; sc = s0; }
  else if (j < n0 + n1) { pos = p1[h]; i = j - n0; n = n1; sc = s1; }
  else { pos = p2[w]; i = j - n0 - n1; n = n2; sc = s2; }
  const float f = expf(__fdiv_rn(__fmul_rn(-9.210340371976184f, (float)i), (float)n));
  const float a = __fdiv_rn(__fmul_rn((float)pos, f), sc);
  cosT[gid] = cosf(a);
  sinT[gid] = sinf(a);
}

__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int T, int H,
                                                       int W, int Cx, int Cin, int Kpad, const int32_t* __restrict__ tok_perm) {
  // one thread per output element of [Ntok][Kpad]; feature = (ph*2+pw)*Cin + c  (pt = 1)
  const int Hp = H / 2, Wp = W / 2;
  const int64_t total = (int64_t)T * Hp * Wp * Kpad;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
    const int f = (int)(g % Kpad);
    const int64_t row = g / Kpad;
    float v = 0.f;
    if (f < 4 * Cin) {
      const int64_t tok = tok_perm ? tok_perm[row] : row;
      const int wp = (int)(tok % Wp), hp = (int)((tok / Wp) % Hp), t = (int)(tok / ((int64_t)Wp * Hp));
      const int c = f % Cin, pp = f / Cin, ph = pp >> 1, pw = pp & 1;
      if (c < Cx) v = x[(((int64_t)t * H + 2 * hp + ph) * W + 2 * wp + pw) * Cx + c];
    }
    out[g] = f2bf(v);
  }
}

__global__ __launch_bounds__(256) void unpatchify_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int T,
                                                         int Hp, int Wp, int C, int ldx, const int32_t* __restrict__ tok_perm) {
  // x [Ntok][C*4] feature (c, ph, pw) -> out (T, 2Hp, 2Wp, C)
  const int F = 4 * C;
  const int64_t total = (int64_t)T * Hp * Wp * F;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
    const int f = (int)(g % F);
    const int64_t row = g / F;
    const int64_t tok = tok_perm ? tok_perm[row] : row;
    const int wp = (int)(tok % Wp), hp = (int)((tok / Wp) % Hp), t = (int)(tok / ((int64_t)Wp * Hp));
    const int c = f >> 2, ph = (f >> 1) & 1, pw = f & 1;
    out[(((int64_t)t * 2 * Hp + 2 * hp + ph) * (2 * Wp) + 2 * wp + pw) * C + c] = x[row * ldx + f];
  }
}

__global__ __launch_bounds__(256) void cfg_euler_kernel(float* __restrict__ img, const bf16_t* __restrict__ vc,
                                                        const bf16_t* __restrict__ vu, float w, float dt,
                                                        const float* __restrict__ dtvec, const int* __restrict__ step, int64_t n) {
  if (dtvec) dt = dtvec[*step];
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < n; g += (int64_t)gridDim.x * 256) {
    float v = bf2f(vc[g]);
    if (vu) {  // uncond + w * (cond - uncond), each eager bf16 op rounds (generation_utils.py:74-76)
      const float u = bf2f(vu[g]);
      v = bf_round(__fadd_rn(u, bf_round(__fmul_rn(w, bf_round(__fsub_rn(v, u))))));
    }
    img[g] = __fadd_rn(img[g], bf_round(__fmul_rn(dt, v)));  // :128, (0-dim fp32)*(bf16) -> bf16
  }
}

__global__ void step_inc_kernel(int* step) { *step += 1; }

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int64_t n) {
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < n; g += (int64_t)gridDim.x * 256) out[g] = f2bf(x[g]);
}

inline int grid_for(int64_t n, int per_block = 256, int cap = 8192) {
  int64_t b = (n + per_block - 1) / per_block;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}
inline int done() { return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP; }

}  // namespace

int k5_launch_ln_modulate(const void* x, const float* scale, const float* shift, void* out, int rows, int D,
                          int ldx, int ldo, hipStream_t s, void* out_e4m3) {
  if (rows <= 0 || D <= 0 || (!out && !out_e4m3)) return K5_ERR_ARG;
  if ((D & 7) || (ldx & 7) || (ldo & 7)) return K5_ERR_ALIGN;
  if (D > 64 * 8 * MAXC) return K5_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(ln_kernel<false>, dim3(ln_grid(rows)), dim3(256), 0, s, (const bf16_t*)x, scale, shift,
                     (bf16_t*)out, (float*)nullptr, rows, D, ldx, ldo, (uint8_t*)out_e4m3);
  return done();
}

int k5_launch_ln_affine(const void* x, const float* w, const float* b, void* out_bf16, float* out_f32, int rows,
                        int D, hipStream_t s) {
  if (rows <= 0 || D <= 0) return K5_ERR_ARG;
  if (D & 7) return K5_ERR_ALIGN;
  if (D > 64 * 8 * MAXC) return K5_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(ln_kernel<true>, dim3(ln_grid(rows)), dim3(256), 0, s, (const bf16_t*)x, w, b,
                     (bf16_t*)out_bf16, out_f32, rows, D, D, D);
  return done();
}

// Ulysses sequence parallelism (engine.hip run_self_attention_ulysses): head-group repacking around the two all-to-alls.
// pack: x [rows][2 D] = (q heads | k heads) of the rank's token rows -> out [P][slot_rows][2 Dp], block g = (q | k) of the heads rank g attends
__global__ __launch_bounds__(256) void ulysses_pack_qk_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int rows, int slot_rows, int D, int Dp) {
  const int cpr = 2 * D / 8;                                   // 16-B chunks per row
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)rows * cpr; i += (int64_t)gridDim.x * 256) {
    const int row = (int)(i / cpr), col = (int)(i % cpr) * 8;
    const int part = col >= D ? 1 : 0, cc = col - part * D, g = cc / Dp, within = cc - g * Dp;
    *reinterpret_cast<u32x4*>(out + ((size_t)g * slot_rows + row) * (2 * Dp) + part * Dp + within) = *reinterpret_cast<const u32x4*>(x + (size_t)row * 2 * D + col);
  }
}
// unpack: in [P][slot_rows][Dp] (block g = the outputs of rank g's heads for this rank's rows) -> out [rows][D]
__global__ __launch_bounds__(256) void ulysses_unpack_o_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int rows, int slot_rows, int D, int Dp) {
  const int cpr = D / 8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)rows * cpr; i += (int64_t)gridDim.x * 256) {
    const int row = (int)(i / cpr), col = (int)(i % cpr) * 8;
    const int g = col / Dp, within = col - g * Dp;
    *reinterpret_cast<u32x4*>(out + (size_t)row * D + col) = *reinterpret_cast<const u32x4*>(in + ((size_t)g * slot_rows + row) * Dp + within);
  }
}

size_t k5_rmsnorm_stats_workspace_bytes(int H) { return (size_t)4096 * 2 * H * sizeof(float); }   // partial rows of up to 2 H entries (norms + radii)

int k5_launch_rmsnorm_rope(void* x, const float* weight, const float* cosT, const float* sinT, int rows, int H,
                           int ld, const int32_t* heads_cfg, hipStream_t s, float out_scale, int scale_from_head, void* scaled_out,
                           int ld_scaled, float* stats, float* stats_ws, float* key_centre, void* mean_q, int mq_stride, void* mean_k, int mk_stride) {
  // heads_cfg (host pointer, optional): {heads_per_weight, rope_heads}; default: one weight, rope on all heads
  if (rows <= 0 || H <= 0) return K5_ERR_ARG;
  const bool means = mean_q || mean_k;
  if (means && ((rows & 63) || (mean_q && mq_stride < rows / 64) || (mean_k && mk_stride < rows / 64))) return K5_ERR_ARG;
  if (ld & 7) return K5_ERR_ALIGN;
  const int hpw = heads_cfg ? heads_cfg[0] : H;
  const int rope_heads = heads_cfg ? heads_cfg[1] : H;
  const int64_t total = (int64_t)rows * H * 8;
  // grid: a multiple of `unit` blocks so that the grid stride (grid * 256 threads) is a multiple of H * 8 threads per row
  int gcd = 256, b8 = H * 8;
  for (int a = gcd, b = b8; b;) { const int t = a % b; a = b; b = t; gcd = a; }
  const int unit = b8 / gcd;
  if (stats && (!stats_ws || H > 256)) return K5_ERR_ARG;
  int64_t cap = 4096 / unit * unit;   // with statistics: one partial row per block (k5_rmsnorm_stats_workspace_bytes)
  if (cap < unit) { if (stats) return K5_ERR_UNSUPPORTED; cap = unit; }
  int64_t blocks = (total + 255) / 256;
  if (blocks > cap) blocks = cap;
  blocks = (blocks + unit - 1) / unit * unit;           // <= cap: cap is a multiple of unit
  if (means) {   // slots (threads per head chunk) = blocks * 256 / (H * 8): every slot the same number k of 64-row blocks where the cap allows
    const int64_t nblk = rows / 64, smax_ = cap * 256 / b8;
    const int64_t k = (nblk + smax_ - 1) / smax_, want = (nblk + k - 1) / k;
    blocks = (want * b8 + 255) / 256;
    blocks = (blocks + unit - 1) / unit * unit;
    if (blocks > cap) blocks = cap;
  }
  // key_centre (nullable, with stats and scaled heads only): [H - scale_from_head][64] floats OUT = the centres of the scaled (key) heads;
  // stats then has H + (H - scale_from_head) entries: the squared norms, then the squared radii around the centres
  const bool centred = key_centre && stats && scale_from_head < H;
  if (centred)
    hipLaunchKernelGGL(key_centre_kernel, dim3(H - scale_from_head), dim3(256), 0, s, (const bf16_t*)x, weight, cosT, sinT, rows, ld, scale_from_head, hpw,
                       rope_heads, out_scale, rows < K5_CENTRE_SAMPLE ? rows : K5_CENTRE_SAMPLE, key_centre);
  const int Hs = centred ? H + (H - scale_from_head) : H;
  if (stats && Hs > 256) return K5_ERR_UNSUPPORTED;   // the statistics live in a 256-entry LDS table; without them any head count goes (cross-attention keys of all blocks: 896)
  if (means)
    hipLaunchKernelGGL(rmsnorm_rope_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, (bf16_t*)x, weight,
                       cosT, sinT, rows, H, hpw, ld, rope_heads, out_scale, scale_from_head, (bf16_t*)scaled_out, ld_scaled, stats ? stats_ws : nullptr,
                       centred ? key_centre : nullptr, (bf16_t*)mean_q, mq_stride, (bf16_t*)mean_k, mk_stride);
  else
    hipLaunchKernelGGL(rmsnorm_rope_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, (bf16_t*)x, weight,
                       cosT, sinT, rows, H, hpw, ld, rope_heads, out_scale, scale_from_head, (bf16_t*)scaled_out, ld_scaled, stats ? stats_ws : nullptr,
                       centred ? key_centre : nullptr, (bf16_t*)nullptr, 0, (bf16_t*)nullptr, 0);
  if (stats) hipLaunchKernelGGL(stats_reduce_kernel, dim3(blocks >= 2048 ? 64 : (blocks >= 512 ? 16 : 1)), dim3(256), 0, s, stats_ws, (int)blocks, Hs, stats);
  return done();
}

int k5_launch_ulysses_pack_qk(const void* x, void* out, int rows, int slot_rows, int D, int P, hipStream_t s) {
  if (rows <= 0 || slot_rows < rows || P <= 0 || D % P || (D / P) % 8) return K5_ERR_ARG;
  hipLaunchKernelGGL(ulysses_pack_qk_kernel, dim3(grid_for((int64_t)rows * (2 * D / 8))), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out, rows, slot_rows, D, D / P);
  return done();
}
int k5_launch_ulysses_unpack_o(const void* in, void* out, int rows, int slot_rows, int D, int P, hipStream_t s) {
  if (rows <= 0 || slot_rows < rows || P <= 0 || D % P || (D / P) % 8) return K5_ERR_ARG;
  hipLaunchKernelGGL(ulysses_unpack_o_kernel, dim3(grid_for((int64_t)rows * (D / 8))), dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, rows, slot_rows, D, D / P);
  return done();
}

int k5_launch_gate_sum(const void* x, const void* y, const float* gate, void* out, int rows, int D, hipStream_t s) {
  if (rows <= 0 || D <= 0) return K5_ERR_ARG;
  if (D & 7) return K5_ERR_ALIGN;
  const int64_t nch = (int64_t)rows * (D / 8);
  hipLaunchKernelGGL(gate_sum_kernel, dim3(grid_for(nch)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)y, gate,
                     (bf16_t*)out, nch, D / 8);
  return done();
}

int k5_launch_gemv_f32(const float* x, const float* W, const float* b, float* y, int N, int K, int silu_in,
                       const float* add, hipStream_t s) {
  if (N <= 0 || K <= 0) return K5_ERR_ARG;
  if (K & 3) return K5_ERR_ALIGN;
  hipLaunchKernelGGL(gemv_f32_kernel, dim3((N + 3) / 4), dim3(256), 0, s, x, W, b, y, N, K, silu_in, add);
  return done();
}

int k5_launch_time_features(float t, float* out, int D, hipStream_t s, const float* tvec, const int* step) {
  if (D <= 0 || (D & 1) || ((tvec != nullptr) != (step != nullptr))) return K5_ERR_ARG;
  hipLaunchKernelGGL(time_features_kernel, dim3((D / 2 + 255) / 256), dim3(256), 0, s, t, tvec, step, out, D);
  return done();
}

int k5_launch_step_inc(int* step, hipStream_t s) {
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, s, step);
  return done();
}

int k5_launch_rope_table(float* cosT, float* sinT, const int32_t* p0, const int32_t* p1, const int32_t* p2, int T,
                         int H, int W, int n0, int n1, int n2, float s0, float s1, float s2, const int32_t* tok_perm,
                         hipStream_t s) {
  const int64_t total = (int64_t)T * H * W * (n0 + n1 + n2);
  if (total <= 0) return K5_ERR_ARG;
  hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cosT, sinT, p0, p1, p2,
                     T, H, W, n0, n1, n2, s0, s1, s2, tok_perm);
  return done();
}

int k5_launch_patchify(const float* x, void* out, int T, int H, int W, int C, int Cin_total, int Kpad,
                       const int32_t* tok_perm, hipStream_t s) {
  if (T <= 0 || (H & 1) || (W & 1) || Kpad < 4 * Cin_total || C <= 0 || C > Cin_total) return K5_ERR_ARG;
  const int64_t total = (int64_t)T * (H / 2) * (W / 2) * Kpad;
  hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, (bf16_t*)out, T, H, W, C, Cin_total, Kpad,
                     tok_perm);
  return done();
}

int k5_launch_unpatchify(const void* x, void* out, int T, int Hp, int Wp, int C, int ldx, const int32_t* tok_perm,
                         hipStream_t s) {
  const int64_t total = (int64_t)T * Hp * Wp * 4 * C;
  if (total <= 0) return K5_ERR_ARG;
  hipLaunchKernelGGL(unpatchify_kernel, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out, T, Hp, Wp,
                     C, ldx, tok_perm);
  return done();
}

int k5_launch_cfg_euler(float* img, const void* vc, const void* vu, float w, float dt, int64_t n, hipStream_t s,
                        const float* dtvec, const int* step) {
  if (n <= 0 || ((dtvec != nullptr) != (step != nullptr))) return K5_ERR_ARG;
  hipLaunchKernelGGL(cfg_euler_kernel, dim3(grid_for(n)), dim3(256), 0, s, img, (const bf16_t*)vc, (const bf16_t*)vu, w,
                     dt, dtvec, step, n);
  return done();
}

namespace {
// weight packing on the device (k5_dit_finalize): src [rows][cols] of dtype sdt (K5_F32 / K5_BF16 / K5_F16) -> dst [rows][ld] bf16 (RNE)
// or fp32; the columns cols .. ld of a padded destination are zeroed
template <bool DST_BF16>
__global__ __launch_bounds__(256) void pack_matrix_kernel(const void* __restrict__ src, int sdt, void* __restrict__ dst, int64_t rows,
                                                          int cols, int ld) {
  const int64_t total = rows * ld;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
    const int64_t r = g / ld;
    const int c = (int)(g - r * ld);
    float v = 0.f;
    if (c < cols) {
      const int64_t si = r * cols + c;
      if (sdt == K5_F32) v = reinterpret_cast<const float*>(src)[si];
      else if (sdt == K5_BF16) v = __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(src)[si] << 16);
      else v = (float)reinterpret_cast<const _Float16*>(src)[si];
    }
    if (DST_BF16) reinterpret_cast<bf16_t*>(dst)[g] = f2bf(v);
    else reinterpret_cast<float*>(dst)[g] = v;
  }
}
}  // namespace

int k5_launch_pack_matrix(const void* src, int src_dtype, void* dst, int dst_bf16, int64_t rows, int cols, int ld, hipStream_t s) {
  if (!src || !dst || rows <= 0 || cols <= 0 || ld < cols) return K5_ERR_ARG;
  if (src_dtype != K5_F32 && src_dtype != K5_BF16 && src_dtype != K5_F16) return K5_ERR_ARG;
  const int64_t total = rows * ld;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (dst_bf16) hipLaunchKernelGGL(pack_matrix_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, src, src_dtype, dst, rows, cols, ld);
  else hipLaunchKernelGGL(pack_matrix_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, src, src_dtype, dst, rows, cols, ld);
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}

int k5_launch_cast_f32_bf16(const float* x, void* out, int64_t n, hipStream_t s) {
  if (n <= 0) return K5_ERR_ARG;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, (bf16_t*)out, n);
  return done();
}
