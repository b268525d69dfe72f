// nabla.hip — NABLA adaptive block-sparse attention map for gfx950 (device-side, no host sync).
//
// Replaces nablaT_v2 (kandinsky/models/utils.py:136-163) + fast_sta_nabla (:108-133):
//   1. block_mean_kernel   qa, ka = mean over each 64-token block (tokens are in fractal order, so a block is one
//                          8x8 spatial tile of one frame), rounded to bf16 like the reference's bf16 `.mean(-2)`;
//   2. nabla_logits_kernel + nabla_select_row_kernel  per (head, query block): logits = bf16(qa . ka_j) / 8, softmax in fp32, keep the
//                          smallest set of blocks whose probability mass is >= P  ==  drop the ascending-sorted
//                          prefix whose cumulative sum stays below 1-P (`cvals >= 1 - thr`), OR the sliding-tile
//                          window |dt|<=wT/2, |dh|<=wH/2, |dw|<=wW/2.  No sort: the cut value is found by bisection on
//                          the fp32 bit pattern (monotone for non-negative floats); entries EQUAL to the cut value are
//                          ranked by index like a stable ascending sort.  Output: one bit per (query block, kv block).
//   3. nabla_union_kernel  the attention workgroup covers 4 query blocks: OR their rows, compact the kv-block ids and
//                          attach a 4-bit membership mask (which of the 4 query blocks wants that kv block).
// The sparse attention kernel itself is attn_fwd.hip (SPARSE variant).
#include <type_traits>

#include "k5_common.h"
#include "k5_kernels.h"

namespace {

// x [N][ld] bf16 holding `heads` heads of 64 -> mean [heads][nb][64] bf16
__global__ __launch_bounds__(256) void block_mean_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int heads, int nb, int ld) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < heads * 8; c += 256) {
    const bf16_t* p = x + (size_t)b * 64 * ld + 8 * c;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < 64; ++t) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(p + (size_t)t * ld);
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[2 * j] += __uint_as_float(raw[j] << 16); acc[2 * j + 1] += __uint_as_float(raw[j] & 0xffff0000u); }
    }
    const int h = c >> 3, dc = c & 7;
    u32x4 pk = {pack_bf16x2(acc[0] * (1.f / 64), acc[1] * (1.f / 64)), pack_bf16x2(acc[2] * (1.f / 64), acc[3] * (1.f / 64)),
                pack_bf16x2(acc[4] * (1.f / 64), acc[5] * (1.f / 64)), pack_bf16x2(acc[6] * (1.f / 64), acc[7] * (1.f / 64))};
    *reinterpret_cast<u32x4*>(out + ((size_t)h * nb + b) * 64 + 8 * dc) = pk;
  }
}

// sequence parallelism: key-block means gathered as per-rank slots [P][H][slot][64] -> the map's layout [H][nb][64]
// (block b lives in slot b / slot_blocks; only the last slot may be short, so the padded index of a real block is its global index)
__global__ __launch_bounds__(256) void means_relayout_kernel(const bf16_t* __restrict__ g, bf16_t* __restrict__ out, int H, int nb, int slot_blocks) {
  const int idx = blockIdx.x * 256 + threadIdx.x;          // one 16-B chunk each: (h, b, c) with c in 0..7
  if (idx >= H * nb * 8) return;
  const int c = idx & 7, b = (idx >> 3) % nb, h = (idx >> 3) / nb;
  const int sl = b / slot_blocks, jj = b - sl * slot_blocks;
  *reinterpret_cast<u32x4*>(out + ((size_t)h * nb + b) * 64 + 8 * c) =
      *reinterpret_cast<const u32x4*>(g + (((size_t)sl * H + h) * slot_blocks + jj) * 64 + 8 * c);
}

struct SelP {
  const bf16_t* qa; const bf16_t* ka;   // [H][nb][64]
  unsigned long long* bits;             // [H][nb][nw]
  int* kv_nb;                           // [H][nb] kept blocks per row (diagnostics / density)
  int H, nb, nw, T, Hb, Wb, wT, wH, wW;
  int nqb, qb0;                         // query blocks handled here: global blocks [qb0, qb0 + nqb) (sequence parallel: the rank's rows)
  float target;                         // 1 - P
  unsigned magic_hw, magic_w;           // ceil(2^32 / (Hb Wb)), ceil(2^32 / Wb): floor(n / d) = umulhi(n, magic) while n d < 2^32
};

constexpr int SEL_MAXNB = 4096;
// values per lane of the row-per-wave selection kernel's instantiations (nabla_select_row_kernel<NV>) for a row of nw 64-block words
inline int sel_row_nv(int nw) { return nw <= 4 ? 4 : nw <= 8 ? 8 : nw <= 16 ? 16 : nw <= 24 ? 24 : nw <= 32 ? 32 : nw <= 48 ? 48 : 64; }

// sum over the 64 lanes, the same value returned in every lane.  DPP inside the 16-lane rows (quad swaps, half-mirror,
// mirror), then two readlanes across the rows — no LDS round trips (the bisection below does 30 of these per row).
K5_DEV float wave_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
  // every lane of a 16-lane row now holds the row's sum
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}
K5_DEV float wave_max_dpp(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true)));
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The map in two kernels (round 3).  Rounds 1-2 fused logits and selection: R = 4 / 2 rows per wave in registers so that a key mean was
// loaded once per R rows — and still every wave walked the head's whole key-mean matrix (468 KB at 3660 blocks: 6 GB of L2 reads per
// layer on a 4-GPU shard of the 1280x768 clip), at one wave per SIMD for the long rows (256 VGPRs: nothing hid the DPP reductions).
// Now the logits are what they are in the reference — one bf16 matmul per head (utils.py:145-147) — written once (2 B per entry:
// 187 MB per layer on that shard), and the selection reads its own row back: one wave per row, NV values per lane, <= 128 VGPRs,
// 4-8 waves per SIMD.  Map per step: 54.8 -> 18.9 ms on that shard, 23.1 -> 14.7 ms on the 768x512 10 s clip (same box, A/B).
// ---------------------------------------------------------------------------------------------------------------------------------

// logits[h][il][j] = bf16(qa[h][il] . ka[h][j])  (bf16 matmul output; the exact / 8 happens in the consumer).  MFMA 16x16x32 with the
// KEY means as the A operand: a lane then holds 4 keys of ONE query row per tile, and with the A rows of tile kt taken from keys
// 16 (i >> 2) + 4 kt + (i & 3) of a 64-key window its four tiles are 16 CONSECUTIVE keys: two 16-byte stores per lane, 128 contiguous
// bytes per row and wave.  Workgroup = 4 waves = 16 query rows x 4 windows of 64 keys per step.  K = 64: operands straight from global.
__global__ __launch_bounds__(256) void nabla_logits_kernel(const bf16_t* __restrict__ qa, const bf16_t* __restrict__ ka, bf16_t* __restrict__ out,
                                                           int H, int nqb, int nb, int ldl, int qtiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
  const int h = blockIdx.x / qtiles, q0 = (blockIdx.x % qtiles) * 16;
  const bf16_t* qp = qa + ((size_t)h * nqb + min(q0 + l15, nqb - 1)) * 64 + 8 * g;
  const bf16x8 qf0 = *reinterpret_cast<const bf16x8*>(qp), qf1 = *reinterpret_cast<const bf16x8*>(qp + 32);   // B: [k = 8 g ..][j = query l15]
  const bf16_t* kh = ka + (size_t)h * nb * 64;
  bf16_t* orow = out + ((size_t)h * nqb + q0 + l15) * ldl;
  const int nwin = ldl / 64;                          // >= ceil(nb / 64): the consumer's row width (64 values per lane x NV)
  for (int w = blockIdx.y * 4 + wave; w < nwin; w += gridDim.y * 4) {
    f32x4 acc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int key = min(64 * w + 16 * (l15 >> 2) + 4 * kt + (l15 & 3), nb - 1);      // A row i = l15 of tile kt
      const bf16_t* kp = kh + (size_t)key * 64 + 8 * g;
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(kp), a1 = *reinterpret_cast<const bf16x8*>(kp + 32);
      acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qf0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qf1, acc[kt], 0, 0, 0);
    }
    // lane (query l15, g): acc[kt][r] = row 4 g + r of tile kt = key 64 w + 16 g + 4 kt + r
    if (64 * w + 64 > nb) {                             // last window: -inf into the padded columns (exp -> 0 in the consumer)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (64 * w + 16 * g + 4 * kt + r >= nb) acc[kt][r] = -__builtin_inff();
    }
    if (q0 + l15 < nqb) {
      const u32x4 lo = {pack_bf16x2(acc[0][0], acc[0][1]), pack_bf16x2(acc[0][2], acc[0][3]), pack_bf16x2(acc[1][0], acc[1][1]), pack_bf16x2(acc[1][2], acc[1][3])};
      const u32x4 hi = {pack_bf16x2(acc[2][0], acc[2][1]), pack_bf16x2(acc[2][2], acc[2][3]), pack_bf16x2(acc[3][0], acc[3][1]), pack_bf16x2(acc[3][2], acc[3][3])};
      u32x4* dst = reinterpret_cast<u32x4*>(orow + 64 * w + 16 * g);
      dst[0] = lo; dst[1] = hi;
    }
  }
}

// One wave = one (head, query block) row; lane l owns key blocks l, l + 64, ... (NV per lane), all in registers; reductions by DPP
// inside the 16-lane rows + readlanes (no LDS).  The bisection starts from the row's own [min p, max p] bit range instead of
// [0, +inf) (~25 steps instead of 31) after one check that the cut exists at all.
template <int NV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void nabla_select_row_kernel(SelP p, const bf16_t* __restrict__ logits, int ldl) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rows_per_head = (p.nqb + 3) / 4 * 4;
  const int gr = blockIdx.x * 4 + wave;
  const int h = gr / rows_per_head, il = gr % rows_per_head;
  if (il >= p.nqb) return;   // wave-uniform; no barrier in this kernel
  const bf16_t* lrow = logits + ((size_t)h * p.nqb + il) * ldl;
  // Padding needs no per-value lane masks (64 SGPR pairs of them spill) and no tests at all: a logits row is 64 NV columns wide and
  // the logits kernel wrote -inf into the columns nb .. 64 NV, so a padded entry is exp(-inf) = 0 — zero mass on either side of any cut.
  float pv[NV];
  float mx = -3.0e38f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const float lg = (float)lrow[v * 64 + lane] * 0.125f;          // bf16 matmul output, then the exact / sqrt(64); ldl = 64 NV
    pv[v] = lg;
    mx = fmaxf(mx, lg);
  }
  mx = wave_max_dpp(mx);
  float sum = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) { const float e = expf(pv[v] - mx); pv[v] = e; sum += e; }
  sum = wave_sum_dpp(sum);
  const float inv = 1.0f / sum;
  float pmin = 3.0e38f, tot = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    pv[v] = pv[v] * inv;
    tot += pv[v];
    pmin = fminf(pmin, pv[v] > 0.f ? pv[v] : 3.0e38f);              // smallest POSITIVE value: zeros (padding, underflow) carry no mass
  }
  tot = wave_sum_dpp(tot);                            // = g(max p): the invariant's upper end
  pmin = -wave_max_dpp(-pmin);
  // smallest value v* with  sum_{p <= v*} p >= target  (bisection over the bit pattern of non-negative floats; invariant
  // g(lo - 1) < target <= g(hi)).  v* is one of the row's values, so [min p, max p] brackets it — unless even the whole row's mass
  // stays below the target (P ~ 0 and rounding): then, as with the [0, +inf) bracket, nothing lies above the cut.
  unsigned lo = __float_as_uint(pmin), hi = __float_as_uint(inv);          // max p = exp(0) * inv exactly
  if (!(tot >= p.target)) lo = hi = 0x7f800000u;
  if (!(p.target > 0.f)) hi = lo;                     // g(min p) >= min p > 0 >= target
  // g(mid) = sum of the values whose bit pattern is <= mid, per lane in index order into ONE accumulator (the order of the one-kernel
  // form above), by EXEC masking: v_cmpx writes the lanes that take part straight into EXEC and a plain v_add follows — two VALU
  // instructions per value and no SGPR-pair traffic (compare + select + add through SGPR masks: three, and the masks of an unrolled
  // 64-value row spill).  EXEC is restored from a saved copy before every compare (v_cmpx overwrites the active lanes' bits).
  while (lo < hi) {                                   // wave-uniform
    const unsigned mid = lo + ((hi - lo) >> 1);
    float gl = 0.f;
#pragma unroll
    for (int v = 0; v < NV; v += 4) {
      unsigned long long sv;
      asm volatile("s_mov_b64 %1, exec\n\t"
                   "v_cmpx_ge_u32_e32 %2, %3\n\t"
                   "v_add_f32_e32 %0, %0, %3\n\t"
                   "s_mov_b64 exec, %1\n\t"
                   "v_cmpx_ge_u32_e32 %2, %4\n\t"
                   "v_add_f32_e32 %0, %0, %4\n\t"
                   "s_mov_b64 exec, %1\n\t"
                   "v_cmpx_ge_u32_e32 %2, %5\n\t"
                   "v_add_f32_e32 %0, %0, %5\n\t"
                   "s_mov_b64 exec, %1\n\t"
                   "v_cmpx_ge_u32_e32 %2, %6\n\t"
                   "v_add_f32_e32 %0, %0, %6\n\t"
                   "s_mov_b64 exec, %1"
                   : "+v"(gl), "=&s"(sv)
                   : "s"(mid), "v"(pv[v]), "v"(pv[v + 1]), "v"(pv[v + 2]), "v"(pv[v + 3])
                   : "vcc");
    }
    const float gsum = wave_sum_dpp(gl);
    if (gsum >= p.target) hi = mid; else lo = mid + 1;
  }
  const int i = p.qb0 + il;
  const unsigned vbits = lo;
  const float vstar = __uint_as_float(vbits);
  float base = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) base += __float_as_uint(pv[v]) < vbits ? pv[v] : 0.f;
  base = wave_sum_dpp(base);
  // ties at v*: a stable ascending sort orders them by index; the m-th tie has cumsum base + m*v*
  int m0 = 1;
  if (vstar > 0.f) { const float need = (p.target - base) / vstar; m0 = (int)ceilf(need); if (m0 < 1) m0 = 1; }
  const int hw_blocks = p.Hb * p.Wb;
  const int ti = i / hw_blocks, hi_ = (i / p.Wb) % p.Hb, wi = i % p.Wb;
  int tie_seen = 0, kept = 0;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    if (c < p.nw) {
      const int j = c * 64 + lane;
      bool keep = false, tie = false;
      if (c + 1 < p.nw || j < p.nb) {                  // only the last word can hold padding
        const unsigned vb = __float_as_uint(pv[c]);
        tie = vb == vbits;
        keep = vb > vbits;
        const int tj = hw_blocks == 1 ? j : (int)__umulhi((unsigned)j, p.magic_hw), rj = j - tj * hw_blocks;
        const int hj = p.Wb == 1 ? rj : (int)__umulhi((unsigned)rj, p.magic_w), wj = rj - hj * p.Wb;
        keep = keep || (abs(ti - tj) <= p.wT / 2 && abs(hi_ - hj) <= p.wH / 2 && abs(wi - wj) <= p.wW / 2);
      }
      const unsigned long long tmask = __ballot(tie);
      if (tmask) {   // wave-uniform and rare: values equal to the cut
        if (tie) {
          const int rank = tie_seen + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(tmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)tmask, 0u)) + 1;
          keep = keep || rank >= m0;
        }
        tie_seen += __popcll(tmask);
      }
      const unsigned long long w = __ballot(keep);
      kept += __popcll(w);
      if (lane == 0) p.bits[((size_t)h * p.nqb + il) * p.nw + c] = w;
      __builtin_amdgcn_sched_barrier(0);   // one word at a time: left free, the scheduler hoists every word's coordinate math and ballots (218 VGPRs, 86 spilled SGPRs at NV = 64)
    }
  }
  if (lane == 0 && p.kv_nb) p.kv_nb[h * p.nqb + il] = kept;
}

// per (h, group of 4 query blocks): list[(h*ng+g)*nb + e] = kv_block | membership << 24 ; cnt[h*ng+g].
// Sequence parallelism: the key blocks [loc0, loc0 + locn) — the rank's own, already in place before the gather — come FIRST and
// cnt_local[h*ng+g] says how many they are: the attention runs them as a first pass while the other ranks' keys travel.
__global__ __launch_bounds__(256) void nabla_union_kernel(const unsigned long long* __restrict__ bits, int* __restrict__ list,
                                                          int* __restrict__ cnt, int* __restrict__ cnt_local, int H, int nqb, int nb,
                                                          int nw, int ng, int loc0, int locn, int G, int pair_stride) {
  const int lane = threadIdx.x & 63;
  const int gi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gi >= H * ng) return;
  const int h = gi / ng, g = gi % ng;
  int pos = 0;
  // the rows' bitmaps once, word `lane` of each row in lane `lane` (nw <= 64: SEL_MAXNB) — the loops below broadcast a word with a readlane
  // instead of a dependent global load per word and sweep (round 4: 115 -> 67 us per layer on a 4-GPU shard of the 3660-block clip)
  unsigned long long wrow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qb = (G == 2 && pair_stride > 0) ? k5_pair_row(g, r, pair_stride, nqb) : G * g + r;
    wrow[r] = (r < G && qb < nqb && lane < nw) ? bits[((size_t)h * nqb + qb) * nw + lane] : 0ull;
  }
  for (int sweep = locn > 0 ? 0 : 1; sweep < 2; ++sweep) {
    for (int c = 0; c < nw; ++c) {
      // bit mask of the local blocks inside word c
      const int lo = max(loc0 - 64 * c, 0), hi = min(loc0 + locn - 64 * c, 64);
      unsigned long long lm = 0ull;
      if (locn > 0 && hi > lo) lm = (hi - lo == 64 ? ~0ull : ((1ull << (hi - lo)) - 1ull)) << lo;
      const unsigned long long take = sweep == 0 ? lm : ~lm;
      if (!take) continue;
      unsigned long long w[4], u = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned int wl = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)wrow[r], c);
        const unsigned int wh = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(wrow[r] >> 32), c);
        w[r] = (((unsigned long long)wh << 32) | wl) & take;
        u |= w[r];
      }
      if ((u >> lane) & 1ull) {
        int mem = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) mem |= (int)((w[r] >> lane) & 1ull) << r;
        list[(size_t)gi * nb + pos + __popcll(u & ((1ull << lane) - 1ull))] = (c * 64 + lane) | (mem << 24);
      }
      pos += __popcll(u);
    }
    if (sweep == 0 && lane == 0) cnt_local[gi] = pos;
  }
  if (lane == 0) { cnt[gi] = pos; if (locn <= 0) cnt_local[gi] = 0; }
}

__global__ __launch_bounds__(256) void nabla_expand_kernel(const unsigned long long* __restrict__ bits, unsigned char* __restrict__ out,
                                                           int64_t rows, int nb, int nw) {
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < rows * nb; g += (int64_t)gridDim.x * 256) {
    const int j = (int)(g % nb);
    const int64_t r = g / nb;
    out[g] = (unsigned char)((bits[r * nw + (j >> 6)] >> (j & 63)) & 1ull);
  }
}

// sum of the per-row kept-block counts -> one atomic per workgroup (profiling only: realised map density)
// ng > 0: entry i is the list of row group i % ng of its head and counts once per ROW of that group (group_rows of them, fewer in a head's last group)
__global__ __launch_bounds__(256) void nabla_count_kernel(const int* __restrict__ kv_nb, int rows, unsigned long long* acc, int ng, int group_rows, int nqb) {
  unsigned long long v = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows; i += gridDim.x * 256) {
    const int mul = ng > 0 ? min(group_rows, nqb - (i % ng) * group_rows) : 1;
    v += (unsigned long long)kv_nb[i] * (unsigned long long)mul;
  }
  __shared__ unsigned long long part[4];
  float dummy = 0.f; (void)dummy;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
}

}  // namespace

int k5_launch_nabla_count(const void* workspace, int H, int nqb, int nb, unsigned long long* acc, hipStream_t s) {
  if (!workspace || !acc || H <= 0 || nqb <= 0) return K5_ERR_ARG;
  const int* kv_nb;
  k5_nabla_workspace_views(const_cast<void*>(workspace), H, nb, nullptr, &kv_nb, nullptr, nullptr);
  const int rows = H * nqb;
  hipLaunchKernelGGL(nabla_count_kernel, dim3((rows + 4095) / 4096), dim3(256), 0, s, kv_nb, rows, acc, 0, 1, 0);
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}

// profiling: key tiles the list-driven attention EXECUTES = sum over (head, row group) of the union list's length (x the rows of a group:
// every row of the group computes or masks every tile of the list) — against k5_launch_nabla_count (tiles the rows WANT) this is the
// launch's union efficiency
int k5_launch_nabla_count_lists(const void* workspace, int H, int nqb, int nb, int group_rows, unsigned long long* acc, hipStream_t s) {
  if (!workspace || !acc || H <= 0 || nqb <= 0 || group_rows <= 0) return K5_ERR_ARG;
  const int* cnt;
  k5_nabla_workspace_views(const_cast<void*>(workspace), H, nb, nullptr, nullptr, nullptr, &cnt);
  const int ng = (nqb + group_rows - 1) / group_rows, n = H * ng;
  hipLaunchKernelGGL(nabla_count_kernel, dim3((n + 4095) / 4096), dim3(256), 0, s, cnt, n, acc, ng, group_rows, nqb);   // executed blocks = list length x the rows that walk it
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}

int k5_launch_nabla_mask_u8(const void* workspace, int H, int nqb, int nb, void* out, hipStream_t s) {
  if (H <= 0 || nqb <= 0 || nqb > nb) return K5_ERR_ARG;
  const unsigned long long* bits;
  k5_nabla_workspace_views(const_cast<void*>(workspace), H, nb, &bits, nullptr, nullptr, nullptr);
  const int64_t rows = (int64_t)H * nqb;
  int64_t blocks = (rows * nb + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(nabla_expand_kernel, dim3((unsigned)blocks), dim3(256), 0, s, bits, (unsigned char*)out, rows, nb, (nb + 63) / 64);
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}

// where the block means live in the workspace: qa [H][nqb][64] (indexed with the SELECTION's nqb), ka [H][nb][64]
void k5_nabla_workspace_means(void* workspace, int H, int nb, void** qa, void** ka) {
  if (qa) *qa = workspace;
  if (ka) *ka = (char*)workspace + (size_t)H * nb * 64 * 2;
}

// nqb = query-block rows the selection will handle (a sequence-parallel rank selects nb / P of the nb rows): only the bf16 logits matrix —
// by far the largest region, 840 MB at 3660 blocks — scales with it; every other region keeps its nb-row size (the views index by nb)
// Round 5: the key-tile lists are the LAST region and sized by the lists that will exist — list_rows per head, one per group of rows
// (0 = one per selected row, the largest case) — instead of nb x nb entries per head whoever asks (1.5 GB at 3660 blocks, on every rank).
size_t k5_nabla_workspace_bytes(int H, int nb, int nqb, int list_rows) {
  const size_t nw = (nb + 63) / 64;
  if (nqb <= 0 || nqb > nb) nqb = nb;
  if (list_rows <= 0 || list_rows > nqb) list_rows = nqb;
  return (size_t)2 * H * nb * 64 * 2      // qa, ka
         + (size_t)H * nb * nw * 8         // bits
         + (size_t)H * nb * 4              // kv_nb
         + (size_t)2 * H * nb * 4 + 256                // counts, counts of the leading local entries (sequence parallelism)
         + (size_t)H * (nqb + 4) * sel_row_nv((int)nw) * 64 * 2    // bf16 block logits [H][nqb rows][64 NV] (round 3: one matmul per head, read back per row)
         + (size_t)H * list_rows * nb * 4;             // key-tile lists [H][groups][nb]
}

// q: [Nq][ldq] bf16 = the query rows handled here (global 64-token blocks [q_block0, q_block0 + Nq/64)), k: [N][ldk] = all
// keys; both after norm_qk + RoPE, fractal token order.  Fills the workspace (regions sized for Nq == N, rows indexed by
// the local query block) with: qa|ka means, block bitmap, per-row counts, per-workgroup union lists.
// block means of `nblocks` 64-token blocks of x [nblocks * 64][ld] (H heads of 64) -> out[(h * stride_blocks + b) * 64 ...]
int k5_launch_nabla_block_means(const void* x, int ld, int H, int nblocks, int stride_blocks, void* out, hipStream_t s) {
  if (H <= 0 || nblocks <= 0 || stride_blocks < nblocks || (ld & 7)) return K5_ERR_ARG;
  hipLaunchKernelGGL(block_mean_kernel, dim3(nblocks), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out, H, stride_blocks, ld);
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}

// gathered per-rank means [P][H][slot_blocks][64] -> the key-means region of `workspace` ([H][nb][64]); then
// k5_launch_nabla_select_rect(..., k = nullptr) selects from them
int k5_launch_nabla_key_means_from_slots(const void* gathered, int H, int nb, int slot_blocks, void* workspace, hipStream_t s) {
  if (H <= 0 || nb <= 0 || slot_blocks <= 0 || !gathered || !workspace) return K5_ERR_ARG;
  bf16_t* ka = (bf16_t*)((char*)workspace + (size_t)H * nb * 64 * 2);
  hipLaunchKernelGGL(means_relayout_kernel, dim3((H * nb * 8 + 255) / 256), dim3(256), 0, s, (const bf16_t*)gathered, ka, H, nb, slot_blocks);
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}

// k == nullptr: the key-block means are already in the workspace (k5_launch_nabla_key_means_from_slots, or k5_launch_rmsnorm_rope's mean_k);
// q == nullptr: so are the query-block means (k5_launch_rmsnorm_rope's mean_q into k5_nabla_workspace_means(..).qa)
int k5_launch_nabla_select_rect(const void* q, const void* k, int ldq, int ldk, int H, int Nq, int q_block0, int N, int T, int Hb,
                                int Wb, int wT, int wH, int wW, float P, void* workspace, hipStream_t s, int local_block0, int local_blocks,
                                int group_rows, int pair_stride) {
  if (group_rows != 1 && group_rows != 2 && group_rows != 4) return K5_ERR_ARG;
  if (pair_stride < 0 || (pair_stride > 0 && group_rows != 2)) return K5_ERR_ARG;
  if (local_blocks < 0 || local_block0 < 0 || (local_blocks > 0 && local_block0 + local_blocks > N / 64)) return K5_ERR_ARG;
  if (H <= 0 || N <= 0 || Nq <= 0 || (N % 64) || (Nq % 64) || T * Hb * Wb * 64 != N) return K5_ERR_ARG;
  if (q_block0 < 0 || q_block0 * 64 + Nq > N) return K5_ERR_ARG;
  if ((q && (ldq & 7)) || (k && (ldk & 7))) return K5_ERR_ALIGN;
  const int nb = N / 64, nqb = Nq / 64, nw = (nb + 63) / 64, ng = (nqb + group_rows - 1) / group_rows;
  if (nb > SEL_MAXNB) return K5_ERR_UNSUPPORTED;
  char* ws = (char*)workspace;
  bf16_t* qa = (bf16_t*)ws; ws += (size_t)H * nb * 64 * 2;
  bf16_t* ka = (bf16_t*)ws; ws += (size_t)H * nb * 64 * 2;
  unsigned long long* bits = (unsigned long long*)ws; ws += (size_t)H * nb * nw * 8;
  int* kv_nb = (int*)ws; ws += (size_t)H * nb * 4;
  int* cnt = (int*)ws;
  int* cnt_local = cnt + (size_t)H * nb;
  bf16_t* logits = (bf16_t*)(((uintptr_t)(cnt_local + (size_t)H * nb) + 255) & ~(uintptr_t)255);   // 256-B aligned: inside the + 256 slack
  int* list = (int*)(logits + (size_t)H * (nqb + 4) * sel_row_nv(nw) * 64);                           // last region: [H][ng][nb] (k5_nabla_workspace_views)
  if (q) hipLaunchKernelGGL(block_mean_kernel, dim3(nqb), dim3(256), 0, s, (const bf16_t*)q, qa, H, nqb, ldq);
  if (k) hipLaunchKernelGGL(block_mean_kernel, dim3(nb), dim3(256), 0, s, (const bf16_t*)k, ka, H, nb, ldk);
  SelP p;
  p.qa = qa; p.ka = ka; p.bits = bits; p.kv_nb = kv_nb; p.H = H; p.nb = nb; p.nw = nw; p.T = T; p.Hb = Hb; p.Wb = Wb;
  p.wT = wT; p.wH = wH; p.wW = wW; p.target = (float)(1.0 - (double)P);
  p.nqb = nqb; p.qb0 = q_block0;
  p.magic_hw = (unsigned)(((1ull << 32) + (unsigned long long)(Hb * Wb) - 1) / (unsigned long long)(Hb * Wb));
  p.magic_w = (unsigned)(((1ull << 32) + (unsigned long long)Wb - 1) / (unsigned long long)Wb);
  // values per lane NV = ceil(nb / 64) rounded up to an instantiated size
  const int nv = nw;
  {
    const int ldl = sel_row_nv(nw) * 64, qtiles = (nqb + 15) / 16;
    const int ysplit = nw >= 16 ? 4 : 1;                      // long rows: four workgroups share a (head, 16-row) strip
    hipLaunchKernelGGL(nabla_logits_kernel, dim3(H * qtiles, ysplit), dim3(256), 0, s, qa, ka, logits, H, nqb, nb, ldl, qtiles);
    const int rows_per_head = (nqb + 3) / 4 * 4;
    auto launch_rows = [&](auto NVC) {
      constexpr int NV = decltype(NVC)::value;
      hipLaunchKernelGGL((nabla_select_row_kernel<NV>), dim3(H * rows_per_head / 4), dim3(256), 0, s, p, logits, ldl);
    };
    switch (sel_row_nv(nv)) {
      case 4: launch_rows(std::integral_constant<int, 4>{}); break;
      case 8: launch_rows(std::integral_constant<int, 8>{}); break;
      case 16: launch_rows(std::integral_constant<int, 16>{}); break;
      case 24: launch_rows(std::integral_constant<int, 24>{}); break;
      case 32: launch_rows(std::integral_constant<int, 32>{}); break;
      case 48: launch_rows(std::integral_constant<int, 48>{}); break;
      default: launch_rows(std::integral_constant<int, 64>{}); break;
    }
  }
  hipLaunchKernelGGL(nabla_union_kernel, dim3((H * ng + 3) / 4), dim3(256), 0, s, bits, list, cnt, cnt_local, H, nqb, nb, nw, ng, local_block0,
                     local_blocks, group_rows, pair_stride);
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}

int k5_launch_nabla_select(const void* q, const void* k, int ldq, int ldk, int H, int N, int T, int Hb, int Wb, int wT, int wH,
                           int wW, float P, void* workspace, hipStream_t s) {
  return k5_launch_nabla_select_rect(q, k, ldq, ldk, H, N, 0, N, T, Hb, Wb, wT, wH, wW, P, workspace, s, 0, 0, 4);
}

// views into the workspace filled above
void k5_nabla_workspace_views(void* workspace, int H, int nb, const unsigned long long** bits, const int** kv_nb, const int** list,
                              const int** cnt, const int** cnt_local, int nqb) {
  const size_t nw = (nb + 63) / 64;
  if (nqb <= 0 || nqb > nb) nqb = nb;
  char* ws = (char*)workspace + (size_t)2 * H * nb * 64 * 2;
  if (bits) *bits = (const unsigned long long*)ws;
  ws += (size_t)H * nb * nw * 8;
  if (kv_nb) *kv_nb = (const int*)ws;
  ws += (size_t)H * nb * 4;
  if (cnt) *cnt = (const int*)ws;
  if (cnt_local) *cnt_local = (const int*)ws + (size_t)H * nb;
  const bf16_t* logits = (const bf16_t*)(((uintptr_t)((const int*)ws + (size_t)2 * H * nb) + 255) & ~(uintptr_t)255);
  if (list) *list = (const int*)(logits + (size_t)H * (nqb + 4) * sel_row_nv((int)nw) * 64);   // behind the logits of the nqb rows selected here
}
