// nabla.hip — NABLA adaptive block-sparse attention map for gfx950 (device-side, no host sync).
//
// Replaces nablaT_v2 (kandinsky/models/utils.py:136-163) + fast_sta_nabla (:108-133):
//   1. block_mean_kernel   qa, ka = mean over each 64-token block (tokens are in fractal order, so a block is one
//                          8x8 spatial tile of one frame), rounded to bf16 like the reference's bf16 `.mean(-2)`;
//   2. nabla_select_kernel per (head, query block): logits = bf16(qa . ka_j) / 8, softmax in fp32, keep the
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

// sum over the 64 lanes, the same value returned in every lane.  DPP inside the 16-lane rows (quad swaps, half-mirror,
// mirror), then two readlanes across the rows — no LDS round trips (the bisection below does 30 of these per row).
K5_DEV float wave_sum_dpp(float v) {
#ifdef K5_NABLA_SHFL
  return wave_sum(v);
#endif
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
#ifdef K5_NABLA_SHFL
  return wave_max(v);
#endif
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

// One wave = R query-block rows of ONE head; lane l owns the key blocks l, l + 64, ... (NV per lane) of each of its rows, all
// in registers.  Round 1 gave every row its own wave and re-read the head's whole key-mean matrix (187 KB at nb = 1464) from
// L2 per row — 7.7 GB per layer, the kernel's time; here a key mean is loaded once per R rows, the dot products run on
// v_dot2c_f32_bf16 (both operands ARE bf16), and the bisection works on registers with DPP reductions, the R rows in
// lockstep (R independent reduction chains in flight).  1.02 -> 0.69 ms per layer at nb = 1464 (profiles/r02_nabla_kernel_stats.md);
// with R = 4 the dot products moved to v_mfma_f32_4x4x4_16b_bf16 (below): 0.69 -> 0.63; the bisection without per-row branches and with
// its compares in SGPR pairs: -> 0.56.
template <int NV, int R>
__global__ __launch_bounds__(256) void nabla_select_kernel(SelP p) {
  __shared__ __attribute__((aligned(16))) bf16_t sq[4 * R * 64];   // the block's 4 R query-block means
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rows_per_block = 4 * R;
  const int blocks_per_head = (p.nqb + rows_per_block - 1) / rows_per_block;
  const int h = blockIdx.x / blocks_per_head, il0 = (blockIdx.x % blocks_per_head) * rows_per_block + wave * R;
  // stage the query means (rows past nqb: clamped, their results are dropped)
  for (int c = threadIdx.x; c < rows_per_block * 8; c += 256) {
    const int r = c >> 3, il = min((blockIdx.x % blocks_per_head) * rows_per_block + r, p.nqb - 1);
    *reinterpret_cast<u32x4*>(sq + r * 64 + 8 * (c & 7)) = *reinterpret_cast<const u32x4*>(p.qa + ((size_t)h * p.nqb + il) * 64 + 8 * (c & 7));
  }
  __syncthreads();
  if (il0 >= p.nqb) return;   // wave-uniform; no barrier follows
  float pv[R][NV];
  float mx[R];
#pragma unroll
  for (int r = 0; r < R; ++r) mx[r] = -3.0e38f;
  // logits: bf16(qa . ka_j) / sqrt(64)   (bf16 matmul output, then an exact /8)
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int j = v * 64 + lane;
    const bf16_t* kp = p.ka + ((size_t)h * p.nb + min(j, p.nb - 1)) * 64;
    u32x4 kk[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) kk[c] = *reinterpret_cast<const u32x4*>(kp + 8 * c);
    if constexpr (R == 4 || R == 2) {
      // four query rows (R = 2: the two rows twice) x this lane's key on the matrix core: v_mfma_f32_4x4x4_16b_bf16 = 16 independent 4x4x4 blocks, block b = lane >> 2.
      // A: lane (b, i) holds query row i's k-slice, B: lane (b, j) holds key 4 b + j's k-slice, D: lane (b, j) register i = row i x its
      // key — exactly the pv[r][v] layout (tools/probes/mfma_4x4x4_layout.hip).  16 k-steps of 4 instead of 4 x 32 quarter-rate
      // v_dot2c per key (49 k of the kernel's ~145 k cycles per wave).
      typedef __attribute__((ext_vector_type(4))) short bf16x4s;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const bf16_t* qrow = sq + (wave * R + (lane & (R - 1))) * 64;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const u32x4 qq = *reinterpret_cast<const u32x4*>(qrow + 8 * c);
        const u32x2 alo = {qq[0], qq[1]}, ahi = {qq[2], qq[3]}, blo = {kk[c][0], kk[c][1]}, bhi = {kk[c][2], kk[c][3]};
        acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(bf16x4s, alo), __builtin_bit_cast(bf16x4s, blo), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(bf16x4s, ahi), __builtin_bit_cast(bf16x4s, bhi), acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float lg = j < p.nb ? bf_round(acc[r]) * 0.125f : -3.0e38f;
        pv[r][v] = lg;
        mx[r] = fmaxf(mx[r], lg);
      }
    } else
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const u32x4 qq = *reinterpret_cast<const u32x4*>(sq + (wave * R + r) * 64 + 8 * c);   // same address in every lane: broadcast
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#ifdef K5_NABLA_FMA
          d = fmaf(__uint_as_float(qq[e] << 16), __uint_as_float(kk[c][e] << 16), d);
          d = fmaf(__uint_as_float(qq[e] & 0xffff0000u), __uint_as_float(kk[c][e] & 0xffff0000u), d);
#else
          // inline asm on purpose: the builtin fed through bit_cast(vector element) was compiled with the element index collapsed
          // to 0 (four identical v_dot2c per 16-byte chunk, ROCm 7.2 clang) — the map then selects from wrong logits
          asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(d) : "v"(qq[e]), "v"(kk[c][e]));
#endif
        }
      }
      const float lg = j < p.nb ? bf_round(d) * 0.125f : -3.0e38f;
      pv[r][v] = lg;
      mx[r] = fmaxf(mx[r], lg);
    }
  }
  float target_base[R];
  unsigned lo[R], hi[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    mx[r] = wave_max_dpp(mx[r]);
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) { const float e = v * 64 + lane < p.nb ? expf(pv[r][v] - mx[r]) : 0.f; pv[r][v] = e; sum += e; }
    sum = wave_sum_dpp(sum);
#pragma unroll
    for (int v = 0; v < NV; ++v) pv[r][v] = pv[r][v] / sum;   // padding lanes: 0 / sum = 0, below every cut, excluded at emission
    lo[r] = 0u; hi[r] = 0x7f800000u;                          // invariant: g(lo-1) < target <= g(hi)
    target_base[r] = 0.f;
  }
  // smallest value v* with  sum_{p <= v*} p  >= target  (bisection over the bit pattern of non-negative floats), R rows in lockstep
  // The interval [0, 0x7f800000] halves every step, so every row takes the same 31 steps: no per-row branch — the R rows' compare /
  // select / add chains and DPP reductions sit in ONE basic block and interleave (with a branch per row every v_cmp -> v_cndmask pair
  // went through VCC back to back: 29 hazard nops per row and step).  Padding lanes hold exactly 0: they add nothing on either side.
  bool more = true;
  while (more) {
    float g[R];
    unsigned mid[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { mid[r] = lo[r] + ((hi[r] - lo[r]) >> 1); g[r] = 0.f; }
    // four compares into four SGPR pairs, then the four selects: through VCC the compiler emits v_cmp / s_nop / v_cndmask one element at
    // a time (gfx950 wants two wait states between a VALU write of a mask and the VALU that reads it; here three instructions lie between)
    auto select4 = [](unsigned ma, unsigned mb, unsigned mc, unsigned md, float a, float b, float c, float d, float& ta, float& tb,
                      float& tc, float& td) __attribute__((always_inline)) {
      unsigned long long m0, m1, m2, m3;
      asm("v_cmp_ge_u32_e64 %4, %8, %12\n\t"
          "v_cmp_ge_u32_e64 %5, %9, %13\n\t"
          "v_cmp_ge_u32_e64 %6, %10, %14\n\t"
          "v_cmp_ge_u32_e64 %7, %11, %15\n\t"
          "v_cndmask_b32_e64 %0, 0, %12, %4\n\t"
          "v_cndmask_b32_e64 %1, 0, %13, %5\n\t"
          "v_cndmask_b32_e64 %2, 0, %14, %6\n\t"
          "v_cndmask_b32_e64 %3, 0, %15, %7"
          : "=&v"(ta), "=&v"(tb), "=&v"(tc), "=&v"(td), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
          : "s"(ma), "s"(mb), "s"(mc), "s"(md), "v"(a), "v"(b), "v"(c), "v"(d));
    };
    if constexpr (R == 4) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        float t0, t1, t2, t3;
        select4(mid[0], mid[1], mid[2], mid[3], pv[0][v], pv[1][v], pv[2][v], pv[3][v], t0, t1, t2, t3);
        g[0] += t0; g[1] += t1; g[2] += t2; g[3] += t3;
      }
    } else if constexpr (R == 2 && NV % 2 == 0) {   // two rows x two consecutive values per batch; two accumulators per row
      float g2[2] = {0.f, 0.f};
#pragma unroll
      for (int v = 0; v < NV; v += 2) {
        float t0, t1, t2, t3;
        select4(mid[0], mid[1], mid[0], mid[1], pv[0][v], pv[1][v], pv[0][v + 1], pv[1][v + 1], t0, t1, t2, t3);
        g[0] += t0; g[1] += t1; g2[0] += t2; g2[1] += t3;
      }
      g[0] += g2[0]; g[1] += g2[1];
    } else {
#pragma unroll
      for (int v = 0; v < NV; ++v)      // v outer, r inner: R independent add chains side by side
#pragma unroll
        for (int r = 0; r < R; ++r) g[r] += __float_as_uint(pv[r][v]) <= mid[r] ? pv[r][v] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) g[r] = wave_sum_dpp(g[r]);
    more = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (lo[r] < hi[r]) { if (g[r] >= p.target) hi[r] = mid[r]; else lo[r] = mid[r] + 1; }   // wave-uniform selects
      more = more || lo[r] < hi[r];
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int il = il0 + r;
    if (il < p.nqb) {         // wave-uniform (a guard, not a break: the loop must unroll fully or pv[][] goes to scratch)
    const int i = p.qb0 + il;
    const unsigned vbits = lo[r];
    const float vstar = __uint_as_float(vbits);
    float base = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) base += (__float_as_uint(pv[r][v]) < vbits && v * 64 + lane < p.nb) ? pv[r][v] : 0.f;
    base = wave_sum_dpp(base);
    // ties at v*: a stable ascending sort orders them by index; the m-th tie has cumsum base + m*v*
    int m0 = 1;
    if (vstar > 0.f) { const float need = (p.target - base) / vstar; m0 = (int)ceilf(need); if (m0 < 1) m0 = 1; }
    // emit bits, 64 kv blocks per word (word c = this lane's value v = c)
    const int hw_blocks = p.Hb * p.Wb;
    const int ti = i / hw_blocks, hi_ = (i / p.Wb) % p.Hb, wi = i % p.Wb;
    int tie_seen = 0, kept = 0;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      if (c < p.nw) {
      const int j = c * 64 + lane;
      bool keep = false, tie = false;
      if (j < p.nb) {
        const unsigned vb = __float_as_uint(pv[r][c]);
        tie = vb == vbits;
        keep = vb > vbits;
        // block coordinates by multiply-high with host-made reciprocals (exact for j < 2^16, divisors <= 2^12): the three runtime
        // integer divisions per entry were ~60 instructions of this loop's ~90
        const int tj = hw_blocks == 1 ? j : (int)__umulhi((unsigned)j, p.magic_hw), rj = j - tj * hw_blocks;   // (a divisor of 1 has no 32-bit magic)
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
      }
    }
    if (lane == 0 && p.kv_nb) p.kv_nb[h * p.nqb + il] = kept;
    }
  }
}

// per (h, group of 4 query blocks): list[(h*ng+g)*nb + e] = kv_block | membership << 24 ; cnt[h*ng+g].
// Sequence parallelism: the key blocks [loc0, loc0 + locn) — the rank's own, already in place before the gather — come FIRST and
// cnt_local[h*ng+g] says how many they are: the attention runs them as a first pass while the other ranks' keys travel.
__global__ __launch_bounds__(256) void nabla_union_kernel(const unsigned long long* __restrict__ bits, int* __restrict__ list,
                                                          int* __restrict__ cnt, int* __restrict__ cnt_local, int H, int nqb, int nb,
                                                          int nw, int ng, int loc0, int locn, int G) {
  const int lane = threadIdx.x & 63;
  const int gi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gi >= H * ng) return;
  const int h = gi / ng, g = gi % ng;
  int pos = 0;
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
        const int qb = G * g + r;
        w[r] = (r < G && qb < nqb) ? bits[((size_t)h * nqb + qb) * nw + c] & take : 0ull;
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
__global__ __launch_bounds__(256) void nabla_count_kernel(const int* __restrict__ kv_nb, int rows, unsigned long long* acc) {
  unsigned long long v = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows; i += gridDim.x * 256) v += (unsigned long long)kv_nb[i];
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
  hipLaunchKernelGGL(nabla_count_kernel, dim3((rows + 4095) / 4096), dim3(256), 0, s, kv_nb, rows, acc);
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

size_t k5_nabla_workspace_bytes(int H, int nb) {
  const size_t nw = (nb + 63) / 64;
  return (size_t)2 * H * nb * 64 * 2      // qa, ka
         + (size_t)H * nb * nw * 8         // bits
         + (size_t)H * nb * 4              // kv_nb
         + (size_t)H * ((nb + 1) / 2) * nb * 4         // union lists (sized for lists per 2 rows; per 4 rows uses half)
         + (size_t)2 * H * ((nb + 1) / 2) * 4 + 256;   // counts, counts of the leading local entries (sequence parallelism)
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

// k == nullptr: the key-block means are already in the workspace (k5_launch_nabla_key_means_from_slots)
int k5_launch_nabla_select_rect(const void* q, const void* k, int ldq, int ldk, int H, int Nq, int q_block0, int N, int T, int Hb,
                                int Wb, int wT, int wH, int wW, float P, void* workspace, hipStream_t s, int local_block0, int local_blocks,
                                int group_rows) {
  if (group_rows != 2 && group_rows != 4) return K5_ERR_ARG;
  if (local_blocks < 0 || local_block0 < 0 || (local_blocks > 0 && local_block0 + local_blocks > N / 64)) return K5_ERR_ARG;
  if (H <= 0 || N <= 0 || Nq <= 0 || (N % 64) || (Nq % 64) || T * Hb * Wb * 64 != N) return K5_ERR_ARG;
  if (q_block0 < 0 || q_block0 * 64 + Nq > N) return K5_ERR_ARG;
  if ((ldq & 7) || (k && (ldk & 7))) return K5_ERR_ALIGN;
  const int nb = N / 64, nqb = Nq / 64, nw = (nb + 63) / 64, ng = (nqb + group_rows - 1) / group_rows;
  const size_t ngmax = (size_t)(nb + 1) / 2;   // region sizes (k5_nabla_workspace_bytes / _views)
  if (nb > SEL_MAXNB) return K5_ERR_UNSUPPORTED;
  char* ws = (char*)workspace;
  bf16_t* qa = (bf16_t*)ws; ws += (size_t)H * nb * 64 * 2;
  bf16_t* ka = (bf16_t*)ws; ws += (size_t)H * nb * 64 * 2;
  unsigned long long* bits = (unsigned long long*)ws; ws += (size_t)H * nb * nw * 8;
  int* kv_nb = (int*)ws; ws += (size_t)H * nb * 4;
  int* list = (int*)ws; ws += (size_t)H * ngmax * nb * 4;
  int* cnt = (int*)ws;
  int* cnt_local = cnt + (size_t)H * ngmax;
  hipLaunchKernelGGL(block_mean_kernel, dim3(nqb), dim3(256), 0, s, (const bf16_t*)q, qa, H, nqb, ldq);
  if (k) hipLaunchKernelGGL(block_mean_kernel, dim3(nb), dim3(256), 0, s, (const bf16_t*)k, ka, H, nb, ldk);
  SelP p;
  p.qa = qa; p.ka = ka; p.bits = bits; p.kv_nb = kv_nb; p.H = H; p.nb = nb; p.nw = nw; p.T = T; p.Hb = Hb; p.Wb = Wb;
  p.wT = wT; p.wH = wH; p.wW = wW; p.target = (float)(1.0 - (double)P);
  p.nqb = nqb; p.qb0 = q_block0;
  p.magic_hw = (unsigned)(((1ull << 32) + (unsigned long long)(Hb * Wb) - 1) / (unsigned long long)(Hb * Wb));
  p.magic_w = (unsigned)(((1ull << 32) + (unsigned long long)Wb - 1) / (unsigned long long)Wb);
  // values per lane NV = ceil(nb / 64) rounded up to an instantiated size; rows per wave R = 4 (2 for the largest maps: registers)
  const int nv = nw;
  auto launch = [&](auto NVC, auto RC) {
    constexpr int NV = decltype(NVC)::value, R = decltype(RC)::value;
    const int bph = (nqb + 4 * R - 1) / (4 * R);
    hipLaunchKernelGGL((nabla_select_kernel<NV, R>), dim3(H * bph), dim3(256), 0, s, p);
  };
  if (nv <= 4) launch(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
  else if (nv <= 8) launch(std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});
  else if (nv <= 16) launch(std::integral_constant<int, 16>{}, std::integral_constant<int, 4>{});
  else if (nv <= 24) launch(std::integral_constant<int, 24>{}, std::integral_constant<int, 4>{});
  // (round 2 until late: <32, 2> and <64, 1>; four / two rows per wave share the key-mean loads and take the MFMA path —
  // 3660 blocks, a 4-GPU shard of the 1280x768 10 s clip: 82.4 -> 60.2 ms of map per step)
  else if (nv <= 32) launch(std::integral_constant<int, 32>{}, std::integral_constant<int, 4>{});
  else launch(std::integral_constant<int, 64>{}, std::integral_constant<int, 2>{});
  hipLaunchKernelGGL(nabla_union_kernel, dim3((H * ng + 3) / 4), dim3(256), 0, s, bits, list, cnt, cnt_local, H, nqb, nb, nw, ng, local_block0,
                     local_blocks, group_rows);
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}

int k5_launch_nabla_select(const void* q, const void* k, int ldq, int ldk, int H, int N, int T, int Hb, int Wb, int wT, int wH,
                           int wW, float P, void* workspace, hipStream_t s) {
  return k5_launch_nabla_select_rect(q, k, ldq, ldk, H, N, 0, N, T, Hb, Wb, wT, wH, wW, P, workspace, s, 0, 0, 4);
}

// views into the workspace filled above
void k5_nabla_workspace_views(void* workspace, int H, int nb, const unsigned long long** bits, const int** kv_nb, const int** list,
                              const int** cnt, const int** cnt_local) {
  const size_t nw = (nb + 63) / 64, ng = (nb + 1) / 2;   // region sizes
  char* ws = (char*)workspace + (size_t)2 * H * nb * 64 * 2;
  if (bits) *bits = (const unsigned long long*)ws;
  ws += (size_t)H * nb * nw * 8;
  if (kv_nb) *kv_nb = (const int*)ws;
  ws += (size_t)H * nb * 4;
  if (list) *list = (const int*)ws;
  ws += (size_t)H * ng * nb * 4;
  if (cnt) *cnt = (const int*)ws;
  if (cnt_local) *cnt_local = (const int*)ws + (size_t)H * ng;
}
