// k5_common.h — shared device helpers for the gfx950 (MI355X / CDNA4) kernels of libk5.
// wave = 64 lanes, MFMA 32x32x16 bf16, LDS 160 KiB/CU.  No portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define K5_DEV __device__ __forceinline__
#define K5_HD __host__ __device__ __forceinline__

// ---- bf16 <-> f32 (round-to-nearest-even, hardware v_cvt_pk_bf16_f32 on gfx950) ----
K5_DEV float bf2f(bf16_t v) { return (float)v; }
K5_DEV bf16_t f2bf(float v) { return (bf16_t)v; }
K5_DEV float bfbits2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
K5_DEV uint32_t pack_bf16x2(float lo, float hi) {
  f32x2 t = {lo, hi};
  bf16x2 r = __builtin_convertvector(t, bf16x2);
  return __builtin_bit_cast(uint32_t, r);
}
K5_DEV float bf_round(float v) { return (float)(bf16_t)v; }

// MFMA 32x32x16 bf16: D[i][j] += sum_k A[i][k] B[k][j]
//   A operand: lane l holds A[i = l&31][k = 8*(l>>5) + 0..7]
//   B operand: lane l holds B[k = 8*(l>>5) + 0..7][j = l&31]
//   C/D:       lane l, reg r holds D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31]
K5_DEV f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
K5_DEV int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// LDS tile of [rows][64] bf16 (128 B rows, eight 16-B chunks per row).  Chunk swizzle
// c' = c ^ ((row >> 1) & 7): two rows share one 256-B bank row, so (row&1, c') covers all
// sixteen 16-B slots for any 16 rows distinct mod 16 -> ds_read_b128 fragment reads of one
// k-chunk from 16 different rows are conflict free (guide §2 / T2).
K5_DEV int lds_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

K5_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
K5_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact-erf GELU (nn.GELU default, approximate='none'), fp32 math
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, below the bf16 rounding every caller applies to the result):
// branch-free, 14 VALU ops against ~40 for the device library's erff.
// Round 6 measured the one-transcendental form 7.1.28, 1 - 1 / (1 + a1 x + ... + a6 x^6)^16 (-DK5_GELU_AS28; as accurate: over all bf16 inputs the
// rounded GELU differs from the exactly rounded one in 141 of 33 410 values against 117, profiles/r06_gelu_erf_forms.log): FF1 + GELU 1036-1052 us
// against 1038-1043 at 47 616 rows, alternating on one box (profiles/r06_gelu_ab.log) — the fused GELU's cost is not its transcendentals.  7.1.26
// stays (the bits of rounds 1-5).
#ifndef K5_GELU_AS28
K5_DEV float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * ax * ax);
  return copysignf(fmaf(-p * t, e, 1.0f), x);
}
#else
K5_DEV float erf_as(float x) {
  const float ax = fabsf(x);
  float p = fmaf(0.0000430638f, ax, 0.0002765672f);
  p = fmaf(p, ax, 0.0001520143f);
  p = fmaf(p, ax, 0.0092705272f);
  p = fmaf(p, ax, 0.0422820123f);
  p = fmaf(p, ax, 0.0705230784f);
  p = fmaf(p, ax, 1.0f);
  p = p * p; p = p * p; p = p * p; p = p * p;          // ^16; overflows to +inf beyond |x| ~ 13, where 1 / inf = 0 is the right answer
  return copysignf(1.0f - __builtin_amdgcn_rcpf(p), x);
}
#endif
K5_DEV float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
// two values at once on the packed fp32 pipe (v_pk_mul / v_pk_fma_f32; rcp and exp2 stay scalar): the SAME operations in the same order as
// gelu_erf, so the bits are the same — 21 VALU instructions per pair instead of 36.  For epilogues with no MFMA beside them (beside MFMAs
// packed math is an anti-lever, which is why the build runs with -fno-slp-vectorize).
typedef float k5_f32x2 __attribute__((ext_vector_type(2)));
K5_DEV uint32_t pack_fp8x4(float a, float b, float c, float d) {   // saturating e4m3 conversion of four values
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (uint32_t)r;
}

// NABLA lists shared by TWO 64-query rows (128-query attention workgroups): which rows group g holds.  S = 0: the adjacent rows 2g, 2g + 1.
// S > 0 (round 4, "frame pairing"): S = blocks per latent frame of the fractal token order — rows b and b + S are the SAME 8 x 8 spatial tile
// in adjacent frames, whose sliding-tile windows overlap in 10 of 11 frames (adjacent tiles of one frame: in 2 of 3 columns), so the union
// list is tighter.  Whole chunks of 2S rows pair i with i + S; a last, shorter chunk falls back to adjacent pairs; ceil(n / 2) groups either way.
K5_HD int k5_pair_row(int g, int r, int S, int n) {
  if (S <= 0) return 2 * g + r;
  const int c = g / S, base = 2 * S * c;
  if (base + 2 * S <= n) return base + (g - c * S) + r * S;
  return base + 2 * (g - c * S) + r;
}

K5_DEV void gelu_erf_x2(float& a, float& b) {
  const k5_f32x2 x0 = {a, b};
  const k5_f32x2 x = x0 * k5_f32x2{0.70710678118654752440f, 0.70710678118654752440f};
  const k5_f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
#ifndef K5_GELU_AS28
  const k5_f32x2 d = __builtin_elementwise_fma(k5_f32x2{0.3275911f, 0.3275911f}, ax, k5_f32x2{1.0f, 1.0f});
  const k5_f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  k5_f32x2 p = __builtin_elementwise_fma(k5_f32x2{1.061405429f, 1.061405429f}, t, k5_f32x2{-1.453152027f, -1.453152027f});
  p = __builtin_elementwise_fma(p, t, k5_f32x2{1.421413741f, 1.421413741f});
  p = __builtin_elementwise_fma(p, t, k5_f32x2{-0.284496736f, -0.284496736f});
  p = __builtin_elementwise_fma(p, t, k5_f32x2{0.254829592f, 0.254829592f});
  const k5_f32x2 a2 = (k5_f32x2{-1.44269504088896340736f, -1.44269504088896340736f} * ax) * ax;
  const k5_f32x2 e = {__builtin_amdgcn_exp2f(a2[0]), __builtin_amdgcn_exp2f(a2[1])};
  const k5_f32x2 r = __builtin_elementwise_fma(-(p * t), e, k5_f32x2{1.0f, 1.0f});
#else
  k5_f32x2 p = __builtin_elementwise_fma(k5_f32x2{0.0000430638f, 0.0000430638f}, ax, k5_f32x2{0.0002765672f, 0.0002765672f});
  p = __builtin_elementwise_fma(p, ax, k5_f32x2{0.0001520143f, 0.0001520143f});
  p = __builtin_elementwise_fma(p, ax, k5_f32x2{0.0092705272f, 0.0092705272f});
  p = __builtin_elementwise_fma(p, ax, k5_f32x2{0.0422820123f, 0.0422820123f});
  p = __builtin_elementwise_fma(p, ax, k5_f32x2{0.0705230784f, 0.0705230784f});
  p = __builtin_elementwise_fma(p, ax, k5_f32x2{1.0f, 1.0f});
  p = p * p; p = p * p; p = p * p; p = p * p;
  const k5_f32x2 r = k5_f32x2{1.0f, 1.0f} - k5_f32x2{__builtin_amdgcn_rcpf(p[0]), __builtin_amdgcn_rcpf(p[1])};
#endif
  const k5_f32x2 er = {copysignf(r[0], x[0]), copysignf(r[1], x[1])};
  const k5_f32x2 o = (k5_f32x2{0.5f, 0.5f} * x0) * (k5_f32x2{1.0f, 1.0f} + er);
  a = o[0]; b = o[1];
}
K5_DEV float silu(float x) { return x / (1.0f + __expf(-x)); }

// bijective XCD-aware block remap (guide T1): physical block b runs on XCD b%8; give each XCD a
// contiguous range of logical tiles so that neighbours share operand panels in that XCD's L2.
K5_DEV int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
