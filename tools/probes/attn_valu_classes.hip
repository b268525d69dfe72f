// Issue-slot ledger of the dense attention tile loop: what does each non-MFMA instruction class of
// attn_fwd_kernel<true, false, true, true, false, 4> cost beside the matrix work?  Per wave and 64-key tile the loop is 36 MFMA 16x16x32 +
// 32 v_exp_f32 + 16 v_cvt_pk_bf16_f32 (+ 16 ds_read_b128, 2 LDS-DMA, waits: not in this probe — no memory traffic here).  Four waves per
// SIMD, each in the blocked order the compiler emits (16 MFMA | 16 exp + 8 cvt | 10 MFMA | 16 exp + 8 cvt | 10 MFMA), interleaved by the
// hardware.  Variants remove or replace ONE class at a time:
//   0 the loop's mix              1 no v_exp_f32             2 no v_cvt_pk              3 MFMA only
//   4 exp2 as full-rate VALU: v_fract + v_sub + 2 v_fma (quadratic in the fraction: 2^-9 relative, bf16-grade) + v_ldexp   (5 ops per value)
//   5 the pack as v_perm_b32 (truncation instead of round-to-nearest-even; same count as v_cvt_pk)
//   6 half of the exponentials (what a 2-values-per-instruction transcendental would cost)
//   hipcc --offload-arch=gfx950 -O3 -o attn_valu_classes.bin attn_valu_classes.hip && ./attn_valu_classes.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(ACC) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))
#define EX(X) asm volatile("v_exp_f32 %0, %0" : "+v"(X))
#define CV(O, X, Y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(O) : "v"(X), "v"(Y))
#define PM(O, X, Y) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(O) : "v"(X), "v"(Y), "v"(0x07060302u))
// 2^x, x <= 0: f = fract(x), i = x - f, p = 1 + f (c1 + f c2), result = ldexp(p, i)
#define EXPOLY(X) do { float f_, i_, p_; asm volatile("v_fract_f32 %0, %1" : "=v"(f_) : "v"(X)); asm volatile("v_sub_f32 %0, %1, %2" : "=v"(i_) : "v"(X), "v"(f_)); \
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(p_) : "v"(f_), "v"(0.3371894346f), "v"(0.6575959662f)); asm volatile("v_fma_f32 %0, %1, %2, 1.0" : "=v"(p_) : "v"(p_), "v"(f_)); \
    int ii_; asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(ii_) : "v"(i_)); asm volatile("v_ldexp_f32 %0, %1, %2" : "=v"(X) : "v"(p_), "v"(ii_)); } while (0)
template <int V>
__global__ __launch_bounds__(1024) void k(float* out, const bf16x8* in, int iters) {
  bf16x8 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
  f32x4 acc[18];
  for (int i = 0; i < 18; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float fx[16];
  for (int i = 0; i < 16; ++i) fx[i] = -(float)(threadIdx.x & 7) - i;
  unsigned fp = 0;
  auto valu = [&]() {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (V == 0 || V == 2 || V == 5) EX(fx[i]);
      if (V == 6 && (i & 1)) EX(fx[i]);
      if (V == 4) { EXPOLY(fx[i]); fx[i] = fx[i] * 0.f - (float)i; }
      if ((V == 0 || V == 1 || V == 4 || V == 6) && (i & 1)) CV(fp, fx[i - 1], fx[i]);
      if (V == 5 && (i & 1)) PM(fp, fx[i - 1], fx[i]);
    }
  };
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) MF(acc[i % 18]);
    valu();
#pragma unroll
    for (int i = 0; i < 10; ++i) MF(acc[(16 + i) % 18]);
    valu();
#pragma unroll
    for (int i = 0; i < 10; ++i) MF(acc[(26 + i) % 18]);
  }
  float s = fp;
  for (int i = 0; i < 16; ++i) s += fx[i];
  for (int i = 0; i < 18; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V> double run(const char* name, float* out, bf16x8* in, double base) {
  const int iters = 3000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<V><<<256, 1024>>>(out, in, 10);
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    hipEventRecord(e0);
    k<V><<<256, 1024>>>(out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  // per SIMD and iteration: 4 waves x 36 MFMA = 2 tiles of 64 queries x 64 keys (64 useful MFMAs each); ns per wave-tile = time / iters (each wave does one tile's worth per iteration)
  const double ns_tile = best * 1e6 / iters / 4.0;     // per wave-tile on a SIMD shared by 4 waves: SIMD time per wave-tile
  printf("%-74s %8.3f ms  %7.1f ns of SIMD time per wave-tile  (%5.0f TFLOP/s of attention work)%s\n", name, best, ns_tile,
         (double)iters * 128 * 16384 * 256.0 * 4 / best / 1e9, base > 0 ? "" : "");
  if (base > 0) printf("%-74s           %+7.1f ns against the loop's mix\n", "", ns_tile - base);
  return ns_tile;
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float* out; bf16x8* in; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&in, 128 * 16);
  unsigned short h[128 * 8];
  for (int i = 0; i < 128 * 8; ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const double b = run<0>("0  the loop's mix: 36 MFMA + 32 v_exp_f32 + 16 v_cvt_pk per wave-tile", out, in, 0);
  run<3>("3  MFMA only", out, in, b);
  run<1>("1  without the 32 v_exp_f32", out, in, b);
  run<2>("2  without the 16 v_cvt_pk_bf16_f32", out, in, b);
  run<6>("6  16 instead of 32 v_exp_f32", out, in, b);
  run<4>("4  exp2 as 6 full-rate VALU ops (fract, sub, 2 fma, cvt_i32, ldexp) instead of v_exp_f32", out, in, b);
  run<5>("5  v_perm_b32 (truncating pack) instead of v_cvt_pk_bf16_f32", out, in, b);
  run<0>("0  the loop's mix (again)", out, in, b);
  return 0;
}
