// What would a hand-ordered one-wave-per-SIMD attention stream buy?  The attention tile loop's instruction mix per 64 queries x
// 64 keys is 72 MFMA 16x16x32 + 64 v_exp_f32 + 32 v_cvt_pk_bf16_f32 (no memory traffic here).
//   V=0: ONE wave per SIMD, hand order: after MFMA i (i < 64) one exp, after every second one also a cvt
//   V=1: FOUR waves per SIMD, each running half of that work in the blocked order a compiler produces
//        (16 MFMA | 24 VALU | 10 MFMA | 24 VALU | 10 MFMA), the hardware interleaves the waves
//   V=2: one wave per SIMD, blocked order (no hand interleave)
//   V=5 / V=6: V=3 / V=1 with the row sums on the VALU instead of the ones-row MFMAs: 32 MFMA + 32 exp + 16 cvt + 16 v_dot2_f32_bf16
//        (sum of the two bf16-rounded probabilities of a packed register) per wave and half tile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(ACC) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))
#define EX(X) asm volatile("v_exp_f32 %0, %0" : "+v"(X))
#define CV(O, X, Y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(O) : "v"(X), "v"(Y))
#define D2(S, PK) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(S) : "v"(PK), "v"(0x3f803f80u))
template <int V>
__global__ __launch_bounds__((V == 1 || V >= 3) ? 1024 : 256) void k(float* out, const bf16x8* in, int iters) {
  bf16x8 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
  constexpr int NACC = (V == 1 || V >= 3) ? 18 : 36;
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float fx[16];
  for (int i = 0; i < 16; ++i) fx[i] = -(float)(threadIdx.x & 7) - i;
  unsigned fp = 0;
  float rs[4] = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    if (V == 5) {   // V = 3 without the ones-row MFMAs, row sums by dot2
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        MF(acc[i % NACC]);
        EX(fx[i & 15]); if (i & 1) { CV(fp, fx[(i - 1) & 15], fx[i & 15]); D2(rs[(i >> 1) & 3], fp); }
      }
    } else if (V == 6) {   // V = 1 without the ones-row MFMAs, row sums by dot2
#pragma unroll
      for (int i = 0; i < 16; ++i) MF(acc[i % NACC]);
#pragma unroll
      for (int i = 0; i < 16; ++i) { EX(fx[i]); if (i & 1) { CV(fp, fx[i - 1], fx[i]); D2(rs[(i >> 1) & 3], fp); } }
#pragma unroll
      for (int i = 0; i < 8; ++i) MF(acc[(16 + i) % NACC]);
#pragma unroll
      for (int i = 0; i < 16; ++i) { EX(fx[i]); if (i & 1) { CV(fp, fx[i - 1], fx[i]); D2(rs[(i >> 1) & 3], fp); } }
#pragma unroll
      for (int i = 0; i < 8; ++i) MF(acc[(24 + i) % NACC]);
    } else if (V == 3) {   // four waves/SIMD, each wave hand-interleaved (36 MFMA + 32 exp + 16 cvt)
#pragma unroll
      for (int i = 0; i < 36; ++i) {
        MF(acc[i % NACC]);
        if (i < 32) { EX(fx[i & 15]); if (i & 1) CV(fp, fx[(i - 1) & 15], fx[i & 15]); }
      }
    } else if (V == 4) {   // four waves/SIMD, the dependency-limited order: S (16 MFMA) | exp(ks0) | PV(ks0) interleaved with exp(ks1) | PV(ks1)
#pragma unroll
      for (int i = 0; i < 16; ++i) MF(acc[i % NACC]);
#pragma unroll
      for (int i = 0; i < 16; ++i) { EX(fx[i]); if (i & 1) CV(fp, fx[i - 1], fx[i]); }
#pragma unroll
      for (int i = 0; i < 10; ++i) { MF(acc[(16 + i) % NACC]); EX(fx[i]); if (i < 6) EX(fx[10 + i]); if (i < 8) CV(fp, fx[i], fx[i + 1]); }
#pragma unroll
      for (int i = 0; i < 10; ++i) MF(acc[(26 + i) % NACC]);
    } else if (V == 0) {
#pragma unroll
      for (int i = 0; i < 72; ++i) {
        MF(acc[i % NACC]);
        if (i < 64) { EX(fx[i & 15]); if (i & 1) CV(fp, fx[(i - 1) & 15], fx[i & 15]); }
      }
    } else {
      const int reps = V == 1 ? 1 : 2;
#pragma unroll
      for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) MF(acc[i % NACC]);
#pragma unroll
        for (int i = 0; i < 16; ++i) { EX(fx[i]); if (i & 1) CV(fp, fx[i - 1], fx[i]); }
#pragma unroll
        for (int i = 0; i < 10; ++i) MF(acc[(16 + i) % NACC]);
#pragma unroll
        for (int i = 0; i < 16; ++i) { EX(fx[i]); if (i & 1) CV(fp, fx[i - 1], fx[i]); }
#pragma unroll
        for (int i = 0; i < 10; ++i) MF(acc[(26 + i) % NACC]);
      }
    }
  }
  float s = fp + rs[0] + rs[1] + rs[2] + rs[3];
  for (int i = 0; i < 16; ++i) s += fx[i];
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V> void run(const char* name, float* out, bf16x8* in) {
  const int iters = 3000, threads = (V == 1 || V >= 3) ? 1024 : 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<V><<<256, threads>>>(out, in, 10);
  hipEventRecord(e0);
  k<V><<<256, threads>>>(out, in, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // work per SIMD and iteration: V=0/2: one wave x 72 MFMA; V=1: four waves x 36 MFMA = 144 MFMA
  const double mf_per_simd = (double)iters * (V >= 5 ? 128 : (V == 1 || V >= 3) ? 144 : 72);
  // attention work per SIMD and iteration: two 64 x 64 tiles of 64 useful MFMAs (V = 0 / 2: one)
  const double attn_fl = (double)iters * ((V == 1 || V >= 3) ? 128 : 64) * 16384 * 256.0 * 4;
  printf("%-62s %.3f ms   %.2f ns per MFMA slot  (%.0f TFLOP/s of MFMA work, %.0f of attention work)\n", name, ms, ms * 1e6 / mf_per_simd,
         256.0 * 4 * mf_per_simd * 16384 / ms / 1e9, attn_fl / ms / 1e9);
}
int main() {
  float* out; bf16x8* in; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&in, 128 * 16);
  unsigned short h[128 * 8];
  for (int i = 0; i < 128 * 8; ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<0>("one wave/SIMD, hand-interleaved (72 MFMA + 64 exp + 32 cvt)", out, in);
  run<1>("four waves/SIMD, blocked order per wave", out, in);
  run<2>("one wave/SIMD, blocked order", out, in);
  run<3>("four waves/SIMD, each hand-interleaved", out, in);
  run<4>("four waves/SIMD, PV(ks0) interleaved with exp(ks1) only", out, in);
  run<5>("four waves/SIMD, hand-interleaved, row sums by v_dot2 (no ones MFMA)", out, in);
  run<6>("four waves/SIMD, blocked order, row sums by v_dot2 (no ones MFMA)", out, in);
  run<3>("four waves/SIMD, each hand-interleaved (again)", out, in);
  run<1>("four waves/SIMD, blocked order per wave (again)", out, in);
  return 0;
}
