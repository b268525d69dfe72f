// Instruction-mix probe for a ONE-wave-per-SIMD attention tile loop on MFMA 32x32x16 (no global traffic): per wave and per
// (64 queries x 64 keys, d = 64) tile  32 MFMA + 64 v_exp_f32 + 32 v_cvt_pk_bf16_f32 + [64 v_add_f32 row sums] + 16 ds_read_b128.
//   V=0  MFMA only (the floor)                              V=1  + exp + cvt, hand-interleaved (2 exp + 1 cvt per gap)
//   V=2  V=1 + 2 v_add per gap (row sums on the VALU)       V=3  V=2 + one ds_read_b128 every second gap
//   V=4  V=3 without the adds but 8 more MFMAs (row sums on the matrix core)
//   V=5  V=3 in the blocked order a compiler emits (16 MFMA | VALU | 16 MFMA | VALU)
//   V=6  TWO waves per SIMD, each half of V=3's work (32 queries), hand-interleaved
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MF(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(a), "v"(b))
#define EX(X) asm volatile("v_exp_f32 %0, %0" : "+v"(X))
#define CV(O, X, Y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(O) : "v"(X), "v"(Y))
#define AD(S, X) asm volatile("v_add_f32 %0, %0, %1" : "+v"(S) : "v"(X))
#define RD(D, A, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(D) : "v"(A), "n"(OFF))
template <int V>
__global__ __launch_bounds__(V == 6 ? 512 : 256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k(float* out, const bf16x8* in, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) ((unsigned*)lds)[i] = i * 2654435761u;
  __syncthreads();
  bf16x8 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
  constexpr int NACC = V == 6 ? 4 : 8;
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float fx[32], rsv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 32; ++i) fx[i] = -(float)(threadIdx.x & 7) - i;
  unsigned fp = 0;
  u32x4 frag[4];
  const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x & 63) * 16;
  constexpr int NM = V == 4 ? 40 : (V == 6 ? 16 : 32);     // MFMAs per tile and wave
  for (int it = 0; it < iters; ++it) {
    if (V == 5) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 16; ++i) MF(acc[i % NACC]);
#pragma unroll
        for (int i = 0; i < 32; ++i) { EX(fx[i]); AD(rsv[i & 7], fx[i]); if (i & 1) CV(fp, fx[i - 1], fx[i]); }
#pragma unroll
        for (int i = 0; i < 8; ++i) RD(frag[i & 3], laddr, (i & 15) * 1024);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        MF(acc[i % NACC]);
        if (V >= 1 && i < (V == 6 ? 16 : 32)) {
          // software-pipelined: the exps of this gap are consumed (cvt, add) in the NEXT gap -> no dependent VALU back to back
          EX(fx[(2 * i) & 31]); EX(fx[(2 * i + 1) & 31]);
          CV(fp, fx[(2 * i - 2) & 31], fx[(2 * i - 1) & 31]);
          if (V == 2 || V == 3 || V == 6) { AD(rsv[(2 * i) & 7], fx[(2 * i - 2) & 31]); AD(rsv[(2 * i + 1) & 7], fx[(2 * i - 1) & 31]); }
          if (V >= 3 && (V == 6 || (i & 1))) RD(frag[i & 3], laddr, (i & 15) * 1024);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = fp; for (int i = 0; i < 8; ++i) s += rsv[i];
  for (int i = 0; i < 32; ++i) s += fx[i];
  for (int i = 0; i < 4; ++i) s += (float)frag[i][0];
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V> void run(const char* name, float* out, bf16x8* in) {
  const int iters = 4000, threads = V == 6 ? 512 : 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<V><<<256, threads>>>(out, in, 10);
  hipEventRecord(e0);
  k<V><<<256, threads>>>(out, in, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // useful work per SIMD and iteration: one (64 q x 64 k, d 64) tile = 4 * 64 * 64 * 64 flop (V=6: two waves x half a tile)
  const double flop = 256.0 * 4 * iters * 4.0 * 64 * 64 * 64;
  printf("%-72s %.3f ms   %.1f ns per tile  (%.0f TFLOP/s of attention work)\n", name, ms, ms * 1e6 / iters, flop / ms / 1e9);
}
int main() {
  float* out; bf16x8* in; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&in, 128 * 16);
  unsigned short h[128 * 8];
  for (int i = 0; i < 128 * 8; ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<0>("V0 one wave/SIMD, 32 MFMA 32x32x16 only", out, in);
  run<1>("V1 + 64 exp + 32 cvt, hand-interleaved", out, in);
  run<2>("V2 + 64 v_add row sums", out, in);
  run<3>("V3 + 16 ds_read_b128", out, in);
  run<4>("V4 row sums on the matrix core instead (40 MFMA, no adds)", out, in);
  run<5>("V5 as V3, blocked order", out, in);
  run<6>("V6 two waves/SIMD, half the tile each, hand-interleaved", out, in);
  return 0;
}
