// How fast can ONE wave per SIMD issue independent MFMA 16x16x32 bf16?  (the 4-wave GEMM's question)
// variants: V=0 acc in AGPR, A varies / B fixed per 8;  V=1 acc in AGPR, A fixed / B varies;  V=2 acc in VGPR (builtin);
//           V=3 like 0 with an s_nop 1 between MFMAs;  V=4 like 0 but 2 waves per SIMD (512 threads)
//           V=6/7/8/9: like 0 with 1/2/3/4 independent VALU fillers (v_exp_f32, every third a v_cvt_pk_bf16_f32) after each MFMA:
//           does the same wave's VALU work hide in the 16-cycle MFMA shadow?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int V>
__global__ __launch_bounds__(V == 4 ? 512 : 256) void k(float* out, const bf16x8* in, int iters) {
  extern __shared__ char sm[];
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[threadIdx.x + 64 * i]; b[i] = in[threadIdx.x + 64 * (8 + i)]; }
  f32x4 acc[8][8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  float fx[8];
  for (int i = 0; i < 8; ++i) fx[i] = -(float)(threadIdx.x & 7) - i;
  unsigned fp = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < (V == 4 ? 32 : 64); ++q) {
      if (V == 0 || V == 4) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[q & 7][q >> 3]) : "v"(a[q & 7]), "v"(b[q >> 3]));
      if (V == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[q & 7][q >> 3]) : "v"(a[q >> 3]), "v"(b[q & 7]));
      if (V == 2) acc[q & 7][q >> 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[q & 7], b[q >> 3], acc[q & 7][q >> 3], 0, 0, 0);
      if (V >= 6) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[q & 7][q >> 3]) : "v"(a[q & 7]), "v"(b[q >> 3]));
#pragma unroll
        for (int f = 0; f < V - 5; ++f) {
          const int r = (q * (V - 5) + f) & 7;
          if (((q * (V - 5) + f) % 3) == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(fp) : "v"(fx[r]), "v"(fx[(r + 1) & 7]));
          else asm volatile("v_exp_f32 %0, %0" : "+v"(fx[r]));
        }
      }
      if (V == 3) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 1" : "+a"(acc[q & 7][q >> 3]) : "v"(a[q & 7]), "v"(b[q >> 3]));
    }
  }
  float s = fp;
  for (int i = 0; i < 8; ++i) s += fx[i];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V> void run(const char* name, int threads, float* out, bf16x8* in, bool rnd) {
  int iters = 4000;
  hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<V><<<256, threads, 131072>>>(out, in, 10);
  hipEventRecord(e0);
  k<V><<<256, threads, 131072>>>(out, in, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = 256.0 * (threads / 64) * iters * (V == 4 ? 32 : 64) * 16384.0;
  printf("%-44s %s: %.3f ms  %.1f TFLOP/s\n", name, rnd ? "random" : "zeros ", ms, fl / ms / 1e9);
}
int main() {
  float* out; bf16x8* in; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&in, 1024 * 16 * 16);
  for (int rnd = 0; rnd < 2; ++rnd) {
    unsigned short* h = (unsigned short*)malloc(1024 * 16 * 16);
    for (int i = 0; i < 1024 * 16 * 8; ++i) h[i] = rnd ? (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15)) : 0;
    hipMemcpy(in, h, 1024 * 16 * 16, hipMemcpyHostToDevice);
    run<0>("1 wave/SIMD agpr, A varies", 256, out, in, rnd);
    run<1>("1 wave/SIMD agpr, B varies", 256, out, in, rnd);
    run<2>("1 wave/SIMD builtin", 256, out, in, rnd);
    run<3>("1 wave/SIMD agpr + s_nop 1", 256, out, in, rnd);
    run<4>("2 waves/SIMD agpr", 512, out, in, rnd);
    run<6>("1 wave/SIMD + 1 VALU filler per MFMA", 256, out, in, rnd);
    run<7>("1 wave/SIMD + 2 VALU fillers per MFMA", 256, out, in, rnd);
    run<8>("1 wave/SIMD + 3 VALU fillers per MFMA", 256, out, in, rnd);
    run<9>("1 wave/SIMD + 4 VALU fillers per MFMA", 256, out, in, rnd);
  }
  return 0;
}
