// Does VALU work of one wave overlap the MFMA stream of its SIMD partner on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
// MODE bit0: waves 0-3 run MFMA loop; bit1: waves 4-7 run VALU loop; bit2: VALU uses exp; bit3: setprio(1) on MFMA waves
// MODE bit4: single role: every wave interleaves NV VALU ops after each MFMA (same wave)
template <int MODE, int NV>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int tid = threadIdx.x;
  const bool grpB = __builtin_amdgcn_readfirstlane(tid) >= 256;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(tid * 0.001f + i); b[i] = (__bf16)(i * 0.5f - tid * 0.002f); }
  f32x16 c0 = {0}, c1 = {0};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = tid * 0.01f + i;
  auto valu8 = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (MODE & 4) ? __builtin_amdgcn_exp2f(v[i]) : fmaf(v[i], 0.999f, 0.001f);
  };
  if (MODE & 16) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        c0 = MFMA(a, b, c0);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i & 7] = (MODE & 4) ? __builtin_amdgcn_exp2f(v[i & 7]) : fmaf(v[i & 7], 0.999f, 0.001f);
        c1 = MFMA(a, b, c1);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[(i + 4) & 7] = (MODE & 4) ? __builtin_amdgcn_exp2f(v[(i + 4) & 7]) : fmaf(v[(i + 4) & 7], 0.999f, 0.001f);
      }
    }
  } else if (!grpB) {
    if (MODE & 1) {
      if (MODE & 8) __builtin_amdgcn_s_setprio(1);
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) { c0 = MFMA(a, b, c0); c1 = MFMA(a, b, c1); }
      }
    }
  } else {
    if (MODE & 2) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) valu8();   // 128 VALU per iteration
      }
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + tid] = s;
}
template <int MODE, int NV> void run(const char* name, float* d) {
  const int iters = 4000, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(512), 0, 0, d, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(512), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-64s %8.3f ms  %7.1f ns/iter\n", name, ms, ms * 1e6 / iters);
}
int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  run<1, 0>("A: 16 MFMA/iter, partner idle", d);
  run<2, 0>("B: 128 v_fma/iter, partner idle", d);
  run<2 | 4, 0>("B: 128 v_exp/iter, partner idle", d);
  run<3, 0>("A: 16 MFMA  ||  B: 128 v_fma   (separate waves, same SIMD)", d);
  run<3 | 8, 0>("A: 16 MFMA (setprio 1)  ||  B: 128 v_fma", d);
  run<3 | 4, 0>("A: 16 MFMA  ||  B: 128 v_exp", d);
  run<16, 0>("all waves: 16 MFMA, 0 VALU between", d);
  run<16, 2>("all waves: 16 MFMA, 2 v_fma after each MFMA (same wave)", d);
  run<16, 4>("all waves: 16 MFMA, 4 v_fma after each MFMA (same wave)", d);
  run<16, 8>("all waves: 16 MFMA, 8 v_fma after each MFMA (same wave)", d);
  run<16 | 4, 4>("all waves: 16 MFMA, 4 v_exp after each MFMA (same wave)", d);
  return 0;
}
