// Peak issue rate of the dense bf16 MFMA shapes on gfx950 at 2 and 4 waves per SIMD (8 / 16 independent accumulators per
// wave, no memory traffic), constant vs pseudo-random operands (clock / power effects).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int RANDOM>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int tid = threadIdx.x;
  bf16x8 a[4], b[2];
  unsigned seed = tid * 2654435761u + blockIdx.x;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { seed = seed * 1664525u + 1013904223u; a[i][e] = (__bf16)(RANDOM ? ((int)(seed >> 20) - 2048) * (1.f / 2048) : 0.5f); }
  for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) { seed = seed * 1664525u + 1013904223u; b[i][e] = (__bf16)(RANDOM ? ((int)(seed >> 20) - 2048) * (1.f / 2048) : 0.25f); }
  float s = 0;
  if (SHAPE == 32) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * j + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[4 * j + i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * j + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j & 1], acc[4 * j + i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  out[blockIdx.x * 512 + tid] = s;
}

template <int SHAPE, int RANDOM> void run(const char* name, float* d, int blocks_per_cu) {
  const int iters = 20000, blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE, RANDOM>), dim3(blocks), dim3(512), 0, 0, d, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<SHAPE, RANDOM>), dim3(blocks), dim3(512), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop_per_iter_wave = SHAPE == 32 ? 8 * 32768.0 : 16 * 16384.0;
  printf("%-44s %8.1f TFLOP/s\n", name, blocks * 8.0 * flop_per_iter_wave * iters / (ms * 1e-3) / 1e12);
}

int main() {
  float* d; hipMalloc(&d, 512 * 512 * 4);
  run<32, 0>("32x32x16 const operands, 2 waves/SIMD", d, 1);
  run<32, 1>("32x32x16 random operands, 2 waves/SIMD", d, 1);
  run<32, 1>("32x32x16 random operands, 4 waves/SIMD", d, 2);
  run<16, 0>("16x16x32 const operands, 2 waves/SIMD", d, 1);
  run<16, 1>("16x16x32 random operands, 2 waves/SIMD", d, 1);
  run<16, 1>("16x16x32 random operands, 4 waves/SIMD", d, 2);
  return 0;
}
