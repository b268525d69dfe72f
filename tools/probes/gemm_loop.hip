// Microbenchmark of the 256x256 GEMM kernel's inner structure on gfx950: 1 workgroup of 8 waves per CU (2 per SIMD),
// per "phase" 8 independent MFMA 32x32x16 (+ optionally 6 ds_read_b128 feeding them, barriers, two-group ping-pong).
// Prints cycles per phase per wave and the matrix-pipe utilisation (8 MFMA x 32 cycles x 2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// V bit0: ds_reads each phase; bit1: barrier after reads + after MFMAs (lockstep); bit2: ping-pong (needs bit1);
// bit3: lgkmcnt(0) before the first barrier; bit4: 16 MFMAs per phase (two k-steps, 12 reads)
template <int V>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 32768; i += 512) ((float*)lds)[i] = (float)(i & 1023) * 1e-3f;
  __syncthreads();
  const int l31 = lane & 31, hi = lane >> 5, sw = (l31 >> 2) & 3;
  const char* wb = lds + (128 * (wave & 1) + l31) * 64 + ((hi ^ sw) << 4);
  const char* xb = lds + 65536 + (64 * (wave >> 1) + l31) * 64 + ((hi ^ sw) << 4);
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 wf[8], xf[4];
  for (int i = 0; i < 8; ++i) wf[i] = *(const bf16x8*)(wb + i * 2048);
  for (int j = 0; j < 4; ++j) xf[j] = *(const bf16x8*)(xb + j * 2048);
  constexpr int NK = (V & 16) ? 2 : 1;
  auto phase = [&](int off) {
    __builtin_amdgcn_sched_barrier(0);
    if (V & 1) {
#pragma unroll
      for (int j = 0; j < 2 * NK; ++j) xf[j] = *(const bf16x8*)(xb + off + j * 2048);
#pragma unroll
      for (int i = 0; i < 4 * NK; ++i) wf[i] = *(const bf16x8*)(wb + off + i * 2048);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (V & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (V & 2) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < NK; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * j + i] = MFMA(wf[4 * kk + i], xf[2 * kk + j], acc[4 * j + i]);
    __builtin_amdgcn_sched_barrier(0);
    if (V & 2) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  if ((V & 4) && wave >= 4) __builtin_amdgcn_s_barrier();
  for (int it = 0; it < iters; it += 2) { phase(0); phase(16384); }
  if ((V & 4) && wave < 4) __builtin_amdgcn_s_barrier();
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 512 + tid] = s;
}

template <int V> void run(const char* name, float* d) {
  const int iters = 4000, blocks = 256;
  hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(512), 131072, 0, d, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(512), 131072, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int nm = (V & 16) ? 16 : 8;
  const double ns = ms * 1e6 / iters;
  const double tf = 256.0 * 8 * nm * 32768.0 * iters / (ms * 1e-3) / 1e12;
  printf("%-46s %7.1f ns/phase  %7.1f TFLOP/s\n", name, ns, tf);
}

int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  run<0>("pure MFMA (8 indep. acc)", d);
  run<1>("+ 6 ds_read_b128 / phase, free running", d);
  run<9>("+ reads, lgkmcnt(0) before MFMAs", d);
  run<3>("+ reads + 2 barriers, lockstep", d);
  run<7>("+ reads + 2 barriers, ping-pong", d);
  run<15>("+ reads + lgkm(0) + 2 barriers, ping-pong", d);
  run<16>("pure MFMA, 16 per phase", d);
  run<17>("16 MFMA + 12 reads, free running", d);
  run<23>("16 MFMA + 12 reads + 2 barriers, ping-pong", d);
  run<31>("16 MFMA + 12 reads + lgkm + barriers, ping-pong", d);
  return 0;
}
