// Microbenchmark of the attention kernel's building blocks on gfx950 (1 workgroup of 8 waves per CU).
// Each variant runs ITER "intervals"; reports cycles per interval per SIMD assuming 2.1 GHz, and the
// MFMA utilisation = 16 MFMA * 32 cyc / interval.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

template <int V>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  const int tid = threadIdx.x, lane = tid & 63;
  const bool grpB = __builtin_amdgcn_readfirstlane(tid) >= 256;
  for (int i = tid; i < 8192; i += 512) ((float*)lds)[i] = (float)i * 1e-3f;
  __syncthreads();
  bf16x8 a[16], b[4];
  f32x16 c0 = {0}, c1 = {0};
  for (int i = 0; i < 4; ++i) b[i] = *(bf16x8*)(lds + ((lane * 16 + i * 1024) & 32767));
  const int l31 = lane & 31, hi = lane >> 5;
  auto swz = [](int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); };
  auto mhalf = [&]() {
    if (V & 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = *(const bf16x8*)(lds + (i >> 3) * 16384 + swz(32 * ((i >> 2) & 1) + l31, 2 * (i & 3) + hi));
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 16; i += 2) { c0 = MFMA(a[i], b[(i >> 1) & 3], c0); c1 = MFMA(a[i + 1], b[(i >> 1) & 3], c1); }
    __builtin_amdgcn_s_setprio(0);
  };
  auto vhalf = [&]() {
    if (V & 4) {  // ~150 VALU ops incl. 32 exp
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float x = c0[r], y = c1[r];
        x = __builtin_amdgcn_exp2f(fmaf(x, 0.18f, -1.f)); y = __builtin_amdgcn_exp2f(fmaf(y, 0.18f, -1.f));
        c0[r] = x + y * 0.5f; c1[r] = y + x * 0.25f + fmaxf(x, y);
      }
    }
  };
  for (int i = 0; i < 16; ++i) a[i] = *(bf16x8*)(lds + ((lane * 16 + i * 512) & 32767));
  if (V & 8) {  // no barriers, every wave: M then V back to back (free running)
    for (int it = 0; it < iters; ++it) { mhalf(); vhalf(); }
  } else if (V & 16) {  // lockstep: all waves M, barrier, all waves V, barrier
    for (int it = 0; it < iters; ++it) { mhalf(); __syncthreads(); vhalf(); __syncthreads(); }
  } else {  // ping-pong
    if (!grpB) { for (int it = 0; it < iters; ++it) { mhalf(); __syncthreads(); vhalf(); __syncthreads(); } __syncthreads(); }
    else { __syncthreads(); for (int it = 0; it < iters; ++it) { mhalf(); __syncthreads(); vhalf(); __syncthreads(); } }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
  out[blockIdx.x * 512 + tid] = s;
}

template <int V> void run(const char* name, float* d, int blocks) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(512), 0, 0, d, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(512), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per = ms * 1e-3 / iters;            // seconds per iteration (= 2 intervals, 2 wave-tiles of MFMA per SIMD)
  const double tf = (double)blocks * 8 * 16 * 32768.0 / per / 1e12;
  printf("%-44s %8.3f ms  %7.0f ns/iter  %7.1f TFLOP/s-equiv\n", name, ms, per * 1e9, tf);
}
int main() {
  float* d; hipMalloc(&d, 2048 * 512 * 4);
  for (int blocks : {256, 512}) {
    printf("blocks=%d\n", blocks);
    run<8>("free-running  MFMA only", d, blocks);
    run<8 | 1>("free-running  MFMA + 16 ds_read_b128", d, blocks);
    run<8 | 1 | 4>("free-running  MFMA + ds_read + VALU", d, blocks);
    run<16 | 1 | 4>("lockstep      MFMA + ds_read + VALU", d, blocks);
    run<0>("ping-pong     MFMA only", d, blocks);
    run<1>("ping-pong     MFMA + ds_read", d, blocks);
    run<1 | 4>("ping-pong     MFMA + ds_read + VALU", d, blocks);
  }
  return 0;
}
