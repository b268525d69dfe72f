// Layout probe for v_mfma_f32_4x4x4_16b_bf16 (16 blocks of 4x4x4): which lane / register holds A[i][k], B[k][j], D[i][j] of block b.
// hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_4x4x4_layout.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef __attribute__((ext_vector_type(4))) short bf16x4s;
typedef __attribute__((ext_vector_type(4))) float f32x4;
static __device__ __host__ uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }
__global__ void k(float* out) {
  const int lane = threadIdx.x, b = lane >> 2, r = lane & 3;
  // hypothesis: A lane (b, i = r) holds A_b[i][k = 0..3]; B lane (b, j = r) holds B_b[k = 0..3][j]; D lane (b, j = r), reg i
  bf16x4s a, bb;
  for (int kk = 0; kk < 4; ++kk) {
    const float av = (float)(1 + r) + 0.25f * kk + (b == 3 ? 8.f : 0.f);          // A_b[i][k]
    const float bv = (kk == 0 ? 1.f : 0.f) * (float)(1 + r) + (kk == 1 ? 16.f * (1 + r) : 0.f);   // B_b[k][j]: picks A[i][0]*(1+j) + A[i][1]*16(1+j)
    a[kk] = (short)f2bf(av); bb[kk] = (short)f2bf(bv);
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, bb, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = c[i];
}
int main() {
  float* d; hipMalloc(&d, 64 * 4 * 4);
  k<<<1, 64>>>(d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) for (int i = 0; i < 4; ++i) {
    const int b = lane >> 2, j = lane & 3;
    const float a0 = (float)(1 + i) + (b == 3 ? 8.f : 0.f), a1 = a0 + 0.25f;
    const float exp = a0 * (1 + j) + a1 * 16.f * (1 + j);
    if (fabsf(h[lane * 4 + i] - exp) > 1e-3f) { if (bad < 8) printf("lane %d reg %d: got %f expected %f\n", lane, i, h[lane * 4 + i], exp); ++bad; }
  }
  printf(bad ? "layout hypothesis WRONG (%d)\n" : "layout hypothesis OK: D lane (b, j) reg i = sum_k A_b[i][k] B_b[k][j]; A lane (b, i), B lane (b, j), k in the 4 packed values\n", bad);
  return 0;
}
