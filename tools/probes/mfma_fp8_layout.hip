// Operand layout probe for v_mfma_scale_f32_16x16x128_f8f6f4 with fp8 (e4m3) operands on gfx950.
// Hypothesis: A operand lane l holds row l&15, k = 32*(l>>4) .. +31 (8 dwords, little-endian bytes); B likewise for column
// l&15; D: lane l reg r = D[4*(l>>4) + r][l&15].  Small-integer operands are exact in e4m3, so the check is exact.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(const uint8_t* A, const uint8_t* B, float* D) {   // A [16][128], B [16][128] fp8 bytes, D [16][16]
  const int l = threadIdx.x, r16 = l & 15, g = l >> 4;
  v8i a, b;
  for (int w = 0; w < 8; ++w) {
    a[w] = *reinterpret_cast<const int*>(A + r16 * 128 + 32 * g + 4 * w);
    b[w] = *reinterpret_cast<const int*>(B + r16 * 128 + 32 * g + 4 * w);
  }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + r16] = acc[r];
}

static uint8_t enc(int v) {   // small integers -8..8 -> e4m3 (bias 7)
  if (v == 0) return 0;
  uint8_t s = v < 0 ? 0x80 : 0; int a = v < 0 ? -v : v;
  int e = 0; while ((1 << (e + 1)) <= a) ++e;          // a in [2^e, 2^(e+1))
  int m = ((a << 3) >> e) & 7;                           // 3 mantissa bits (exact for a <= 15 with e <= 3)
  return s | (uint8_t)((e + 7) << 3) | (uint8_t)m;
}

int main() {
  std::vector<uint8_t> A(16 * 128), B(16 * 128); std::vector<int> Ai(16 * 128), Bi(16 * 128);
  unsigned seed = 12345;
  for (int i = 0; i < 16 * 128; ++i) { seed = seed * 1664525u + 1013904223u; Ai[i] = (int)((seed >> 16) % 9) - 4; A[i] = enc(Ai[i]);
                                       seed = seed * 1664525u + 1013904223u; Bi[i] = (int)((seed >> 16) % 9) - 4; B[i] = enc(Bi[i]); }
  uint8_t *dA, *dB; float* dD; hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  std::vector<float> D(256); hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    long ref = 0; for (int kk = 0; kk < 128; ++kk) ref += (long)Ai[i * 128 + kk] * Bi[j * 128 + kk];
    double e = fabs((double)D[i * 16 + j] - (double)ref); if (e > maxerr) maxerr = e;
  }
  printf("fp8 16x16x128 layout hypothesis: max |D - ref| = %g  (D[0][0]=%g D[3][5]=%g)\n", maxerr, D[0], D[3 * 16 + 5]);
  return 0;
}
