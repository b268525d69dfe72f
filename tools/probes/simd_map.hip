// Probe: which SIMD does wave w of a 512-thread (8-wave) workgroup land on?  (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = hw;
}
int main() {
  for (int threads : {256, 512}) {
    unsigned* d; const int nb = 1024, nw = threads / 64;
    hipMalloc(&d, nb * nw * 4);
    hipLaunchKernelGGL(k, dim3(nb), dim3(threads), 0, 0, d);
    unsigned* h = (unsigned*)malloc(nb * nw * 4);
    hipMemcpy(h, d, nb * nw * 4, hipMemcpyDeviceToHost);
    printf("threads=%d: SIMD_ID (bits 5:4) per wave, first 12 blocks; cu=(bits 11:8)\n", threads);
    for (int b = 0; b < 12; ++b) {
      printf(" blk %2d:", b);
      for (int w = 0; w < nw; ++w) printf(" %u", (h[b * nw + w] >> 4) & 3);
      printf("   cu %u se %u raw0 %08x\n", (h[b * nw] >> 8) & 15, (h[b * nw] >> 13) & 7, h[b * nw]);
    }
    int pat[8][4] = {};
    for (int b = 0; b < nb; ++b) for (int w = 0; w < nw; ++w) pat[w][(h[b * nw + w] >> 4) & 3]++;
    int same04 = 0, same01 = 0;
    for (int b = 0; b < nb; ++b) if (nw == 8) {
      same04 += ((h[b * nw] >> 4) & 3) == ((h[b * nw + 4] >> 4) & 3);
      same01 += ((h[b * nw] >> 4) & 3) == ((h[b * nw + 1] >> 4) & 3);
    }
    if (nw == 8) printf(" wave0/wave4 same SIMD in %d of %d blocks; wave0/wave1 same SIMD in %d\n", same04, nb, same01);
  }
  return 0;
}
