// Which XCD does workgroup b of a 1-D grid land on?  The stream-K schedule of the four-wave GEMM assumes b & 7 (round-robin dispatch) and lets
// workgroups of one XCD exchange partial sums through that XCD's L2.  Prints the HW_REG_XCC_ID of every workgroup for a few grid sizes / LDS
// footprints (133 KB = one workgroup per CU, as the GEMM).   hipcc --offload-arch=gfx950 -O2 -o xcc_map tools/probes/xcc_map.hip && ./xcc_map
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
  extern __shared__ char lds[];
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  unsigned cu;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(cu));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = x; out[2 * blockIdx.x + 1] = cu; }
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
  if (threadIdx.x == 1) lds[0] = 1;
}
int main() {
  for (int grid : {256, 512, 168, 1024}) for (int ldsb : {0, 136 * 1024}) {
    unsigned* d; hipMalloc(&d, grid * 8); hipMemset(d, 0xff, grid * 8);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), ldsb, 0, d, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * grid); hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
    int bad = 0; for (int b = 0; b < grid; ++b) bad += ((h[2 * b] & 15) != (unsigned)(b & 7));
    printf("grid %4d lds %6d: %d of %d workgroups NOT on XCD b & 7; first 16 xcc ids:", grid, ldsb, bad, grid);
    for (int b = 0; b < 16; ++b) printf(" %u", h[2 * b] & 15);
    printf("   raw[0] = 0x%x\n", h[0]);
    hipFree(d);
  }
  return 0;
}
