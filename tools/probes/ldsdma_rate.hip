// How fast can one CU pull an L2-resident stream into its LDS with `buffer_load_dwordx4 ... lds`, and does it depend on how many waves issue?
// Every workgroup (one per CU: 128 KB of LDS claimed) has NW waves; each wave keeps DEPTH 1-KB pieces in flight into its own LDS slots, walking its
// own slice of a buffer that fits the L2s / the Infinity Cache (default 64 MB, read ITERS times).  Also: the same with plain global_load_dwordx4 into
// registers (no LDS), for the other path.  Prints GB/s per CU and TB/s per chip.
//   hipcc --offload-arch=gfx950 -O3 -o ldsdma_rate.bin tools/probes/ldsdma_rate.hip && ./ldsdma_rate.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void lds_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NW, int DEPTH, bool TO_LDS, bool SHARED>
__global__ __launch_bounds__(64 * NW) void stream(const char* buf, size_t bytes_per_wg, int pieces_per_wave, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = buf + (size_t)(SHARED ? (blockIdx.x & 7) : blockIdx.x) * bytes_per_wg + (size_t)wave * pieces_per_wave * 1024;   // SHARED: the 32 workgroups of an XCD read the same bytes (L2 hits, as the GEMM's operand tiles)
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, pieces_per_wave * 1024, 0x00020000);
  uint32_t v = lane * 16;
  f32x4 acc = {0, 0, 0, 0};
  for (int i = 0; i < pieces_per_wave; i += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (TO_LDS) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_t*)(dsm + (wave * DEPTH + d) * 1024), 16, v, (uint32_t)(i + d) * 1024u, 0, 0);
      else { f32x4 t; asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(t) : "v"(v), "s"(r), "s"((uint32_t)(i + d) * 1024u)); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 1)); acc += t; }
    }
    if (TO_LDS) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH / 2) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!TO_LDS && acc[0] == 12345.f) sink[0] = acc[1];
  if (TO_LDS && dsm[threadIdx.x] == 77 && sink == nullptr) sink[1] = 1;
}

#define RUN(NW, DEPTH, TO_LDS, SHARED) do { \
    const int ppw = (int)(per_wg / 1024 / NW); \
    CHECK(hipFuncSetAttribute((const void*)stream<NW, DEPTH, TO_LDS, SHARED>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)); \
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stream<NW, DEPTH, TO_LDS, SHARED>), dim3(cus), dim3(64 * NW), 128 * 1024, 0, buf, per_wg, ppw, sink); \
    CHECK(hipDeviceSynchronize()); \
    CHECK(hipEventRecord(e0)); \
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((stream<NW, DEPTH, TO_LDS, SHARED>), dim3(cus), dim3(64 * NW), 128 * 1024, 0, buf, per_wg, ppw, sink); \
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); \
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); \
    const double bytes = (double)ppw * NW * 1024 * cus * iters; \
    printf("%-4s %-6s %2d waves x %2d pieces in flight: %6.1f GB/s per CU, %5.2f TB/s\n", TO_LDS ? "LDS" : "VGPR", SHARED ? "shared" : "own", NW, DEPTH, bytes / (ms * 1e-3) / cus / 1e9, bytes / (ms * 1e-3) / 1e12); } while (0)

int main(int argc, char** argv) {
  const size_t total = (size_t)(argc > 1 ? atoi(argv[1]) : 64) << 20;
  const int iters = 20;
  int dev = 0; hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  const size_t per_wg = total / cus / (16 * 1024) * (16 * 1024);
  char* buf; float* sink;
  CHECK(hipMalloc(&buf, total + (1 << 20))); CHECK(hipMemset(buf, 1, total)); CHECK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("%d CUs, %zu MB read %d times, %zu KB per workgroup and pass\n", cus, total >> 20, iters, per_wg >> 10);
  RUN(4, 8, true, false); RUN(4, 8, false, false);
  RUN(1, 8, true, true); RUN(2, 8, true, true); RUN(4, 4, true, true); RUN(4, 8, true, true); RUN(4, 16, true, true); RUN(8, 8, true, true); RUN(8, 16, true, true); RUN(16, 8, true, true);
  RUN(1, 8, false, true); RUN(4, 8, false, true); RUN(8, 8, false, true); RUN(16, 8, false, true);
  return 0;
}
