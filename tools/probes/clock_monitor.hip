// The shader clock a kernel actually runs at (DVFS): ONE wave that samples s_memtime (tick = shader cycle, MI355X_MICROARCH.md) next to
// s_memrealtime (constant 100 MHz) every `period` wall ticks, launched on a side stream BEFORE the kernel under test so that it sits on a CU
// beside it.  tools/clock_under_load.py drives it (GEMM and attention launches of the library on the main stream).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libclockmon.so clock_monitor.hip
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(64) void clock_monitor(unsigned long long* out, int n, int period) {
  if (threadIdx.x) return;
  unsigned long long next = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
    unsigned long long rt;
    do { __builtin_amdgcn_s_sleep(32); rt = __builtin_amdgcn_s_memrealtime(); } while (rt < next);
    out[2 * i] = rt;
    out[2 * i + 1] = __builtin_amdgcn_s_memtime();
    next = rt + period;
  }
}
extern "C" int clock_monitor_launch(void* out, int n, int period, void* stream) {
  hipLaunchKernelGGL(clock_monitor, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out, n, period);
  return (int)hipGetLastError();
}
