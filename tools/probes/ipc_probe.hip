// ipc_probe.hip — does the one-sided IPC transport (kandinsky-5_amd/csrc/ipc_comm.h) work between PROCESSES on this box?
//   hipcc --offload-arch=gfx950 -O2 -I kandinsky-5_amd/csrc tools/probes/ipc_probe.hip -o tools/probes/ipc_probe.bin -lrt
//   tools/probes/ipc_probe.bin [ranks=2] [MB per rank=64] [rounds=20]
// P processes are forked BEFORE the first HIP call, all on device 0 (or device rank % ndev with K5_IPC_SPREAD=1).  Each fills its slot
// of an in-place all-gather buffer with a pattern that depends on (rank, round), runs the collective, checks every peer's slot on the
// device, and reports the bytes per second it pulled; then an all-to-all and a sliced exchange; then a deliberate re-allocation of
// the buffer (the export table must notice).  Exit code 0 = every check passed on every rank.
#include "ipc_comm.h"

#include <sys/wait.h>

__global__ void fill_kernel(uint32_t* p, size_t n, uint32_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = seed * 2654435761u + (uint32_t)i * 40503u;
}
__global__ void check_kernel(const uint32_t* p, size_t n, uint32_t seed, unsigned long long* bad) {
  unsigned long long b = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b += p[i] != seed * 2654435761u + (uint32_t)i * 40503u;
  if (b) atomicAdd(bad, b);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s -> %s\n", rank, #x, hipGetErrorString(e_)); return 2; } } while (0)
#define GK(x) do { if ((x) != 0) { fprintf(stderr, "rank %d: %s: %s\n", rank, #x, g.err.c_str()); return 3; } } while (0)

static int run_rank(const char* name, int rank, int world, size_t mb, int rounds) {
  int ndev = 0;
  CK(hipGetDeviceCount(&ndev));
  CK(hipSetDevice(getenv("K5_IPC_SPREAD") ? rank % ndev : 0));
  k5ipc::Group g;
  GK(g.open(name, rank, world));
  const size_t chunk = mb << 20, words = chunk / 4;
  uint32_t* buf = nullptr; unsigned long long* bad = nullptr;
  CK(hipMalloc((void**)&buf, chunk * world));
  CK(hipMalloc((void**)&bad, 8));
  CK(hipMemset(bad, 0, 8));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best_ms = 1e30f;
  for (int r = 0; r < rounds; ++r) {
    fill_kernel<<<512, 256, 0, s>>>(buf + (size_t)rank * words, words, (uint32_t)(rank * 1000 + r));
    CK(hipEventRecord(e0, s));
    GK(g.all_gather_inplace(buf, chunk, s));
    CK(hipEventRecord(e1, s));
    for (int p = 0; p < world; ++p) check_kernel<<<512, 256, 0, s>>>(buf + (size_t)p * words, words, (uint32_t)(p * 1000 + r), bad);
    CK(hipStreamSynchronize(s));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0 && ms < best_ms) best_ms = ms;
  }
  unsigned long long hb = 0;
  CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
  printf("rank %d: all-gather x %d of %zu MB per rank: %llu bad words, best %.3f ms = %.1f GB/s pulled (flag page: %s memory)\n", rank, rounds, mb, hb, best_ms,
         (double)chunk * (world - 1) / best_ms * 1e-6, g.flags_fine ? "fine-grained" : "coarse-grained");
  int rc = hb ? 4 : 0;

  // sliced exchange: four quarters of the slot, each its own collective
  for (int r = 0; r < 3; ++r) {
    fill_kernel<<<512, 256, 0, s>>>(buf + (size_t)rank * words, words, (uint32_t)(rank * 1000 + 500 + r));
    for (int q = 0; q < 4; ++q) GK(g.slot_exchange(buf, chunk, q * (chunk / 4), chunk / 4, s));
    for (int p = 0; p < world; ++p) check_kernel<<<512, 256, 0, s>>>(buf + (size_t)p * words, words, (uint32_t)(p * 1000 + 500 + r), bad);
  }
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
  printf("rank %d: sliced exchange: %llu bad words\n", rank, hb);
  if (hb) rc = 4;

  // all-to-all: block p of my send buffer carries seed (rank, p); afterwards block p of recv carries (p, rank)
  uint32_t* recv = nullptr;
  CK(hipMalloc((void**)&recv, chunk * world));
  for (int p = 0; p < world; ++p) fill_kernel<<<512, 256, 0, s>>>(buf + (size_t)p * words, words, (uint32_t)(7000 + rank * 64 + p));
  CK(hipMemcpyAsync(recv + (size_t)rank * words, buf + (size_t)rank * words, chunk, hipMemcpyDeviceToDevice, s));
  GK(g.all_to_all(buf, recv, chunk, s));
  for (int p = 0; p < world; ++p) check_kernel<<<512, 256, 0, s>>>(recv + (size_t)p * words, words, (uint32_t)(7000 + p * 64 + rank), bad);
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
  printf("rank %d: all-to-all: %llu bad words\n", rank, hb);
  if (hb) rc = 4;

  // the buffer is freed and re-made (rank-dependent size, so the addresses differ): the export table must follow
  CK(hipFree(buf)); k5ipc::note_free(buf);
  uint32_t* dummy = nullptr;
  if (rank & 1) CK(hipMalloc((void**)&dummy, 3 << 20));
  CK(hipMalloc((void**)&buf, chunk * world));
  fill_kernel<<<512, 256, 0, s>>>(buf + (size_t)rank * words, words, (uint32_t)(rank * 1000 + 900));
  GK(g.all_gather_inplace(buf, chunk, s));
  for (int p = 0; p < world; ++p) check_kernel<<<512, 256, 0, s>>>(buf + (size_t)p * words, words, (uint32_t)(p * 1000 + 900), bad);
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
  printf("rank %d: after re-allocation: %llu bad words\n", rank, hb);
  if (hb) rc = 4;

  // small odd-sized gather (the key statistics: 28 floats per rank)
  float* st = nullptr;
  CK(hipMalloc((void**)&st, 28 * 4 * world));
  fill_kernel<<<1, 64, 0, s>>>((uint32_t*)st + rank * 28, 28, (uint32_t)(rank + 31));
  GK(g.all_gather_inplace(st, 28 * 4, s));
  for (int p = 0; p < world; ++p) check_kernel<<<1, 64, 0, s>>>((uint32_t*)st + p * 28, 28, (uint32_t)(p + 31), bad);
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
  uint32_t ew = 0;
  GK(g.error_word(&ew));
  printf("rank %d: 112-byte gather: %llu bad words; timed-out waits: %s (0x%08x); %lld collectives, %.1f MB pulled\n", rank, hb, ew ? "YES" : "none", ew,
         g.collectives, g.bytes_pulled * 1e-6);
  if (hb || ew) rc = 4;
  if (g.barrier()) rc = 5;
  g.close_all();
  return rc;
}

int main(int argc, char** argv) {
  const int world = argc > 1 ? atoi(argv[1]) : 2;
  const size_t mb = argc > 2 ? (size_t)atoi(argv[2]) : 64;
  const int rounds = argc > 3 ? atoi(argv[3]) : 20;
  char name[64];
  snprintf(name, sizeof(name), "/k5ipc_probe_%d", (int)getpid());
  std::vector<pid_t> kids;
  for (int r = 0; r < world; ++r) {
    const pid_t pid = fork();
    if (pid == 0) { const int rc = run_rank(name, r, world, mb, rounds); fflush(stdout); fflush(stderr); _exit(rc); }
    kids.push_back(pid);
  }
  int rc = 0;
  for (pid_t k : kids) { int st = 0; waitpid(k, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st)) rc = 1; }
  printf("ipc_probe: %d ranks: %s\n", world, rc ? "FAILED" : "ok");
  return rc;
}
