// ipc_comm.h — the one-sided transport of the sequence-parallel exchange: peers read each other's K / V^T slots (and the CFG
// pair's velocity) straight out of IPC-mapped device buffers, ordered by epoch flags in device memory.  SURVEY.md §8(e) names it
// next to RCCL ("direct peer writes over xGMI into IPC-mapped buffers"); it replaces the all-reduces of the reference's DTensor
// plan (kandinsky/models/parallelize.py:11-102) under the launch contract of kandinsky/utils.py:40-55 (one process per rank).
//
// Why it exists beside RCCL: (1) RCCL refuses two ranks on one device, so on a one-GPU lease the process boundary of the sharded
// path — torch.distributed.run -> LOCAL_RANK -> handle exchange -> per-process streams -> rank_check — could never execute;
// this transport runs P processes on ONE device as well as on P devices of an xGMI node.  (2) It has no rendezvous inside the
// library: a rank's slice is readable by its peers the moment the producing kernel's stream reaches the signal, so the sliced
// exchange costs one flag store per slice instead of a grouped send / recv.
//
// Mechanics
//   control block   POSIX shared memory (name from the host, which broadcasts it over torch.distributed): a sense-reversing
//                   barrier, every rank's flag-buffer handle, every rank's table of exported allocations, and a two-deep slot
//                   per rank that says which allocation + offset the rank passed to the collective being issued.
//   device flags    per rank uint32 [2][MAXR] in device memory, exported once: READY[p] = "rank p's bytes for epoch e are
//                   written", PULLED[p] = "rank p has read mine".  Written by the peers' signal kernels with system-scope release
//                   stores, polled by the rank's own wait kernel with system-scope acquire loads (bounded: a wait that outlives
//                   K5_IPC_TIMEOUT_S — default 60 s of the 100-MHz wall clock — raises the error word instead of hanging the GPU).
//   a collective    host: publish (allocation, offset), ONE barrier, map the peers' allocations (cached by serial number; an
//                   allocation freed and re-made is re-opened); device, on the caller's stream: signal READY (bumps the epoch, which
//                   lives in the rank's own flag buffer) -> per peer { wait READY[p] ; copy p's bytes out of p's buffer } -> signal
//                   PULLED -> wait PULLED[all].  The call therefore completes, stream-wise, exactly when an RCCL all-gather would:
//                   the caller may overwrite its own slot afterwards.  The kernels carry no host-side state (pointers are those of
//                   buffers that do not move between steps), so a captured step that contains collectives replays correctly.
// The same header is compiled into tools/probes/ipc_probe.hip, which checks the mechanism on its own (P forked processes).
#pragma once
#include <hip/hip_runtime.h>

#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

namespace k5ipc {

constexpr int MAXR = 16;     // ranks per group (one node: 8; CFG pair: 2)
constexpr int MAXREG = 64;   // exported allocations per rank
constexpr uint32_t MAGIC = 0x4b354950u;   // "K5IP"

struct ShmReg { uint64_t base, size, serial; hipIpcMemHandle_t h; };
struct ShmCall { uint32_t idx, pad; uint64_t serial, off, bytes; };
struct ShmRank {
  hipIpcMemHandle_t flags_h;
  int pid, device;
  ShmReg reg[MAXREG];
  ShmCall call[2];
};
struct Shm {
  std::atomic<uint32_t> magic, arrived, bar_count, bar_gen, failed;
  int world;
  ShmRank r[MAXR];
};

struct PeerFlags { uint32_t* f[MAXR]; };

// ---- device side ----
// flag buffer of a rank (uint32): [0, MAXR) READY | [MAXR, 2 MAXR) PULLED | [ERRW] first timed-out wait | [CTR] the epoch of the collective in flight.
// The epoch lives ON THE DEVICE: the first kernel of a collective bumps it, the others read it — nothing about a collective's kernels depends on a
// host-side counter, so a captured graph that contains collectives can be replayed (k5_dit_set_graph).  All collectives of a group are issued in one
// order on streams that are chained by events (engine.hip: the side stream's gathers, then the velocity gather after the join), so every rank's
// device counts the same collectives in the same order.
constexpr int ERRW = 2 * MAXR, CTR = 2 * MAXR + 1;
__global__ void ipc_signal_kernel(PeerFlags pf, uint32_t* mine, int world, int rank, int which, int bump) {
  const int p = threadIdx.x;
  const uint32_t epoch = mine[CTR] + (bump ? 1u : 0u);
  __syncthreads();
  if (bump && p == 0) mine[CTR] = epoch;
  if (p >= world || p == rank) return;
  __hip_atomic_store(pf.f[p] + which * MAXR + rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// waits until flags[which][p] has reached the current epoch for p = first .. first + count - 1 (p != rank); limit in 100-MHz ticks
__global__ void ipc_wait_kernel(uint32_t* flags, int first, int count, int rank, int which, unsigned long long limit) {
  const int p = first + threadIdx.x;
  if ((int)threadIdx.x >= count || p == rank) return;
  const uint32_t epoch = flags[CTR];
  const uint32_t* f = flags + which * MAXR + p;
  const unsigned long long t0 = wall_clock64();
  // relaxed polls (a system-scope load goes to memory on its own; an acquire per poll would invalidate this XCD's L2 under the kernels
  // that run beside the wait), one acquire when the flag is there — the consumer is the next kernel of the stream anyway
  while ((int32_t)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
    __builtin_amdgcn_s_sleep(16);
    if (wall_clock64() - t0 > limit) {
      __hip_atomic_store(flags + ERRW, 0x80000000u | ((uint32_t)which << 24) | ((uint32_t)p << 16) | (epoch & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}
// dst / src 16-byte aligned, n16 = 16-byte units; tail = remaining 4-byte words
typedef uint32_t ipc_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void ipc_copy_kernel(ipc_u32x4* __restrict__ dst, const ipc_u32x4* __restrict__ src, size_t n16, int tail_words) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
  if (blockIdx.x == 0 && (int)threadIdx.x < tail_words)
    ((uint32_t*)(dst + n16))[threadIdx.x] = ((const uint32_t*)(src + n16))[threadIdx.x];
}
__global__ void ipc_copy_words_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// ---- host side ----
struct Group;
inline std::mutex& registry_mutex() { static std::mutex m; return m; }
inline std::vector<Group*>& registry() { static std::vector<Group*> v; return v; }

struct Group {
  std::string name, err;
  int rank = 0, world = 1;
  Shm* shm = nullptr;
  uint32_t* flags = nullptr;        // mine (device)
  bool flags_fine = false;          // the flag page is fine-grained memory (open)
  uint32_t* errword = nullptr;      // device: first wait that timed out
  PeerFlags peer_flags{};           // the peers' flag buffers mapped here (mine at [rank])
  uint64_t calls = 0, next_serial = 1;
  unsigned long long wait_limit = 6000000000ull;   // 60 s of the 100-MHz wall clock
  double host_timeout_s = 120.0;
  struct Mapped { uint64_t serial = 0; void* base = nullptr; };
  Mapped mapped[MAXR][MAXREG];
  struct Range { void* base; size_t size; };
  std::vector<Range> ranges;      // allocations of THIS process already looked up (forget() drops them)
  long long bytes_pulled = 0, collectives = 0;

  int fail(const char* fmt, const char* a = "", long long b = 0) {
    char buf[512]; snprintf(buf, sizeof(buf), fmt, a, b); err = buf;
    if (shm) shm->failed.store(1, std::memory_order_release);
    return -1;
  }
  static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

  // sense-reversing barrier over the processes of the group; -1 on timeout or when a peer has failed
  int barrier() {
    const uint32_t gen = shm->bar_gen.load(std::memory_order_acquire);
    if (shm->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world) {
      shm->bar_count.store(0, std::memory_order_relaxed);
      shm->bar_gen.store(gen + 1, std::memory_order_release);
    } else {
      const double t0 = now();
      int spins = 0;
      while (shm->bar_gen.load(std::memory_order_acquire) == gen) {
        if (++spins < 2000) continue;
        sched_yield();
        if ((spins & 1023) == 0) {
          if (shm->failed.load(std::memory_order_acquire)) return fail("a peer of IPC group %s reported a failure", name.c_str());
          if (now() - t0 > host_timeout_s) return fail("IPC group %s: a peer did not reach the barrier within %lld s", name.c_str(), (long long)host_timeout_s);
        }
      }
    }
    if (shm->failed.load(std::memory_order_acquire)) return fail("a peer of IPC group %s reported a failure", name.c_str());
    return 0;
  }

#define K5IPC_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail("IPC transport: %s failed (code %lld)", #x, (long long)e_); } while (0)

  int open(const char* shm_name, int rank_, int world_) {
    if (!shm_name || !shm_name[0] || world_ < 1 || world_ > MAXR || rank_ < 0 || rank_ >= world_) return fail("bad IPC group arguments");
    name = shm_name[0] == '/' ? shm_name : std::string("/") + shm_name;
    rank = rank_; world = world_;
    if (const char* e = getenv("K5_IPC_TIMEOUT_S")) { const double s = atof(e); if (s > 0) { wait_limit = (unsigned long long)(s * 1e8); host_timeout_s = 2 * s; } }
    const int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0) return fail("shm_open(%s) failed: errno %lld", name.c_str(), errno);
    if (ftruncate(fd, sizeof(Shm)) != 0) { close(fd); return fail("ftruncate(%s) failed: errno %lld", name.c_str(), errno); }
    void* m = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return fail("mmap(%s) failed: errno %lld", name.c_str(), errno);
    shm = (Shm*)m;   // a fresh segment is zero-filled: every atomic starts at 0
    // The flag page is FINE-GRAINED device memory where the runtime offers it: peers on OTHER devices store into it and this rank's wait kernels poll it
    // while kernels run, which coarse-grained memory only promises to make visible at kernel boundaries (between processes of ONE device the same L2 /
    // memory side serves both and either kind works: what the one-GPU runs of this transport exercised).  Falls back to plain hipMalloc if the
    // allocation or its export is refused.
    int dev = 0;
    K5IPC_HIP(hipGetDevice(&dev));
    ShmRank& me = shm->r[rank];
    me.pid = (int)getpid(); me.device = dev;
    static const bool coarse = getenv("K5_IPC_COARSE_FLAGS") != nullptr;   // A/B switch
    flags_fine = false;
    if (!coarse && hipExtMallocWithFlags((void**)&flags, 4096, hipDeviceMallocFinegrained) == hipSuccess && flags) {
      flags_fine = true;
      if (hipMemset(flags, 0, 4096) != hipSuccess || (world > 1 && hipIpcGetMemHandle(&me.flags_h, flags) != hipSuccess)) {
        (void)hipGetLastError(); (void)hipFree(flags); flags = nullptr; flags_fine = false;
      }
    } else { (void)hipGetLastError(); flags = nullptr; }
    if (!flags) {
      K5IPC_HIP(hipMalloc((void**)&flags, 4096));
      K5IPC_HIP(hipMemset(flags, 0, 4096));
      if (world > 1) K5IPC_HIP(hipIpcGetMemHandle(&me.flags_h, flags));
    }
    errword = flags + ERRW;
    K5IPC_HIP(hipDeviceSynchronize());
    shm->arrived.fetch_add(1, std::memory_order_acq_rel);
    const double t0 = now();
    while (shm->arrived.load(std::memory_order_acquire) < (uint32_t)world) {
      sched_yield();
      if (now() - t0 > host_timeout_s) return fail("IPC group %s: only %lld ranks arrived", name.c_str(), (long long)shm->arrived.load());
    }
    peer_flags.f[rank] = flags;
    for (int p = 0; p < world; ++p) {
      if (p == rank) continue;
      void* q = nullptr;
      K5IPC_HIP(hipIpcOpenMemHandle(&q, shm->r[p].flags_h, hipIpcMemLazyEnablePeerAccess));
      peer_flags.f[p] = (uint32_t*)q;
    }
    if (barrier()) return -1;
    if (rank == 0) shm_unlink(name.c_str());   // everybody has it mapped: the name can go, the segment lives until the last unmap
    std::lock_guard<std::mutex> g(registry_mutex());
    registry().push_back(this);
    return 0;
  }

  void close_all() {
    {
      std::lock_guard<std::mutex> g(registry_mutex());
      auto& v = registry();
      for (size_t i = 0; i < v.size(); ++i) if (v[i] == this) { v.erase(v.begin() + i); break; }
    }
    (void)hipDeviceSynchronize();
    for (int p = 0; p < world; ++p) {
      if (p == rank) continue;
      for (auto& mp : mapped[p]) if (mp.base) { (void)hipIpcCloseMemHandle(mp.base); mp = Mapped(); }
      if (peer_flags.f[p]) (void)hipIpcCloseMemHandle(peer_flags.f[p]);
    }
    if (flags) (void)hipFree(flags);
    flags = nullptr;
    if (shm) munmap(shm, sizeof(Shm));
    shm = nullptr;
  }

  // an allocation of this process went away (DevBuf::release): its export is void — a later allocation at the same address gets a new serial
  void forget(void* base) {
    if (!shm) return;
    ShmRank& me = shm->r[rank];
    for (auto& rg : me.reg) if (rg.base == (uint64_t)(uintptr_t)base) { rg.base = 0; rg.size = 0; rg.serial = 0; }
    for (size_t i = 0; i < ranges.size(); ++i) if (ranges[i].base == base) { ranges.erase(ranges.begin() + i); break; }
  }

  // host half of a collective: tell the peers which allocation / offset `buf` is, learn theirs.  peer[p] = rank p's `buf` mapped here.
  int resolve(const void* buf, size_t bytes, void** peer) {
    ShmRank& me = shm->r[rank];
    hipDeviceptr_t base = nullptr; size_t size = 0;
    for (const Range& rg : ranges) if ((const char*)buf >= (const char*)rg.base && (const char*)buf < (const char*)rg.base + rg.size) { base = rg.base; size = rg.size; break; }
    if (!base) {   // (not inside a stream capture: the first step of a sampling run is never captured)
      K5IPC_HIP(hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)buf));
      ranges.push_back(Range{base, size});
    }
    int idx = -1, free_idx = -1;
    for (int i = 0; i < MAXREG; ++i) {
      if (me.reg[i].base == (uint64_t)(uintptr_t)base && me.reg[i].size == size) { idx = i; break; }
      if (me.reg[i].base == (uint64_t)(uintptr_t)base) me.reg[i] = ShmReg{};   // same address, another size: the old allocation is gone
      if (free_idx < 0 && me.reg[i].base == 0) free_idx = i;
    }
    if (idx < 0) {
      if (free_idx < 0) return fail("IPC group %s: more than %lld exported allocations", name.c_str(), MAXREG);
      idx = free_idx;
      ShmReg rg{};
      K5IPC_HIP(hipIpcGetMemHandle(&rg.h, base));
      rg.base = (uint64_t)(uintptr_t)base; rg.size = size; rg.serial = next_serial++;
      me.reg[idx] = rg;
    }
    ShmCall& c = me.call[calls & 1];
    c.idx = (uint32_t)idx; c.serial = me.reg[idx].serial; c.off = (uint64_t)((const char*)buf - (const char*)base); c.bytes = bytes;
    if (barrier()) return -1;   // (release / acquire through the barrier's atomics)
    for (int p = 0; p < world; ++p) {
      if (p == rank) { peer[p] = const_cast<void*>(buf); continue; }
      const ShmCall pc = shm->r[p].call[calls & 1];
      if (pc.bytes != bytes) return fail("IPC group %s: ranks disagree about the size of a collective (%lld bytes here)", name.c_str(), (long long)bytes);
      Mapped& mp = mapped[p][pc.idx];
      if (mp.serial != pc.serial) {
        if (mp.base) { (void)hipIpcCloseMemHandle(mp.base); mp = Mapped(); }
        void* q = nullptr;
        K5IPC_HIP(hipIpcOpenMemHandle(&q, shm->r[p].reg[pc.idx].h, hipIpcMemLazyEnablePeerAccess));
        mp.base = q; mp.serial = pc.serial;
      }
      peer[p] = (char*)mp.base + pc.off;
    }
    ++calls;
    return 0;
  }

  int signal(int which, hipStream_t s) {   // which = 0 (READY) opens a collective: it bumps the device-side epoch
    hipLaunchKernelGGL(ipc_signal_kernel, dim3(1), dim3(64), 0, s, peer_flags, flags, world, rank, which, which == 0 ? 1 : 0);
    return 0;
  }
  int wait(int first, int count, int which, hipStream_t s) {
    hipLaunchKernelGGL(ipc_wait_kernel, dim3(1), dim3(64), 0, s, flags, first, count, rank, which, wait_limit);
    return 0;
  }
  int copy(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (!bytes) return 0;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0 && (bytes & 3) == 0) {
      const size_t n16 = bytes >> 4;
      const int tail = (int)((bytes & 15) >> 2);
      const unsigned grid = (unsigned)(n16 / 256 > 2048 ? 2048 : (n16 + 255) / 256 ? (n16 + 255) / 256 : 1);
      hipLaunchKernelGGL(ipc_copy_kernel, dim3(grid), dim3(256), 0, s, (ipc_u32x4*)dst, (const ipc_u32x4*)src, n16, tail);
    } else if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 3) == 0) {
      const size_t n = bytes >> 2;
      const unsigned grid = (unsigned)(n / 256 > 1024 ? 1024 : (n + 255) / 256);
      hipLaunchKernelGGL(ipc_copy_words_kernel, dim3(grid), dim3(256), 0, s, (uint32_t*)dst, (const uint32_t*)src, n);
    } else return fail("IPC transport: a copy that is not 4-byte aligned");
    bytes_pulled += (long long)bytes;
    return 0;
  }

  // in-place all-gather: rank p's chunk sits at buf + p * chunk in rank p's buffer and ends up there in everybody's
  int all_gather_inplace(void* buf, size_t chunk, hipStream_t s) { return slot_exchange(buf, chunk, 0, chunk, s); }

  // bytes [off, off + cnt) of every rank's slot (slot_bytes each) travel to every peer
  int slot_exchange(void* buf, size_t slot_bytes, size_t off, size_t cnt, hipStream_t s) {
    if (world == 1 || cnt == 0) return 0;
    void* peer[MAXR];
    if (resolve(buf, slot_bytes, peer)) return -1;
    ++collectives;
    signal(0, s);
    for (int i = 1; i < world; ++i) {
      const int p = (rank + i) % world;   // every rank starts at a different peer
      wait(p, 1, 0, s);
      if (copy((char*)buf + (size_t)p * slot_bytes + off, (const char*)peer[p] + (size_t)p * slot_bytes + off, cnt, s)) return -1;
    }
    signal(1, s);
    wait(0, world, 1, s);
    return 0;
  }

  // all-to-all: block p of `send` goes to rank p, block p of `recv` comes from rank p (the rank's own block is the caller's copy)
  int all_to_all(const void* send, void* recv, size_t block_bytes, hipStream_t s) {
    if (world == 1) return 0;
    void* peer[MAXR];
    if (resolve(send, block_bytes, peer)) return -1;
    ++collectives;
    signal(0, s);
    for (int i = 1; i < world; ++i) {
      const int p = (rank + i) % world;
      wait(p, 1, 0, s);
      if (copy((char*)recv + (size_t)p * block_bytes, (const char*)peer[p] + (size_t)rank * block_bytes, block_bytes, s)) return -1;
    }
    signal(1, s);
    wait(0, world, 1, s);
    return 0;
  }

  // first wait that timed out on this rank's device since the group was made (0 = none); synchronises the device
  int error_word(uint32_t* out) {
    K5IPC_HIP(hipDeviceSynchronize());
    K5IPC_HIP(hipMemcpy(out, errword, 4, hipMemcpyDeviceToHost));
    return 0;
  }
#undef K5IPC_HIP
};

inline void note_free(void* base) {
  if (!base) return;
  std::lock_guard<std::mutex> g(registry_mutex());
  for (Group* grp : registry()) grp->forget(base);
}

}  // namespace k5ipc
