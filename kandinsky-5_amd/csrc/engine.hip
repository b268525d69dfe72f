// engine.hip — the denoising engine behind the C ABI: DiffusionTransformer3D forward
// (kandinsky/models/dit.py:155-181) and the flow-matching Euler / CFG loop
// (kandinsky/generation_utils.py:39-129), scheduled as a fixed sequence of gfx950 kernels on one
// HIP stream.  Weights are packed once (Wq|Wk concatenated, padded K, bf16 / fp32-island copies),
// workspaces are persistent, RoPE tables and the batched AdaLN modulation are computed once per
// forward.  No host synchronisation inside forward()/sample().
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <pthread.h>
#include <rccl/rccl.h>  // types / enums only: the library is dlopen()ed (see Comm)

#include "k5_common.h"
#include "k5_kernels.h"
#include "ipc_comm.h"   // one-sided IPC transport of the sharded path (P processes on one device or on the devices of one node)

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
void k5_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* k5_last_error(void) { return g_err; }
extern "C" int k5_abi_version(void) { return K5_ABI_VERSION; }

#define HIPCHK(x)                                                                          \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      k5_set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
      return K5_ERR_HIP;                                                                   \
    }                                                                                      \
  } while (0)
#define K5CHK(x)                                                             \
  do {                                                                       \
    int s_ = (x);                                                            \
    if (s_ != K5_OK) {                                                       \
      if (!g_err[0]) k5_set_error("%s -> status %d (%s:%d)", #x, s_, __FILE__, __LINE__); \
      return s_;                                                             \
    }                                                                        \
  } while (0)

namespace {

struct DevBuf {   // owning device buffer (move-only): freed with whatever holds it — the handle, a staged tensor, a block's weights
  void* p = nullptr;
  size_t bytes = 0;
  bool owned = true;   // false: a view into somebody else's buffer (alias), never freed from here
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), owned(o.owned) { o.p = nullptr; o.bytes = 0; o.owned = true; }
  DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; bytes = o.bytes; owned = o.owned; o.p = nullptr; o.bytes = 0; o.owned = true; } return *this; }
  ~DevBuf() { release(); }
  int ensure(size_t n) {
    if (n <= bytes) return K5_OK;
    release();
    HIPCHK(hipMalloc(&p, n));
    bytes = n;
    return K5_OK;
  }
  void release() { if (p && owned) { (void)hipFree(p); k5ipc::note_free(p); } p = nullptr; bytes = 0; owned = true; }   // note_free: an IPC export of this allocation is void from here on
  void alias(void* ptr, size_t n) { release(); p = ptr; bytes = n; owned = false; }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct HostTensor {  // staged state_dict entry: vectors (biases, norms) as fp32 on the host, matrices as a raw copy ON THE DEVICE in
  std::vector<float> data;          // the checkpoint's dtype — they are converted / padded / concatenated by a device kernel
  std::vector<int64_t> shape;       // (k5_launch_pack_matrix), so a checkpoint that already lives on the GPU never touches the host
  DevBuf dev; int dtype = K5_F32;   // and a host one costs its own bytes once (round 1: everything through host fp32, 4 B/param)
  size_t count() const { size_t n = 1; for (int64_t v : shape) n *= (size_t)v; return n; }
};

float half_to_float(uint16_t h) {
  const uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
  uint32_t bits;
  if (e == 0) {
    if (m == 0) bits = s << 31;
    else {
      int ee = -1; uint32_t mm = m;
      do { ++ee; mm <<= 1; } while (!(mm & 1024));
      bits = (s << 31) | ((uint32_t)(112 - ee) << 23) | ((mm & 1023) << 13);
    }
  } else if (e == 31) bits = (s << 31) | 0x7f800000u | (m << 13);
  else bits = (s << 31) | ((e + 112) << 23) | (m << 13);
  float f; memcpy(&f, &bits, 4); return f;
}
uint16_t f32_to_bf16_rne(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
float bf16_round_host(float f) { uint32_t u = (uint32_t)f32_to_bf16_rne(f) << 16; float r; memcpy(&r, &u, 4); return r; }

struct AttnW {  // one attention module, packed
  DevBuf wqk, wq, wk, wv, wo;       // bf16
  DevBuf bqk, bq, bk, bv, bo, norm; // fp32 (bias values bf16-rounded); norm = [q_norm | k_norm]
  DevBuf wqk8, wv8, wo8, sqk, sv, so;   // opt-in e4m3 copies (k5_dit_set_fp8 bits 1 / 2): weights + per-output-channel scales
  float score_bound = 0.f;          // |q.k| <= 64 max|w_q| max|w_k| after norm_qk (RoPE preserves norms)
  mutable DevBuf pref;              // visual self-attention: [2][H] heads the per-row-offset softmax served badly the last time this layer ran for
                                    // the cond (0) / uncond (1) branch (k5_launch_attn_pref_update); zeroed at the start of every k5_sample and
                                    // by k5_dit_set_option("attn_pref_reset") — never carried from one sampling run, prompt or shape to the next
};
struct BlockW {
  AttnW self_attn, cross_attn;
  DevBuf w1, w2;  // feed_forward in/out, bf16
  DevBuf w1_f8, w2_f8, s1_f8, s2_f8;  // opt-in fp8 copies (k5_dit_set_fp8): e4m3 weights + per-output-channel scales
  size_t mod_off = 0;  // offset (floats) of this block's modulation vector in mod_all
};

struct Prof { double ms = 0; int64_t n = 0; };

// RCCL over xGMI, one process per GPU.  The library is dlopen()ed (RTLD_LOCAL) from a path given by the
// host — normally the librccl.so that torch already mapped — so libk5 has no link-time RCCL dependency
// and never ends up with a second, conflicting copy of the nccl* symbols in the global namespace.
// Loopback group: P handles of ONE process on ONE GPU act as the P ranks of a sequence-parallel run (k5_loopback_create /
// k5_dit_comm_init_loopback).  Each rank is driven by its own host thread; a collective is a rendezvous of those threads
// (two pthread barriers) around device-to-device copies ordered by events, so every rank executes exactly the code path,
// offsets and launch sequence of a real multi-GPU run — what moves the bytes is hipMemcpyAsync instead of RCCL.
struct LoopGroup {
  int world = 0;
  pthread_barrier_t bar;
  std::vector<void*> ptr;
  std::vector<hipEvent_t> ready, pulled;
  std::vector<int> joined;
};

struct Comm {
  void* lib = nullptr;
  ncclComm_t comm = nullptr;
  LoopGroup* loop = nullptr;
  k5ipc::Group* ipc = nullptr;   // one process per rank, peers' buffers IPC-mapped (ipc_comm.h): k5_dit_comm_init_ipc / k5_dit_cfg_pair_init_ipc
  bool active() const { return comm != nullptr || loop != nullptr || ipc != nullptr; }
  int ipc_status(int r) { if (r) { k5_set_error("%s", ipc->err.c_str()); return K5_ERR_STATE; } return K5_OK; }
  int rank = 0, world = 1;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;   // optional: sliced exchange
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;   // optional: what the communicator itself says its size is (k5_dit_get_option "rccl_ranks")
  int open(const char* path) {
    if (lib) return K5_OK;
    const char* cands[] = {path, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* c : cands) {
      if (!c || !c[0]) continue;
      lib = dlopen(c, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) { k5_set_error("cannot dlopen RCCL (%s): %s", path ? path : "default names", dlerror()); return K5_ERR_STATE; }
    GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    Send = (decltype(Send))dlsym(lib, "ncclSend"); Recv = (decltype(Recv))dlsym(lib, "ncclRecv");
    GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart"); GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
    CommCount = (decltype(CommCount))dlsym(lib, "ncclCommCount");
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllGather || !GetErrorString) {
      k5_set_error("RCCL library lacks a required nccl* symbol"); return K5_ERR_STATE;
    }
    return K5_OK;
  }
  // in-place all-gather: every rank's chunk already sits at buf + rank*count*elem
  int all_gather_inplace(void* buf, size_t count_per_rank, size_t elem_bytes, hipStream_t s) {
    char* b = (char*)buf;
    if (ipc) return ipc_status(ipc->all_gather_inplace(buf, count_per_rank * elem_bytes, s));
    if (loop) {
      // pull every peer's chunk out of the peer's copy of the buffer; the call completes (stream-wise) only when every peer
      // has pulled mine, as an RCCL all-gather does — the caller may overwrite its own slot afterwards
      const size_t chunk = count_per_rank * elem_bytes;
      loop->ptr[rank] = buf;
      HIPCHK(hipEventRecord(loop->ready[rank], s));
      pthread_barrier_wait(&loop->bar);
      for (int p = 0; p < world; ++p) {
        if (p == rank) continue;
        HIPCHK(hipStreamWaitEvent(s, loop->ready[p], 0));
        HIPCHK(hipMemcpyAsync(b + (size_t)p * chunk, (const char*)loop->ptr[p] + (size_t)p * chunk, chunk, hipMemcpyDeviceToDevice, s));
      }
      HIPCHK(hipEventRecord(loop->pulled[rank], s));
      pthread_barrier_wait(&loop->bar);
      for (int p = 0; p < world; ++p)
        if (p != rank) HIPCHK(hipStreamWaitEvent(s, loop->pulled[p], 0));
      return K5_OK;
    }
    const ncclResult_t r = AllGather(b + (size_t)rank * count_per_rank * elem_bytes, b, count_per_rank * elem_bytes,
                                     ncclUint8, comm, s);
    if (r != ncclSuccess) { k5_set_error("ncclAllGather: %s", GetErrorString(r)); return K5_ERR_HIP; }
    return K5_OK;
  }
  bool can_exchange() const { return loop || ipc || (Send && Recv && GroupStart && GroupEnd); }
  // all-to-all (Ulysses): block p of `send` (block_bytes each) goes to rank p, block p of `recv` comes from rank p; send != recv.
  // One grouped send/recv per peer (all xGMI links at once); the rank's own block is a device-to-device copy.
  int all_to_all(const void* send, void* recv, size_t block_bytes, hipStream_t s) {
    const char* sb = (const char*)send; char* rb = (char*)recv;
    HIPCHK(hipMemcpyAsync(rb + (size_t)rank * block_bytes, sb + (size_t)rank * block_bytes, block_bytes, hipMemcpyDeviceToDevice, s));
    if (world == 1) return K5_OK;
    if (ipc) return ipc_status(ipc->all_to_all(send, recv, block_bytes, s));
    if (loop) {
      loop->ptr[rank] = const_cast<void*>(send);
      HIPCHK(hipEventRecord(loop->ready[rank], s));
      pthread_barrier_wait(&loop->bar);
      for (int p = 0; p < world; ++p) {
        if (p == rank) continue;
        HIPCHK(hipStreamWaitEvent(s, loop->ready[p], 0));
        HIPCHK(hipMemcpyAsync(rb + (size_t)p * block_bytes, (const char*)loop->ptr[p] + (size_t)rank * block_bytes, block_bytes, hipMemcpyDeviceToDevice, s));
      }
      HIPCHK(hipEventRecord(loop->pulled[rank], s));
      pthread_barrier_wait(&loop->bar);
      for (int p = 0; p < world; ++p)
        if (p != rank) HIPCHK(hipStreamWaitEvent(s, loop->pulled[p], 0));   // the peers have read my send buffer: it may be rewritten
      return K5_OK;
    }
    if (!can_exchange()) { k5_set_error("RCCL library lacks ncclSend / ncclRecv / ncclGroup*"); return K5_ERR_STATE; }
    ncclResult_t r = GroupStart();
    for (int p = 0; p < world && r == ncclSuccess; ++p) {
      if (p == rank) continue;
      r = Send(sb + (size_t)p * block_bytes, block_bytes, ncclUint8, p, comm, s);
      if (r == ncclSuccess) r = Recv(rb + (size_t)p * block_bytes, block_bytes, ncclUint8, p, comm, s);
    }
    const ncclResult_t e = GroupEnd();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) { k5_set_error("ncclSend/ncclRecv group (all-to-all): %s", GetErrorString(r)); return K5_ERR_HIP; }
    return K5_OK;
  }
  // part of an in-place all-gather: the bytes [off, off + cnt) of every rank's slot (slot_bytes each, rank p's at p * slot_bytes)
  // travel to every peer — one grouped send/recv per peer, i.e. all seven xGMI links of the GPU at once, so the first slice of ALL
  // peers has landed when a fraction cnt / slot_bytes of the gather time has passed (a ring all-gather completes nothing early).
  int slot_exchange(void* buf, size_t slot_bytes, size_t off, size_t cnt, hipStream_t s) {
    char* b = (char*)buf;
    if (cnt == 0 || world == 1) return K5_OK;
    if (ipc) return ipc_status(ipc->slot_exchange(buf, slot_bytes, off, cnt, s));
    if (loop) {
      loop->ptr[rank] = buf;
      HIPCHK(hipEventRecord(loop->ready[rank], s));
      pthread_barrier_wait(&loop->bar);
      for (int p = 0; p < world; ++p) {
        if (p == rank) continue;
        HIPCHK(hipStreamWaitEvent(s, loop->ready[p], 0));
        HIPCHK(hipMemcpyAsync(b + (size_t)p * slot_bytes + off, (const char*)loop->ptr[p] + (size_t)p * slot_bytes + off, cnt, hipMemcpyDeviceToDevice, s));
      }
      HIPCHK(hipEventRecord(loop->pulled[rank], s));
      pthread_barrier_wait(&loop->bar);
      for (int p = 0; p < world; ++p)
        if (p != rank) HIPCHK(hipStreamWaitEvent(s, loop->pulled[p], 0));
      return K5_OK;
    }
    if (!can_exchange()) { k5_set_error("RCCL library lacks ncclSend / ncclRecv / ncclGroup*"); return K5_ERR_STATE; }
    ncclResult_t r = GroupStart();
    for (int p = 0; p < world && r == ncclSuccess; ++p) {
      if (p == rank) continue;
      r = Send(b + (size_t)rank * slot_bytes + off, cnt, ncclUint8, p, comm, s);
      if (r == ncclSuccess) r = Recv(b + (size_t)p * slot_bytes + off, cnt, ncclUint8, p, comm, s);
    }
    const ncclResult_t e = GroupEnd();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) { k5_set_error("ncclSend/ncclRecv group: %s", GetErrorString(r)); return K5_ERR_HIP; }
    return K5_OK;
  }
};

}  // namespace

struct k5_dit {
  k5_dit_config cfg{};
  int D = 0, FF = 0, TD = 0, Hh = 0, Kvis = 0, KvisPad = 0, Fout = 0;
  bool finalized = false;
  std::map<std::string, HostTensor> staged;
  std::vector<std::string> expected;

  // packed weights
  DevBuf time_w1, time_b1, time_w2, time_b2;                   // fp32
  DevBuf text_w, text_b, text_lnw, text_lnb;                   // bf16 W, fp32 rest
  DevBuf pool_w, pool_b, pool_lnw, pool_lnb;
  DevBuf vis_w, vis_b, out_w, out_b;
  DevBuf mod_w, mod_b;                                         // all Modulation layers stacked, fp32
  size_t mod_rows = 0, out_mod_off = 0;
  std::vector<BlockW> tblocks, vblocks;

  // workspaces
  DevBuf ws_text_in, ws_text, ws_th, ws_tqk, ws_tvt, ws_to, ws_tff;
  DevBuf ws_pool_in, ws_pool_lin, ws_pool_f32, ws_time, ws_tfeat, ws_th1, ws_temb, ws_mod;
  DevBuf ws_xp, ws_vis, ws_h, ws_qk, ws_vt, ws_o, ws_ff, ws_ck, ws_cvt, ws_y;
  DevBuf ws_vcos, ws_vsin, ws_pos;
  DevBuf ws_vel_c, ws_vel_u;
  // sequence parallelism (token shards): this rank owns rows [sp_rank*n_loc, (sp_rank+1)*n_loc) of the N visual
  // tokens; K and V^T of every block are all-gathered in place into ws_kfull [N][D] / ws_vtfull [P][D][n_loc]
  Comm comm;
  int sp_rank = 0, sp_world = 1;
  DevBuf ws_q, ws_kfull, ws_vtfull, ws_attn_state;
  // CFG-parallel (reference semantics generation_utils.py:53-76): this handle runs ONE branch of classifier-free
  // guidance — 0 = conditional, 1 = unconditional — and is paired with the handle that runs the other one (a 2-rank communicator of
  // its own, next to the sequence-parallel one).  Inside k5_sample the pair exchanges the bf16 velocities (one in-place all-gather of
  // 2 x T H W 16 x 2 B per step on a side stream) and both apply the identical bf16 combine + fp32 Euler update.
  Comm pair;
  int cfg_branch = -1;
  hipStream_t pair_stream = nullptr;
  hipEvent_t ev_vel_ready = nullptr, ev_vel_done = nullptr;
  DevBuf ws_vel_pair;                              // [2][T H W 16] bf16: slot 0 = conditional, 1 = unconditional velocity
  bool use_fp8 = false;                            // visual feed-forward GEMMs in W8A8 e4m3 (k5_dit_set_fp8 bit 0)
  int fp8_fuse_ln = 1;                             // fp8 modes: the LayerNorm in front of an fp8 projection writes e4m3 directly (same bits as LayerNorm + quantisation pass)
  int fp8_mask = 0;                                // k5_dit_set_fp8: bit 0 feed-forward, bit 1 q | k | V^T projections, bit 2 out projection of the visual self-attention
  DevBuf ws_h8, ws_ff8;                            // fp8 activations of that path
  // Cross-attention keys / values of ALL visual blocks in one go (round 4, "cross_kv_batched"): they depend on the text stream only, so the 2 x
  // num_visual_blocks projections of 256 text rows (34 + 32 us each: 28-tile launches) become two GEMMs against the stacked weights
  // [blocks * D][D] before the visual stack, the key RMSNorms one launch — same kernels on the same per-element sums: bit-identical
  DevBuf cx_wk_all, cx_wv_all, cx_bk_all, cx_bv_all, cx_knorm_all, ws_ck_all, ws_cvt_all;
  int cross_kv_batched = 1;
  DevBuf ws_sched;                                 // sampler tables on the device: t*1000 [steps] | dt [steps] | step counter
  bool use_graph = false;                          // k5_sample replays one captured step (k5_dit_set_graph)
  hipStream_t graph_stream = nullptr;              // capture needs a real stream: the caller's may be the legacy null stream
  hipEvent_t ev_graph = nullptr;
  DevBuf ws_kc;                                    // NABLA: keys pre-multiplied by the softmax scale (separate from the map's keys);
                                                   // under sequence parallelism the reverse: the rank's UNSCALED keys (the scaled ones are gathered)
  DevBuf ws_kmeans;                                // NABLA under sequence parallelism: gathered key-block means [P][H][slot blocks][64]
  DevBuf ws_attn_bal;                              // states of the split tail jobs (k5_launch_attention_bf16_range, balanced)
  // data-derived softmax bound of the visual self-attention (pre-scaled keys): per-head max |q|^2 [Hh] | max |k'|^2 [P][Hh]
  // (fp32, filled by the rmsnorm/RoPE kernel, consumed by k5_launch_attn_flags), the per-head variant flags [Hh] (int) and
  // two u64 counters: heads that ran the fixed-offset / the online-max kernel since the last reset
  DevBuf ws_attn_stats, ws_attn_flags, ws_attn_cnt, ws_attn_part;   // ws_attn_part: per-block partial maxima of the norm kernel
  long long nabla_possible = 0;                    // profiling: 64x64 blocks the NABLA maps could have kept (kept: ws_attn_cnt[2])
  int nabla_list_rows = 4;                         // rows per key-tile list of the last profiled map (ws_attn_cnt[3] = sum of list lengths)
  int attn_mode = 0;                               // K5_ATTN_AUTO / K5_ATTN_ONLINE (k5_dit_set_option "attn_mode")
  int sp_pass1_tiles = 0;                          // k5_dit_set_option "sp_pass1_tiles" (0 = all local key tiles)
  // "nabla_group_rows": 64-query rows per key-tile list = per attention workgroup (one GPU).  4: 256-query workgroups; 2: 128-query
  // workgroups (tighter lists: -17 % attention at kept density 0.05, -8 % at 0.12, +3 % at 0.81; bit-identical results); 0 (default):
  // chosen per forward from the kept density of the PREVIOUS forward's first map (counted on the device, copied to pinned host memory
  // without a synchronisation: a stale or missing value only costs speed)
  int nabla_group_rows = 0, nabla_grp_now = 4;
  // "nabla_pair_frames" (round 4): the two 64-query rows of a 128-query list are the same spatial tile in ADJACENT FRAMES (k5_pair_row) instead
  // of adjacent tiles of one frame: their sliding-tile windows share 10 of 11 frames.  Same arithmetic per row (a row's key tiles are walked
  // in the same ascending order), so outputs are bit-identical with it on or off.
  int nabla_pair_frames = 1;
  void* nabla_tap = nullptr; long long nabla_tap_cap = 0, nabla_tap_n = 0;   // k5_dit_set_nabla_tap: caller's device buffer, its size, maps taken so far
  int nabla_fuse_means = 1;   // the q / k block means come out of the norm + RoPE pass (k5_launch_rmsnorm_rope mean_q / mean_k); 0: their own pass (rounds 1-3)
  DevBuf ws_nabla_kept;                            // u64 kept-block count of the forward's first NABLA map
  unsigned long long* h_nabla_kept = nullptr;      // its pinned host copy
  long long nabla_hint_possible = 0;               // blocks that map could have kept
  bool nabla_hint_pending = false;
  int sp_nabla_passes = 1;                         // "sp_nabla_passes" = 2: NABLA under sequence parallelism attends the rank's own key blocks during the gather
  int fuse_qnorm = 0;                              // "attn_fuse_qnorm": norm_qk + RoPE of the visual queries inside the attention kernel
  // "attn_fuse_qnorm_auto" (default 1): k5_sample decides per CALL.  Step 0 runs with the standalone norm pass and collects whether any head left
  // the plain fixed-offset form (online, anchored, or marked as falling back: leave_sig); if none did, steps 1.. normalise the queries inside the
  // attention kernel (fuse_now) — the same bits (both kernels spell the norm + rotation out operation for operation), half of the norm pass less per
  // block and a faster attention instantiation.  Heads that
  // leave the window keep today's routing (anchored offsets need the normalised queries in memory), so such calls stay as they were.
  int fuse_qnorm_auto = 1;
  bool fuse_now = false, leave_collect = false;
  int fuse_last = -1;                              // "attn_fuse_qnorm_used": what the last k5_sample decided (1 fused from step 1 on, 0 not, -1 not evaluated)
  DevBuf ws_leave_sig;
  unsigned int* h_leave_sig = nullptr;             // pinned
  bool row_offsets = true;                         // "attn_row_offsets": per-row offsets of the fixed-offset softmax (bound up to 190)
  bool anchor = true;                              // "attn_anchor": heads beyond that window run the fixed form on anchored offsets (one-GPU path)
  DevBuf ws_attn_anchor;                           // [H][rows] anchored offsets (k5_launch_attn_row_anchor)
  int sp_slices = 1;                               // "sp_slices": the K / V^T exchange of a block in this many slices (dense attention)
  int sp_mode = 0;                                 // "sp_mode": 0 = K / V^T all-gather (any rank count), 1 = Ulysses all-to-all (heads % ranks == 0, dense attention)
  DevBuf ws_u_send, ws_u_recv, ws_u_vsend, ws_u_vrecv, ws_u_o, ws_u_orecv, ws_u_stats;   // Ulysses exchange buffers
  hipEvent_t ev_u_o = nullptr, ev_u_back = nullptr;
  hipEvent_t ev_slice[4] = {};                     // slice s of every peer has landed
  bool emulated = false;                           // "emulate_world": timing-only layout, results are garbage
  bool last_nabla = false;                         // attention type of the last forward ("fp8_effective": Ulysses, which keeps q | k | V^T in bf16, is a dense-attention schedule)
  // Self-tuning sequence-parallel schedule (round 4): which exchange wins — one in-place all-gather per block, the sliced exchange, Ulysses
  // all-to-all; for NABLA one or two passes over the lists — is a property of the NODE (xGMI link rates, how RCCL drives them) that no
  // single-GPU box can measure.  So the first sharded forward of a handle (world > 1) times one block's self-attention section under
  // every admissible candidate on its own shapes, the ranks agree on max-over-ranks per candidate, and the fastest is kept for the life
  // of the handle.  Knobs the caller set explicitly are left alone, and only the knobs a run VARIED are assigned from its winner.
  // OPT-IN since round 5 ("sp_autotune" = 1, K5_SP_AUTOTUNE=1, bench.py --sp-autotune): the winner comes from wall-clock timings and the schedules
  // sum in different orders (one or two NABLA passes, gather / sliced exchange / Ulysses), so with it on the same seed may give different bits on
  // two nodes or two runs; off, a handle always runs the all-gather schedule unless told otherwise.
  bool sp_autotune = false, sp_tuned = false;
  int sp_user_set = 0;                             // bit 0: sp_mode, 1: sp_slices, 2: sp_nabla_passes were set through k5_dit_set_option
  std::string sp_report;                           // JSON text of the last tuning run (k5_dit_sp_schedule)
  DevBuf ws_tune;
  hipStream_t comm_stream = nullptr;              // all-gathers run here, overlapped with pass 1 of the attention
  hipEvent_t ev_k = nullptr, ev_v = nullptr, ev_gathered = nullptr, ev_stats = nullptr, ev_means = nullptr;
  // NABLA: fractal token permutation (cached per shape) and the selection workspace
  DevBuf ws_perm, ws_nabla; int perm_shape[3] = {0, 0, 0}; bool key_fractal = false;
  // rope cache keys
  std::vector<int32_t> key_vpos; float key_scale[3] = {0, 0, 0}; int key_shape[3] = {0, 0, 0};
  struct TextRope { std::vector<int32_t> key; DevBuf cosT, sinT, pos; };
  std::vector<TextRope> text_rope;  // small cache: cond / null-cond position vectors

  // Step-invariant text prologue: TextEmbeddings(text) and the pooled projection (dit.py:132,134) depend on the
  // prompt only, so k5_sample computes them once per call and branch (slot 0 = cond, 1 = uncond) and every later step copies
  // 0.9 MB instead of re-running two GEMMs + two LayerNorms.  k5_dit_forward (one call per step, caller-owned buffers that
  // may change between calls) does not cache.
  struct TextCache { DevBuf text, pool; bool valid = false; int L = 0; } text_cache[2];

  // MagCache (reference kandinsky/magcache_utils.py:16-101): skip the visual blocks on some calls and re-apply the
  // cached bf16 residual of the same cond / uncond slot.  Decisions depend on the ratio table and the call counter only.
  struct MagCache {
    bool on = false, no_cfg = false;
    std::vector<double> table;                     // [2 * num_steps], already interpolated by the host mirror
    double thresh = 0.12, retention = 0.2;
    int K = 2, cnt = 0;
    int first = 0, stride = 1;                     // which calls of the reference's sequence this handle sees
    double acc_err[2] = {0, 0}, acc_ratio[2] = {1, 1};
    int acc_steps[2] = {0, 0};
    DevBuf residual[2]; size_t res_elems[2] = {0, 0};
    DevBuf pm_one;                                 // fp32 [+1 x D | -1 x D]
    long long n_ran = 0, n_skipped = 0;
  } mag;

  // profiling
  int profiling = 0;                               // 0 off, 1 every kernel family, 2 only the visual self-attention (the roofline kernel)
  std::map<std::string, Prof> prof;
  long long prof_self_blocks = 0;                  // visual blocks whose self-attention ran while profiling was on ("self_blocks")
  struct Pending { std::string fam; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> ev_pool;

  ~k5_dit() {
    for (auto& e : ev_pool) (void)hipEventDestroy(e);
    for (auto& pnd : pending) { (void)hipEventDestroy(pnd.a); (void)hipEventDestroy(pnd.b); }
  }
};

namespace {

// ---------------------------------------------------------------------------------------------
// profiling helpers: bracket a kernel family with events on the engine stream
// ---------------------------------------------------------------------------------------------
hipEvent_t get_event(k5_dit* d) {
  if (!d->ev_pool.empty()) { hipEvent_t e = d->ev_pool.back(); d->ev_pool.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}
struct Scope {
  k5_dit* d; hipStream_t s; const char* fam; hipEvent_t a{}, b{}; bool on;
  Scope(k5_dit* d_, hipStream_t s_, const char* f) : d(d_), s(s_), fam(f), on(d_->profiling == 1 || (d_->profiling == 2 && !strcmp(f, "attn_self"))) {
    if (on) { a = get_event(d); b = get_event(d); (void)hipEventRecord(a, s); }
  }
  ~Scope() { if (on) { (void)hipEventRecord(b, s); d->pending.push_back({fam, a, b}); } }
};
void drain_profile(k5_dit* d) {
  for (auto& p : d->pending) {
    (void)hipEventSynchronize(p.b);
    float ms = 0; (void)hipEventElapsedTime(&ms, p.a, p.b);
    auto& e = d->prof[p.fam]; e.ms += ms; e.n += 1;
    d->ev_pool.push_back(p.a); d->ev_pool.push_back(p.b);
  }
  d->pending.clear();
}

// ---------------------------------------------------------------------------------------------
// expected state_dict keys (dit.py:100-127)
// ---------------------------------------------------------------------------------------------
void expected_keys(const k5_dit_config& c, std::vector<std::string>& out) {
  auto lin = [&](const std::string& n, bool bias = true) {
    out.push_back(n + ".weight");
    if (bias) out.push_back(n + ".bias");
  };
  lin("time_embeddings.in_layer"); lin("time_embeddings.out_layer");
  lin("text_embeddings.in_layer"); out.push_back("text_embeddings.norm.weight"); out.push_back("text_embeddings.norm.bias");
  lin("pooled_text_embeddings.in_layer"); out.push_back("pooled_text_embeddings.norm.weight"); out.push_back("pooled_text_embeddings.norm.bias");
  lin("visual_embeddings.in_layer");
  auto attn = [&](const std::string& p) {
    lin(p + ".to_query"); lin(p + ".to_key"); lin(p + ".to_value");
    out.push_back(p + ".query_norm.weight"); out.push_back(p + ".key_norm.weight");
    lin(p + ".out_layer");
  };
  for (int i = 0; i < c.num_text_blocks; ++i) {
    const std::string p = "text_transformer_blocks." + std::to_string(i);
    lin(p + ".text_modulation.out_layer"); attn(p + ".self_attention");
    lin(p + ".feed_forward.in_layer", false); lin(p + ".feed_forward.out_layer", false);
  }
  for (int i = 0; i < c.num_visual_blocks; ++i) {
    const std::string p = "visual_transformer_blocks." + std::to_string(i);
    lin(p + ".visual_modulation.out_layer"); attn(p + ".self_attention"); attn(p + ".cross_attention");
    lin(p + ".feed_forward.in_layer", false); lin(p + ".feed_forward.out_layer", false);
  }
  lin("out_layer.modulation.out_layer"); lin("out_layer.out_layer");
}

// upload helpers ---------------------------------------------------------------------------
int upload_f32(DevBuf& b, const float* src, size_t n) {
  K5CHK(b.ensure(n * 4));
  HIPCHK(hipMemcpy(b.p, src, n * 4, hipMemcpyHostToDevice));
  return K5_OK;
}
int upload_bias_bf16r(DevBuf& b, const float* src, size_t n) {  // fp32 array of bf16-rounded values
  std::vector<float> tmp(n);
  for (size_t i = 0; i < n; ++i) tmp[i] = bf16_round_host(src[i]);
  return upload_f32(b, tmp.data(), n);
}

// staged matrix -> rows [row0, row0 + rows) of dst [total_rows][ld] (bf16 or fp32), on the device
int pack_rows(DevBuf& dst, bool bf16, const HostTensor* t, size_t rows, size_t cols, size_t ld, size_t row0 = 0, size_t total_rows = 0) {
  if (!t || !t->dev.p || t->count() != rows * cols) { k5_set_error("internal: matrix not staged on the device"); return K5_ERR_STATE; }
  if (total_rows == 0) total_rows = rows;
  const size_t es = bf16 ? 2 : 4;
  K5CHK(dst.ensure(total_rows * ld * es));
  return k5_launch_pack_matrix(t->dev.p, t->dtype, (char*)dst.p + row0 * ld * es, bf16 ? 1 : 0, (int64_t)rows, (int)cols, (int)ld, nullptr);
}
int pack_bf16(DevBuf& dst, const HostTensor* t, size_t rows, size_t cols, size_t ld, size_t row0 = 0, size_t total_rows = 0) {
  return pack_rows(dst, true, t, rows, cols, ld, row0, total_rows);
}
int pack_f32(DevBuf& dst, const HostTensor* t, size_t rows, size_t cols, size_t row0 = 0, size_t total_rows = 0) {
  return pack_rows(dst, false, t, rows, cols, cols, row0, total_rows);
}

const HostTensor* find(k5_dit* d, const std::string& k) {
  auto it = d->staged.find(k);
  return it == d->staged.end() ? nullptr : &it->second;
}

// shape check of one staged tensor: [r][c], or [r] when c == 0 (a mismatched tensor handed through the C ABI must come back as
// K5_ERR_ARG, not as an out-of-bounds host read while packing)
bool shape_ok(k5_dit* d, const std::string& k, size_t r, size_t c) {
  const HostTensor* t = find(d, k);
  const bool ok = t && ((c == 0 && t->shape.size() == 1 && (size_t)t->shape[0] == r) ||
                        (c && t->shape.size() == 2 && (size_t)t->shape[0] == r && (size_t)t->shape[1] == c));
  if (!ok) k5_set_error("shape mismatch for %s", k.c_str());
  return ok;
}

int pack_attn(k5_dit* d, const std::string& p, AttnW& a, bool fuse_qk) {
  const size_t D = d->D;
  for (const char* nm : {".to_query", ".to_key", ".to_value", ".out_layer"})
    if (!shape_ok(d, p + nm + ".weight", D, D) || !shape_ok(d, p + nm + ".bias", D, 0)) return K5_ERR_ARG;
  if (!shape_ok(d, p + ".query_norm.weight", 64, 0) || !shape_ok(d, p + ".key_norm.weight", 64, 0)) return K5_ERR_ARG;
  const HostTensor *wq = find(d, p + ".to_query.weight"), *wk = find(d, p + ".to_key.weight"),
                   *wv = find(d, p + ".to_value.weight"), *wo = find(d, p + ".out_layer.weight"),
                   *bq = find(d, p + ".to_query.bias"), *bk = find(d, p + ".to_key.bias"),
                   *bv = find(d, p + ".to_value.bias"), *bo = find(d, p + ".out_layer.bias"),
                   *nq = find(d, p + ".query_norm.weight"), *nk = find(d, p + ".key_norm.weight");
  if (fuse_qk) {
    std::vector<float> b(2 * D);
    memcpy(b.data(), bq->data.data(), D * 4); memcpy(b.data() + D, bk->data.data(), D * 4);
    K5CHK(pack_bf16(a.wqk, wq, D, D, D, 0, 2 * D)); K5CHK(pack_bf16(a.wqk, wk, D, D, D, D, 2 * D));
    K5CHK(upload_bias_bf16r(a.bqk, b.data(), 2 * D));
  } else {
    K5CHK(pack_bf16(a.wq, wq, D, D, D)); K5CHK(upload_bias_bf16r(a.bq, bq->data.data(), D));
    K5CHK(pack_bf16(a.wk, wk, D, D, D)); K5CHK(upload_bias_bf16r(a.bk, bk->data.data(), D));
  }
  K5CHK(pack_bf16(a.wv, wv, D, D, D)); K5CHK(upload_bias_bf16r(a.bv, bv->data.data(), D));
  K5CHK(pack_bf16(a.wo, wo, D, D, D)); K5CHK(upload_bias_bf16r(a.bo, bo->data.data(), D));
  std::vector<float> n(128);
  memcpy(n.data(), nq->data.data(), 64 * 4); memcpy(n.data() + 64, nk->data.data(), 64 * 4);
  K5CHK(upload_f32(a.norm, n.data(), 128));
  float mq = 0.f, mk = 0.f;
  for (int i = 0; i < 64; ++i) { mq = fmaxf(mq, fabsf(n[i])); mk = fmaxf(mk, fabsf(n[64 + i])); }
  a.score_bound = 64.f * mq * mk * 1.05f;  // 5 % margin for the bf16 roundings after RMSNorm / RoPE
  return K5_OK;
}

inline size_t rup(size_t v, size_t a) { return (v + a - 1) / a * a; }

// The visual self-attention runs with the keys pre-multiplied by log2(e)/8 (attn_fwd_kernel<.., PRE>: the scores are the exp2
// arguments) whenever the rows come in whole 64-key tiles.  Which softmax form a head takes — constant offset 0, or the lazy
// online max — is decided ON THE DEVICE from the data: the rmsnorm/RoPE kernel leaves max |q_h|^2 and max |k'_h|^2, and
// k5_launch_attn_flags turns |q|max |k'|max <= 90 into the per-head flag both attention launches read.  (Round 1 derived
// the bound from the QK-norm weights, 64 max|w_q| max|w_k|, which no trained checkpoint is known to satisfy.)
// Text and cross attention (tiny) keep the weight-derived bound + the 32x32 online kernel as their fallback.
constexpr float K5_SOFTMAX_C = 0.125f * 1.44269504088896340736f;

int ensure_zeroed(DevBuf& b, size_t n, hipStream_t s) {
  if (n <= b.bytes) return K5_OK;
  K5CHK(b.ensure(n));
  HIPCHK(hipMemsetAsync(b.p, 0, b.bytes, s));
  return K5_OK;
}
// statistics / flags / counters of the data-derived softmax bound; layout of ws_attn_stats: [q: Hh][k: sp_world x Hh]
int ensure_attn_flags(k5_dit* d, hipStream_t s) {
  K5CHK(ensure_zeroed(d->ws_attn_stats, (size_t)d->Hh * (2 + d->sp_world) * 4, s));   // + [r: Hh] squared key radii (one GPU: centred offsets)
  K5CHK(d->ws_attn_flags.ensure((size_t)d->Hh * (3 + 64) * 4));   // int flags[H] | float kmax[H] | krad[H] | centre[H][64] (per-row offsets of the fixed-offset form)
  K5CHK(d->ws_attn_part.ensure(k5_rmsnorm_stats_workspace_bytes(2 * d->Hh)));
  K5CHK(ensure_zeroed(d->ws_attn_cnt, 32, s));
  return K5_OK;
}

// ---------------------------------------------------------------------------------------------
// one attention module on `rows` tokens:  x_resid += gate * out_l(attn(...)) fused in the out GEMM
// ---------------------------------------------------------------------------------------------
struct NablaArgs { int T, Hb, Wb, wT, wH, wW; float P; };

// the forward's first NABLA map (nqb query-block rows x nb key blocks per head, in ws_nabla): its kept density steers the NEXT forward's
// attention workgroup size (nabla_group_rows = 0)
int nabla_density_hint(k5_dit* d, int H, int nqb, int nb, hipStream_t s) {
  if (!d->nabla_hint_pending || d->nabla_group_rows != 0) return K5_OK;
  d->nabla_hint_pending = false;
  K5CHK(d->ws_nabla_kept.ensure(8));
  if (!d->h_nabla_kept) { HIPCHK(hipHostMalloc((void**)&d->h_nabla_kept, 8, hipHostMallocDefault)); *d->h_nabla_kept = 0ull; }
  HIPCHK(hipMemsetAsync(d->ws_nabla_kept.p, 0, 8, s));
  K5CHK(k5_launch_nabla_count(d->ws_nabla.p, H, nqb, nb, d->ws_nabla_kept.as<unsigned long long>(), s));
  HIPCHK(hipMemcpyAsync(d->h_nabla_kept, d->ws_nabla_kept.p, 8, hipMemcpyDeviceToHost, s));
  d->nabla_hint_possible = (long long)H * nqb * nb;
  return K5_OK;
}

// do the q | k | V^T projections of the VISUAL self-attention take the e4m3 activations (k5_dit_set_fp8 bit 1)?  One rule for the callee and for
// the block loop, which then lets the LayerNorm write ws_h8 directly (h8_ready) instead of a bf16 h + a quantisation pass
bool sa_fp8_in(const k5_dit* d, const AttnW& a, int rows) {
  return (d->fp8_mask & 2) && a.wqk8.p && rows >= 256 && !((int)rup(rows, 8) & 15);
}
bool sa_sp_fp8_in(const k5_dit* d, const AttnW& a, int rows, int rows_pad, bool nabla) {
  const int P = d->sp_world;
  return (d->fp8_mask & 2) && a.wqk8.p && rows >= 256 && !((rows_pad / ((!nabla && d->sp_slices > 1 && P > 1) ? d->sp_slices : 1)) & 15);
}
// "nabla_fuse_means": the fused form gives every thread 64 consecutive rows of one 16-byte column chunk, i.e. the launch has rows / 64 x heads x 8
// threads — a sequence-parallel shard of the 10 s clip (366 blocks x 28 heads: 82 k threads, a sixth of the resident capacity) runs it latency-
// bound and loses 3 ms per step to gain 1 (measured, incl. an 8-deep prefetch ring); from ~150 k threads up it wins.  2 = always.
// bytes of the NABLA workspace for nqb selected rows of nb: the lists are per group of >= 2 rows unless "nabla_group_rows" = 1 was asked for
size_t nabla_ws_bytes(const k5_dit* d, int H, int nb, int nqb) {
  const int gmin = d->nabla_group_rows == 1 ? 1 : 2;
  return k5_nabla_workspace_bytes(H, nb, nqb, (nqb + gmin - 1) / gmin);
}
bool nabla_means_fused(const k5_dit* d, int rows, int heads) {
  return d->nabla_fuse_means == 2 || (d->nabla_fuse_means == 1 && (long long)(rows / 64) * heads * 8 >= 150000);
}
bool ff_fp8_in(const k5_dit* d, const BlockW& b, int rows) { return d->use_fp8 && b.w1_f8.p && rows >= 256; }

int run_self_attention(k5_dit* d, hipStream_t s, const AttnW& a, const void* h, int rows, void* qk, void* vt,
                       void* o, const float* cosT, const float* sinT, void* resid, const float* gate,
                       const char* fam_attn, const NablaArgs* nabla = nullptr, int pref_slot = 0, bool h8_ready = false) {
  const int D = d->D, H = d->Hh;
  const int ldvt = (int)rup(rows, 8);
  const bool vis = !strcmp(fam_attn, "attn_self");
  const bool f8_in = vis && sa_fp8_in(d, a, rows);   // opt-in lossy mode: e4m3 projections (gemm_fp8.hip)
  const bool f8_out = vis && (d->fp8_mask & 4) && a.wo8.p && rows >= 256;
  if (h8_ready && !f8_in) { k5_set_error("internal: e4m3 activations handed to a bf16 projection"); return K5_ERR_ARG; }
  if (f8_in) {
    K5CHK(d->ws_h8.ensure((size_t)rows * D));
    if (!h8_ready) {
      Scope sc(d, s, "elementwise");
      K5CHK(k5_launch_quant_rows_fp8(h, d->ws_h8.p, nullptr, rows, D, D, D, s));
    }
    Scope sc(d, s, "gemm");
    K5CHK(k5_launch_gemm_fp8(d->ws_h8.p, a.wqk8.p, a.sqk.as<float>(), qk, rows, 2 * D, D, D, D, 2 * D, K5_EPI_BIAS, nullptr, 0, nullptr, s, a.bqk.as<float>(), 0));
    K5CHK(k5_launch_gemm_fp8(a.wv8.p, d->ws_h8.p, a.sv.as<float>(), vt, D, rows, D, D, D, ldvt, K5_EPI_BIAS, nullptr, 0, nullptr, s, a.bv.as<float>(), 1));
  } else {
    Scope sc(d, s, "gemm");
    K5CHK(k5_launch_gemm_bf16(h, a.wqk.p, a.bqk.as<float>(), qk, rows, 2 * D, D, D, D, 2 * D, K5_EPI_BIAS, nullptr, 0, nullptr, s));
    K5CHK(k5_launch_gemm_bf16(a.wv.p, h, a.bv.as<float>(), vt, D, rows, D, D, D, ldvt, K5_EPI_BIAS_M, nullptr, 0, nullptr, s));
  }
  const bool pre = vis && rows % 64 == 0;   // visual blocks only (not the text blocks)
  const bool by_data = pre && d->attn_mode == K5_ATTN_AUTO;            // per-head flags from the data
  const int* hflags = nullptr;
  // per-row softmax offsets: heads with a Cauchy-Schwarz bound up to 190 keep the fixed-offset kernel, each query row on its own
  // constant offset |q| max|k'| - 90
  const float* kmax = nullptr;
  K5KeyCentre kcen{nullptr, nullptr};
  const K5KeyCentre* kcp = nullptr;
  // dense visual blocks: norm_qk + RoPE of the queries happen in the attention kernel's Q load ("attn_fuse_qnorm")
  const bool fuse_q = pre && !nabla && (d->fuse_qnorm || d->fuse_now) && ((by_data && d->row_offsets) || d->attn_mode == K5_ATTN_ONLINE);
  unsigned int* leave_sig = (d->leave_collect && d->ws_leave_sig.p) ? d->ws_leave_sig.as<unsigned int>() : nullptr;
  if (by_data) K5CHK(ensure_attn_flags(d, s));   // before the counters' address is taken
  const K5QueryNorm qn{a.norm.as<float>(), cosT, sinT, by_data ? d->ws_attn_cnt.as<unsigned long long>() : nullptr};
  {
    Scope sc(d, s, "elementwise");
    const int32_t hc[2] = {H, 2 * H};
    // NABLA: the block map needs the 64-token means of the UNSCALED keys (and queries).  Round 4 ("nabla_fuse_means", default): the norm pass
    // takes them itself, so the keys are scaled in place as on the dense path — no second read of q | k, no scaled copy.  Otherwise (rounds
    // 1-3) the unscaled keys stay in place for k5_launch_nabla_select_rect's own means pass and the scaled copy goes to its own buffer.
    void* kc = nullptr;
    void *mq = nullptr, *mk = nullptr;
    const bool fuse_means = pre && nabla && nabla_means_fused(d, rows, 2 * H);
    if (fuse_means) {
      K5CHK(d->ws_nabla.ensure(nabla_ws_bytes(d, H, rows / 64, rows / 64)));
      k5_nabla_workspace_means(d->ws_nabla.p, H, rows / 64, &mq, &mk);
    } else if (pre && nabla) { K5CHK(d->ws_kc.ensure((size_t)rows * D * 2)); kc = d->ws_kc.p; }
    float* stats = nullptr;
    if (by_data) stats = d->ws_attn_stats.as<float>();   // [q heads | k' heads] = the call's 2H heads (| squared key radii with the centred offsets)
    // centred per-row offsets (K5KeyCentre): the keys' sample-mean centre and their radius around it come out of the same pass
    float* centre = (by_data && d->row_offsets) ? d->ws_attn_flags.as<float>() + 3 * H : nullptr;
    if (fuse_q)   // keys only; the query statistic stays 0 (the fixed-offset workgroups take the head-level decision, K5QueryNorm)
      K5CHK(k5_launch_rmsnorm_rope((bf16_t*)qk + D, a.norm.as<float>() + 64, cosT, sinT, rows, H, 2 * D, nullptr, s, K5_SOFTMAX_C, 0, nullptr, 0,
                                   stats ? stats + H : nullptr, d->ws_attn_part.as<float>(), centre));
    else
      K5CHK(k5_launch_rmsnorm_rope(qk, a.norm.as<float>(), cosT, sinT, rows, 2 * H, 2 * D, hc, s, K5_SOFTMAX_C, pre ? H : 0x7fffffff, kc, D, stats, d->ws_attn_part.as<float>(),
                                   pre ? centre : nullptr, mq, rows / 64, mk, rows / 64));
    if (by_data) {
      hflags = d->ws_attn_flags.as<int>();
      float* kmax_w = d->row_offsets ? d->ws_attn_flags.as<float>() + H : nullptr;
      kmax = kmax_w;
      if (kmax_w) K5CHK(ensure_zeroed(a.pref, (size_t)2 * H * 4, s));
      // heads beyond the Cauchy-Schwarz window: anchored offsets (needs the normalised queries in memory: not with the fused query norm)
      const bool anchored = centre && d->anchor && !fuse_q && !nabla;   // dense attention only (k5_launch_attn_row_anchor)
      K5CHK(k5_launch_attn_flags(stats, stats + H, 1, H, H, 0, d->ws_attn_flags.as<int>(), d->ws_attn_cnt.as<unsigned long long>(), s, kmax_w,
                                 kmax_w ? (a.pref.as<int>() + (size_t)pref_slot * H) : nullptr, centre ? stats + 2 * H : nullptr, centre ? kmax_w + H : nullptr, 1, 0, anchored, leave_sig));
      if (centre) { kcen.centre = centre; kcen.radius = kmax_w + H; kcp = &kcen; }
      if (anchored) {
        K5CHK(d->ws_attn_anchor.ensure((size_t)H * rows * 4));
        K5CHK(k5_launch_attn_row_anchor(qk, (const bf16_t*)qk + D, H, rows, rows, 2 * D, 2 * D, 0, rows, kmax_w, d->ws_attn_anchor.as<float>(), s));
        kcen.row_anchor = d->ws_attn_anchor.as<float>();
      }
    }
  }
  const int variant = pre ? d->attn_mode : K5_ATTN_AUTO;
  if (nabla) {  // nablaT_v2 map (utils.py:136-163) + block-sparse attention (nn.py:257-280)
    const int nb = rows / 64;
    const int grp = pre ? d->nabla_grp_now : 4;   // 64-query rows per key-tile list = per attention workgroup
    const int pair = (grp == 2 && d->nabla_pair_frames) ? nabla->Hb * nabla->Wb : 0;
    K5CHK(d->ws_nabla.ensure(nabla_ws_bytes(d, H, nb, nb)));
    {
      Scope sc(d, s, "nabla_map");
      const bool fm = pre && nabla_means_fused(d, rows, 2 * H);   // the means are in the workspace already
      K5CHK(k5_launch_nabla_select_rect(fm ? nullptr : qk, fm ? nullptr : (const bf16_t*)qk + D, 2 * D, 2 * D, H, rows, 0, rows, nabla->T, nabla->Hb, nabla->Wb, nabla->wT,
                                        nabla->wH, nabla->wW, nabla->P, d->ws_nabla.p, s, 0, 0, grp, pair));
    }
    if (d->nabla_tap) {   // diagnostics (k5_dit_set_nabla_tap): this launch's map, expanded, behind the ones already taken
      const long long sz = (long long)H * nb * nb;
      if ((d->nabla_tap_n + 1) * sz <= d->nabla_tap_cap)
        K5CHK(k5_launch_nabla_mask_u8(d->ws_nabla.p, H, nb, nb, (char*)d->nabla_tap + d->nabla_tap_n * sz, s));
      ++d->nabla_tap_n;
    }
    if (d->profiling) {   // realised density of the map (bench.py: attention FLOPs actually done) and tiles the launch executes for it
      K5CHK(ensure_zeroed(d->ws_attn_cnt, 32, s));
      K5CHK(k5_launch_nabla_count(d->ws_nabla.p, H, nb, nb, d->ws_attn_cnt.as<unsigned long long>() + 2, s));
      K5CHK(k5_launch_nabla_count_lists(d->ws_nabla.p, H, nb, nb, grp, d->ws_attn_cnt.as<unsigned long long>() + 3, s));
      d->nabla_possible += (long long)H * nb * nb;
      d->nabla_list_rows = grp;
    }
    K5CHK(nabla_density_hint(d, H, nb, nb, s));
    const int *list, *cnt;
    k5_nabla_workspace_views(d->ws_nabla.p, H, nb, nullptr, nullptr, &list, &cnt);
    K5CHK(d->ws_attn_bal.ensure(k5_attention_balance_bytes(H, rows)));   // here only for the per-job fallback flags of the per-row offsets
    Scope sc(d, s, fam_attn);
    // (no tail balancing here: 10 248 jobs are 20 rounds of unequal lists — measured -0.6 % at density 0.81, +1 % at 0.12, +2.4 % at
    // 0.05; a token shard's 5 rounds are another matter, run_self_attention_sp)
    const bool kin = !pre || nabla_means_fused(d, rows, 2 * H);   // keys in place in the fused q | k buffer (scaled there when pre)
    K5CHK(k5_launch_attention_bf16_sparse(qk, kin ? (const bf16_t*)qk + D : d->ws_kc.as<bf16_t>(), vt, o, H, rows, rows, 2 * D, kin ? 2 * D : D,
                                          ldvt, D, pre ? 0.f : a.score_bound, list, cnt, nb, 0, 0, s, pre, hflags, variant, kmax, nullptr,
                                          pre ? d->ws_attn_bal.as<float>() : nullptr, grp, false, kcp, pair));
  } else {
    K5CHK(d->ws_attn_bal.ensure(k5_attention_balance_bytes(H, rows)));
    Scope sc(d, s, fam_attn);
    K5CHK(k5_launch_attention_bf16_range(qk, (const bf16_t*)qk + D, vt, o, H, rows, rows, 2 * D, 2 * D, ldvt, D, pre ? 0.f : a.score_bound, 0, 0, 0, -1,
                                         0x7fffffff, 0, nullptr, 0, s, d->ws_attn_bal.as<float>(), pre, hflags, variant, nullptr, kmax, 0,
                                         fuse_q ? &qn : nullptr, kcp));
  }
  if (kmax) K5CHK(k5_launch_attn_pref_update(d->ws_attn_bal.as<float>(), H, rows, nabla ? (pre ? d->nabla_grp_now : 4) : 4, (a.pref.as<int>() + (size_t)pref_slot * H), s, leave_sig));
  if (f8_out) {
    K5CHK(d->ws_h8.ensure((size_t)rows * D));
    {
      Scope sc(d, s, "elementwise");
      K5CHK(k5_launch_quant_rows_fp8(o, d->ws_h8.p, nullptr, rows, D, D, D, s));
    }
    Scope sc(d, s, "gemm");
    K5CHK(k5_launch_gemm_fp8(d->ws_h8.p, a.wo8.p, a.so.as<float>(), resid, rows, D, D, D, D, D, K5_EPI_GATE, resid, D, gate, s, a.bo.as<float>(), 0));
    return K5_OK;
  }
  {
    Scope sc(d, s, "gemm");
    K5CHK(k5_launch_gemm_bf16(o, a.wo.p, a.bo.as<float>(), resid, rows, D, D, D, D, D, K5_EPI_GATE, resid, D, gate, s));
  }
  return K5_OK;
}

// Sequence-parallel visual self-attention: `rows` = this rank's token rows, N = rows * world keys in total.
// q / k / v^T are projected for the local rows only; k and v^T land directly in this rank's slot of the
// gather buffers, one in-place all-gather each makes every rank hold all keys, then attention runs for the
// local query rows.  No head-count constraint (28 heads do not divide by 8), no activation all-reduce.
// Uneven shards: every rank's slot in the gather buffers holds rows_pad = ceil(blocks / P) * 64 rows; only the LAST rank may
// own fewer (rows < rows_pad), so the padded row index of every real key equals its global index and the unused tail of the
// last slot is never read (the key-tile ranges stop at N).
int run_self_attention_sp(k5_dit* d, hipStream_t s, const AttnW& a, const void* h, int rows, int rows_pad, int N, void* o,
                          const float* cosT, const float* sinT, void* resid, const float* gate, const NablaArgs* nabla = nullptr, int pref_slot = 0,
                          bool h8_ready = false) {
  const int D = d->D, H = d->Hh, P = d->sp_world, r = d->sp_rank;
  const int ldv = rows_pad;  // rows, rows_pad, N are multiples of 64 (checked by the caller)
  bf16_t* q = d->ws_q.as<bf16_t>();
  bf16_t* kfull = d->ws_kfull.as<bf16_t>();
  bf16_t* vtfull = d->ws_vtfull.as<bf16_t>();
  bf16_t* kloc = kfull + (size_t)r * rows_pad * D;
  bf16_t* vtloc = vtfull + (size_t)r * D * ldv;
  const bf16_t* wq = a.wqk.as<bf16_t>();
  const bf16_t* wk = wq + (size_t)D * D;
  // pre-scaled keys, softmax form per head from the data (the |k'|^2 maxima of all ranks are gathered with the keys) — dense AND
  // NABLA.  The NABLA map is computed from the UNSCALED keys, of which it only needs the 64-token block means: the unscaled
  // keys stay local (ws_kc), their means (28 x 64 values per block) travel with the gather, and the gathered keys are the scaled ones —
  // the same kernels, flags and per-row offsets as on one GPU (round 1 / early round 2: unscaled keys, weight-derived bound, and
  // with it the online-max 32x32 kernel for any checkpoint whose QK-norm gains exceed max|w_q| max|w_k| = 3.96).
  const bool pre = true;
  const bool by_data = d->attn_mode == K5_ATTN_AUTO;
  float *qstat = nullptr, *kstat = nullptr;
  if (by_data) { K5CHK(ensure_attn_flags(d, s)); qstat = d->ws_attn_stats.as<float>(); kstat = qstat + H; }
  const int slot_blocks = rows_pad / 64;
  bf16_t* kun = kloc;                       // where the projection + norm + RoPE of the local keys happen
  bf16_t* kmeans = nullptr;                 // NABLA: [P][H][slot_blocks][64] key-block means, this rank's slot filled here
  if (nabla) {
    K5CHK(d->ws_kc.ensure((size_t)rows * D * 2));
    K5CHK(d->ws_kmeans.ensure((size_t)P * H * slot_blocks * 64 * 2));
    kun = d->ws_kc.as<bf16_t>();
    kmeans = d->ws_kmeans.as<bf16_t>();
  }
  // opt-in e4m3 projections (k5_dit_set_fp8 bit 1), as on one GPU: the rank's rows of h are quantised once and feed the K, V^T and Q GEMMs
  const bool f8_in = sa_sp_fp8_in(d, a, rows, rows_pad, nabla != nullptr);
  const uint8_t* wq8 = a.wqk8.as<uint8_t>();
  const uint8_t* wk8 = wq8 ? wq8 + (size_t)D * D : nullptr;
  if (h8_ready && !f8_in) { k5_set_error("internal: e4m3 activations handed to a bf16 projection"); return K5_ERR_ARG; }
  if (f8_in && !h8_ready) {
    K5CHK(d->ws_h8.ensure((size_t)rows * D));
    Scope sc(d, s, "elementwise");
    K5CHK(k5_launch_quant_rows_fp8(h, d->ws_h8.p, nullptr, rows, D, D, D, s));
  }
  {
    Scope sc(d, s, "gemm");
    if (f8_in) K5CHK(k5_launch_gemm_fp8(d->ws_h8.p, wk8, a.sqk.as<float>() + D, kun, rows, D, D, D, D, D, K5_EPI_BIAS, nullptr, 0, nullptr, s, a.bqk.as<float>() + D, 0));
    else K5CHK(k5_launch_gemm_bf16(h, wk, a.bqk.as<float>() + D, kun, rows, D, D, D, D, D, K5_EPI_BIAS, nullptr, 0, nullptr, s));
  }
  {
    Scope sc(d, s, "elementwise");
    // dense: scaled in place; NABLA: unscaled in place (kun), the scaled copy goes to this rank's slot of the gather buffer
    // NABLA: the unscaled keys' block means (all the map needs of them) come out of this pass ("nabla_fuse_means"; before: a second pass over kun)
    const bool fm = nabla && nabla_means_fused(d, rows, H);
    K5CHK(k5_launch_rmsnorm_rope(kun, a.norm.as<float>() + 64, cosT, sinT, rows, H, D, nullptr, s, K5_SOFTMAX_C, 0,
                                 nabla ? kloc : nullptr, nabla ? D : 0, by_data ? kstat + (size_t)r * H : nullptr, d->ws_attn_part.as<float>(), nullptr,
                                 nullptr, 0, fm ? kmeans + (size_t)r * H * slot_blocks * 64 : nullptr, slot_blocks));
  }
  if (nabla && !nabla_means_fused(d, rows, H)) {
    Scope sc(d, s, "nabla_map");
    K5CHK(k5_launch_nabla_block_means(kun, D, H, rows / 64, slot_blocks, kmeans + (size_t)r * H * slot_blocks * 64, s));
  }
  HIPCHK(hipEventRecord(d->ev_k, s));
  // sliced exchange (dense attention, "sp_slices" = S > 1): the slot of every rank is S slices of rows_pad / S tokens; V^T is laid
  // out slice-major inside the slot — [S][D][rows_pad / S] — so that a slice is contiguous on both operands (for the attention kernel
  // these are simply P S chunks of rows_pad / S keys) and is projected by one GEMM per slice
  const int S = (!nabla && d->sp_slices > 1 && P > 1) ? d->sp_slices : 1;
  const int cols = rows_pad / S;   // tokens per slice (the caller sized rows_pad as a multiple of 64 S)
  {
    Scope sc(d, s, "gemm");
    for (int sl = 0; sl < S; ++sl) {
      const int nsl = std::min(cols, rows - sl * cols);
      if (nsl <= 0) break;
      if (f8_in) K5CHK(k5_launch_gemm_fp8(a.wv8.p, d->ws_h8.as<uint8_t>() + (size_t)sl * cols * D, a.sv.as<float>(), vtloc + (size_t)sl * D * cols, D, nsl, D, D, D,
                                          cols, K5_EPI_BIAS, nullptr, 0, nullptr, s, a.bv.as<float>(), 1));
      else K5CHK(k5_launch_gemm_bf16(a.wv.p, (const bf16_t*)h + (size_t)sl * cols * D, a.bv.as<float>(), vtloc + (size_t)sl * D * cols, D, nsl, D, D, D,
                                     cols, K5_EPI_BIAS_M, nullptr, 0, nullptr, s));
    }
  }
  HIPCHK(hipEventRecord(d->ev_v, s));
  {
    Scope sc(d, s, "gemm");
    if (f8_in) K5CHK(k5_launch_gemm_fp8(d->ws_h8.p, wq8, a.sqk.as<float>(), q, rows, D, D, D, D, D, K5_EPI_BIAS, nullptr, 0, nullptr, s, a.bqk.as<float>(), 0));
    else K5CHK(k5_launch_gemm_bf16(h, wq, a.bqk.as<float>(), q, rows, D, D, D, D, D, K5_EPI_BIAS, nullptr, 0, nullptr, s));
  }
  // see run_self_attention; here only with "attn_fuse_qnorm" = 2: every pass of the schedule redoes the norm, and at shard sizes that
  // costs what the standalone pass over the local queries does (emulated P = 8: 82.8 vs 82.6 ms per step)
  const bool fuse_q = !nabla && d->fuse_qnorm > 1 && ((by_data && d->row_offsets) || d->attn_mode == K5_ATTN_ONLINE);
  const K5QueryNorm qn{a.norm.as<float>(), cosT, sinT, by_data ? d->ws_attn_cnt.as<unsigned long long>() : nullptr};
  const K5QueryNorm* qnp = fuse_q ? &qn : nullptr;
  if (!fuse_q) {
    void* mq = nullptr;
    if (nabla && nabla_means_fused(d, rows, H)) {   // the query-block means of the rank's rows, straight into the map's workspace
      K5CHK(d->ws_nabla.ensure(nabla_ws_bytes(d, H, N / 64, rows / 64)));
      k5_nabla_workspace_means(d->ws_nabla.p, H, N / 64, &mq, nullptr);
    }
    Scope sc(d, s, "elementwise");
    K5CHK(k5_launch_rmsnorm_rope(q, a.norm.as<float>(), cosT, sinT, rows, H, D, nullptr, s, 1.f, 0x7fffffff, nullptr, 0, qstat, d->ws_attn_part.as<float>(), nullptr,
                                 mq, rows / 64, nullptr, 0));
  }
  // all-gathers on the side stream (they only need k / v^T, which are complete at ev_k / ev_v) ...
  hipStream_t cs = d->comm_stream;
  HIPCHK(hipStreamWaitEvent(cs, d->ev_k, 0));
  {
    Scope sc(d, cs, "comm");
    if (by_data) {   // H floats per rank, first: the flags must exist before pass 1 (both passes take the same form per head)
      K5CHK(d->comm.all_gather_inplace(kstat, (size_t)H, 4, cs));
      HIPCHK(hipEventRecord(d->ev_stats, cs));
    }
    if (nabla) {   // 28 x 64 values per block: lands long before the keys, and the map is computed while they travel
      K5CHK(d->comm.all_gather_inplace(kmeans, (size_t)H * slot_blocks * 64, 2, cs));
      HIPCHK(hipEventRecord(d->ev_means, cs));
    }
    if (S == 1) {
      K5CHK(d->comm.all_gather_inplace(kfull, (size_t)rows_pad * D, 2, cs));
      HIPCHK(hipStreamWaitEvent(cs, d->ev_v, 0));
      K5CHK(d->comm.all_gather_inplace(vtfull, (size_t)D * ldv, 2, cs));
    } else {
      const size_t slot = (size_t)rows_pad * D * 2, sl_bytes = (size_t)cols * D * 2;   // the same for K and for V^T
      for (int sl = 0; sl < S; ++sl) {
        K5CHK(d->comm.slot_exchange(kfull, slot, sl * sl_bytes, sl_bytes, cs));
        if (sl == 0) HIPCHK(hipStreamWaitEvent(cs, d->ev_v, 0));
        K5CHK(d->comm.slot_exchange(vtfull, slot, sl * sl_bytes, sl_bytes, cs));
        HIPCHK(hipEventRecord(d->ev_slice[sl], cs));
      }
    }
  }
  HIPCHK(hipEventRecord(d->ev_gathered, cs));
  const int* hflags = nullptr;
  // per-row softmax offsets (§4.1) across the passes: a row that underflows in ANY pass marks its head late (flag 2), the fixed
  // form then skips the head and the online form of the LAST pass recomputes it from scratch over all keys (late_pass 1 / 2)
  const float* kmax = nullptr;
  K5KeyCentre kcen{nullptr, nullptr, nullptr};
  const K5KeyCentre* kcp = nullptr;
  if (by_data) {
    HIPCHK(hipStreamWaitEvent(s, d->ev_stats, 0));
    hflags = d->ws_attn_flags.as<int>();
    float* kmax_w = d->row_offsets ? d->ws_attn_flags.as<float>() + H : nullptr;
    kmax = kmax_w;
    if (kmax_w) K5CHK(ensure_zeroed(a.pref, (size_t)2 * H * 4, s));
    // heads beyond the window of the plain offsets: anchored offsets, sampled from the rank's OWN keys (the row's own block is among them;
    // they are in place before the gather) — every pass of the schedule then runs the head on them
    const bool anchored = kmax_w && d->anchor && !fuse_q && !nabla;   // dense attention only
    K5CHK(k5_launch_attn_flags(qstat, kstat, P, H, H, 0, d->ws_attn_flags.as<int>(), d->ws_attn_cnt.as<unsigned long long>(), s, kmax_w,
                               kmax_w ? (a.pref.as<int>() + (size_t)pref_slot * H) : nullptr, nullptr, nullptr, 1, 0, anchored));
    if (anchored) {
      K5CHK(d->ws_attn_anchor.ensure((size_t)H * rows * 4));
      K5CHK(k5_launch_attn_row_anchor(q, kloc, H, rows, rows, D, D, 0, N, kmax_w, d->ws_attn_anchor.as<float>(), s));
      kcen.row_anchor = d->ws_attn_anchor.as<float>();
      kcp = &kcen;
    }
  }
  const int variant = pre ? d->attn_mode : K5_ATTN_AUTO;
  if (nabla) {
    // NABLA under sequence parallelism: the map rows of this rank's query blocks need the block means of
    // ALL keys -> wait for the gathered means, bring them into the map's layout, select (local query blocks x all key blocks)
    // while K' / V^T are still on their way; then the list-driven attention on the chunked V^T layout (single pass).
    HIPCHK(hipStreamWaitEvent(s, d->ev_means, 0));
    const int nb = N / 64;
    // sparse maps: 128-query workgroups (lists per two rows, see nabla_group_rows) — without the split-job / two-pass machinery, which
    // lives on the 256-query form; dense maps: that form, balanced
    const int grp = (d->nabla_grp_now < 4 && d->sp_nabla_passes == 1) ? d->nabla_grp_now : 4;
    const int pair = (grp == 2 && d->nabla_pair_frames) ? nabla->Hb * nabla->Wb : 0;   // rows l and l + S of the rank's shard: same tile, next frame
    K5CHK(d->ws_nabla.ensure(nabla_ws_bytes(d, H, nb, rows / 64)));   // the logits and list regions by the rank's own query-block rows
    {
      Scope sc(d, s, "nabla_map");
      K5CHK(k5_launch_nabla_key_means_from_slots(kmeans, H, nb, slot_blocks, d->ws_nabla.p, s));
      K5CHK(k5_launch_nabla_select_rect(nabla_means_fused(d, rows, H) ? nullptr : q, nullptr, D, 0, H, rows, r * slot_blocks, N, nabla->T, nabla->Hb, nabla->Wb, nabla->wT,
                                        nabla->wH, nabla->wW, nabla->P, d->ws_nabla.p, s, r * slot_blocks, rows / 64, grp, pair));   // own key blocks lead the lists
    }
    if (d->profiling) {
      K5CHK(ensure_zeroed(d->ws_attn_cnt, 32, s));
      K5CHK(k5_launch_nabla_count(d->ws_nabla.p, H, rows / 64, nb, d->ws_attn_cnt.as<unsigned long long>() + 2, s));
      d->nabla_possible += (long long)H * (rows / 64) * nb;
    }
    const int *list, *cnt, *cnt_local;
    k5_nabla_workspace_views(d->ws_nabla.p, H, nb, nullptr, nullptr, &list, &cnt, &cnt_local, rows / 64);
    K5CHK(d->ws_attn_bal.ensure(k5_attention_balance_bytes(H, rows)));
    K5CHK(nabla_density_hint(d, H, rows / 64, nb, s));
    if (grp < 4) {
      HIPCHK(hipStreamWaitEvent(s, d->ev_gathered, 0));
      Scope sc(d, s, "attn_self");
      K5CHK(k5_launch_attention_bf16_sparse(q, kfull, vtfull, o, H, rows, N, D, D, ldv, D, 0.f, list, cnt, nb, rows_pad,
                                            (long long)D * ldv, s, true, hflags, variant, kmax, nullptr, d->ws_attn_bal.as<float>(), grp, true, kcp, pair));
    } else if (d->sp_nabla_passes > 1 && P > 1) {
      // two passes over every list: the rank's own key blocks (they lead the lists; K' / V^T of them are in place) while the other
      // ranks' keys travel — state out —, then the rest once the gather has landed (resume, normalise); late fallback as in the dense
      // schedule (a head whose row underflows on its per-row offset in pass 1 is recomputed from scratch by the online form of pass 2)
      K5CHK(d->ws_attn_state.ensure(k5_attention_state_bytes(H, rows)));
      {
        const K5SparsePass p1{nullptr, d->ws_attn_state.as<float>(), 2, kmax ? 1 : 0};
        Scope sc(d, s, "attn_self");
        K5CHK(k5_launch_attention_bf16_sparse(q, kfull, vtfull, o, H, rows, N, D, D, ldv, D, 0.f, list, cnt_local, nb, rows_pad,
                                              (long long)D * ldv, s, true, hflags, variant, kmax, &p1, d->ws_attn_bal.as<float>(), 4, true, kcp));
      }
      HIPCHK(hipStreamWaitEvent(s, d->ev_gathered, 0));
      const K5SparsePass p2{cnt_local, d->ws_attn_state.as<float>(), 1, kmax ? 2 : 0};
      Scope sc(d, s, "attn_self");
      K5CHK(k5_launch_attention_bf16_sparse(q, kfull, vtfull, o, H, rows, N, D, D, ldv, D, 0.f, list, cnt, nb, rows_pad,
                                            (long long)D * ldv, s, true, hflags, variant, kmax, &p2, d->ws_attn_bal.as<float>(), 4, true, kcp));
    } else {
      HIPCHK(hipStreamWaitEvent(s, d->ev_gathered, 0));
      Scope sc(d, s, "attn_self");
      K5CHK(k5_launch_attention_bf16_sparse(q, kfull, vtfull, o, H, rows, N, D, D, ldv, D, 0.f, list, cnt, nb, rows_pad,
                                            (long long)D * ldv, s, true, hflags, variant, kmax, nullptr, d->ws_attn_bal.as<float>(), 4, true, kcp));
    }
  } else {
    // ... while the main stream attends the local query rows to the LOCAL key chunk (pass 1, leaves the fp32 state),
    // then, once every chunk has arrived, to all the other chunks (pass 2, resumes the state and normalises).
    const int tpc = rows / 64, tpc_pad = rows_pad / 64, total = N / 64;
    // Pass 1 exists to cover the gather (its tail jobs are split and merged back into the state like those of the final pass),
    // so it should be no longer than the gather.  How long the gather takes is a property of the node (one
    // xGMI link per GPU pair: ~341 MB / (P x link rate) per block, i.e. about 0.4 of a rank's attention time at any P if a link
    // gives ~70 GB/s each way, much less if RCCL drives several paths) and cannot be measured here, so the default is the
    // safe one — all local key tiles — and k5_dit_set_option("sp_pass1_tiles") sets it on a real node (emulated, P = 2: 41
    // tiles instead of 372 take the step from 282 to 259 ms).
    int k1 = (d->sp_pass1_tiles > 0 && S == 1) ? d->sp_pass1_tiles : tpc;   // the sliced schedule attends all local tiles first
    k1 = k1 > tpc ? tpc : k1;
    K5CHK(d->ws_attn_state.ensure(k5_attention_state_bytes(H, rows)));
    K5CHK(d->ws_attn_bal.ensure(k5_attention_balance_bytes(H, rows)));
    {
      Scope sc(d, s, "attn_self");
      K5CHK(k5_launch_attention_bf16_range(q, kfull, vtfull, o, H, rows, N, D, D, S > 1 ? cols : ldv, D, 0.f, S > 1 ? cols : rows_pad,
                                           S > 1 ? (long long)D * cols : (long long)D * ldv,
                                           r * tpc_pad, k1, 0x7fffffff, 0, d->ws_attn_state.as<float>(), 2, s, d->ws_attn_bal.as<float>(), true, hflags, variant,
                                           nullptr, kmax, kmax ? 1 : 0, qnp, kcp));
    }
    if (S == 1) {
      HIPCHK(hipStreamWaitEvent(s, d->ev_gathered, 0));
      Scope sc(d, s, "attn_self");
      K5CHK(k5_launch_attention_bf16_range(q, kfull, vtfull, o, H, rows, N, D, D, ldv, D, 0.f, rows_pad, (long long)D * ldv,
                                           0, total - k1, r * tpc_pad, k1, d->ws_attn_state.as<float>(), 1, s, d->ws_attn_bal.as<float>(),
                                           true, hflags, variant, nullptr, kmax, kmax ? 2 : 0, qnp, kcp));
    } else {
      // one pass per slice, as soon as that slice of every peer has landed: slice sl of rank p = key tiles [p tpc_pad + sl tps, + tps)
      // (P - 1 segments: mine was pass 1; the last rank's slot may end early — it is the last segment, so the count is cut short).
      // The state is resumed and saved between the passes (tail jobs balanced in every pass); the last one normalises.
      const int tps = tpc_pad / S;
      const int last_rank_tiles = total - (P - 1) * tpc_pad;                 // > 0 (checked by the caller)
      int cnts[4] = {0, 0, 0, 0}, last_pass = 0;
      for (int sl = 0; sl < S; ++sl) {
        cnts[sl] = (P - 1) * tps;
        if (r != P - 1) cnts[sl] -= tps - std::max(0, std::min(tps, last_rank_tiles - sl * tps));
        if (cnts[sl] > 0) last_pass = sl;                                     // slice 0 always has tiles
      }
      for (int sl = 0; sl < S; ++sl) {
        const int cnt = cnts[sl];
        HIPCHK(hipStreamWaitEvent(s, d->ev_slice[sl], 0));
        if (cnt <= 0) continue;
        const K5TileSegments seg{tps, tpc_pad, r};
        const bool fin = sl == last_pass;
        Scope sc(d, s, "attn_self");
        K5CHK(k5_launch_attention_bf16_range(q, kfull, vtfull, o, H, rows, N, D, D, cols, D, 0.f, cols, (long long)D * cols,
                                             sl * tps, cnt, 0x7fffffff, 0, d->ws_attn_state.as<float>(), fin ? 1 : 3, s,
                                             d->ws_attn_bal.as<float>(), true, hflags, variant, &seg, kmax, kmax ? (fin ? 2 : 1) : 0, qnp, kcp));
      }
    }
  }
  if (kmax) K5CHK(k5_launch_attn_pref_update(d->ws_attn_bal.as<float>(), H, rows, (nabla && d->nabla_grp_now < 4 && d->sp_nabla_passes == 1) ? d->nabla_grp_now : 4, (a.pref.as<int>() + (size_t)pref_slot * H), s));
  {
    Scope sc(d, s, "gemm");
    K5CHK(k5_launch_gemm_bf16(o, a.wo.p, a.bo.as<float>(), resid, rows, D, D, D, D, D, K5_EPI_GATE, resid, D, gate, s));
  }
  return K5_OK;
}

// Ulysses-style sequence parallelism (north_star; "sp_mode" = 1): instead of gathering every rank's K / V^T, the ranks trade token rows for
// HEADS — an all-to-all turns (rows of this rank x all 28 heads) into (all N rows x the 28 / P heads of this rank) for q | k and V^T, the
// rank runs the ONE-GPU attention (a single balanced pass over all keys: no fp32 state between passes) for its heads, and a second
// all-to-all brings the outputs back to the token shards.  Per rank and block 4 x (P - 1) / P^2 x N x D x 2 B leave the GPU (3/16 of
// the 170 MB of a 5 s clip per tensor at P = 4) against 2 x (P - 1) / P x N x D x 2 B of ingress for the gather: a ratio of 2 / P —
// the same bytes at P = 2, half at P = 4, 2/7 at P = 7 — but the attention cannot start before q, k AND V^T have landed, where the gather hides behind pass 1.
// Needs heads % P == 0 (28 = 2 x 2 x 7: P in 2, 4, 7, 14) and dense attention; anything else keeps the gather.  Replaces the head split
// of the reference's tensor-parallel plan (parallelize.py:87-91) together with its all-reduces.
// Layouts: send [P][rows_pad][2 Dp] (block g: q | k' of the heads of rank g, this rank's rows) -> recv [P][rows_pad][2 Dp] = all rows
// (rank-major = token order: slots are contiguous token ranges) of this rank's heads; V^T [D][rows_pad] is destination-major as the GEMM
// writes it (block g = rows g Dp ..) -> vrecv [P][Dp][rows_pad] = the chunked V^T layout of the attention kernel; o [P rows_pad][Dp]
// -> orecv [P][rows_pad][Dp] (block g: the outputs of rank g's heads for this rank's rows) -> unpacked to [rows][D].
int run_self_attention_ulysses(k5_dit* d, hipStream_t s, const AttnW& a, const void* h, int rows, int rows_pad, int N, void* o,
                               const float* cosT, const float* sinT, void* resid, const float* gate, int pref_slot = 0) {
  const int D = d->D, H = d->Hh, P = d->sp_world, r = d->sp_rank, Hp = H / P, Dp = D / P;
  const bool by_data = d->attn_mode == K5_ATTN_AUTO;
  K5CHK(d->ws_qk.ensure((size_t)rows * 2 * D * 2));
  K5CHK(d->ws_u_send.ensure((size_t)P * rows_pad * 2 * Dp * 2)); K5CHK(d->ws_u_recv.ensure((size_t)P * rows_pad * 2 * Dp * 2));
  K5CHK(d->ws_u_vsend.ensure((size_t)D * rows_pad * 2)); K5CHK(d->ws_u_vrecv.ensure((size_t)D * rows_pad * 2));
  K5CHK(d->ws_u_o.ensure((size_t)P * rows_pad * Dp * 2)); K5CHK(d->ws_u_orecv.ensure((size_t)P * rows_pad * Dp * 2));
  K5CHK(ensure_zeroed(d->ws_u_stats, (size_t)P * 2 * H * 4, s));
  if (!d->ev_u_o) { HIPCHK(hipEventCreateWithFlags(&d->ev_u_o, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&d->ev_u_back, hipEventDisableTiming)); }
  if (by_data) K5CHK(ensure_attn_flags(d, s));
  float* ustats = d->ws_u_stats.as<float>();         // [P][2 H]: rank p's maxima of |q_h|^2 (H) and |k'_h|^2 (H) over ITS rows
  if (by_data) HIPCHK(hipMemsetAsync(ustats + (size_t)r * 2 * H, 0, (size_t)2 * H * 4, s));   // the norm pass max-accumulates; only my heads' entries were consumed
  bf16_t* qk = d->ws_qk.as<bf16_t>();
  {
    Scope sc(d, s, "gemm");
    K5CHK(k5_launch_gemm_bf16(h, a.wqk.p, a.bqk.as<float>(), qk, rows, 2 * D, D, D, D, 2 * D, K5_EPI_BIAS, nullptr, 0, nullptr, s));
  }
  {
    Scope sc(d, s, "elementwise");
    const int32_t hc[2] = {H, 2 * H};
    K5CHK(k5_launch_rmsnorm_rope(qk, a.norm.as<float>(), cosT, sinT, rows, 2 * H, 2 * D, hc, s, K5_SOFTMAX_C, H, nullptr, 0,
                                 by_data ? ustats + (size_t)r * 2 * H : nullptr, d->ws_attn_part.as<float>()));
    K5CHK(k5_launch_ulysses_pack_qk(qk, d->ws_u_send.p, rows, rows_pad, D, P, s));
  }
  HIPCHK(hipEventRecord(d->ev_k, s));
  hipStream_t cs = d->comm_stream;
  HIPCHK(hipStreamWaitEvent(cs, d->ev_k, 0));
  {
    Scope sc(d, cs, "comm");
    if (by_data) { K5CHK(d->comm.all_gather_inplace(ustats, (size_t)2 * H, 4, cs)); HIPCHK(hipEventRecord(d->ev_stats, cs)); }
    K5CHK(d->comm.all_to_all(d->ws_u_send.p, d->ws_u_recv.p, (size_t)rows_pad * 2 * Dp * 2, cs));
  }
  {
    Scope sc(d, s, "gemm");   // V^T of the rank's rows while q | k travel
    K5CHK(k5_launch_gemm_bf16(a.wv.p, h, a.bv.as<float>(), d->ws_u_vsend.p, D, rows, D, D, D, rows_pad, K5_EPI_BIAS_M, nullptr, 0, nullptr, s));
  }
  HIPCHK(hipEventRecord(d->ev_v, s));
  HIPCHK(hipStreamWaitEvent(cs, d->ev_v, 0));
  {
    Scope sc(d, cs, "comm");
    K5CHK(d->comm.all_to_all(d->ws_u_vsend.p, d->ws_u_vrecv.p, (size_t)Dp * rows_pad * 2, cs));
  }
  HIPCHK(hipEventRecord(d->ev_gathered, cs));
  const int* hflags = nullptr;
  const float* kmax = nullptr;
  if (by_data) {   // flags of MY heads from every rank's maxima (each rank held some rows of them)
    HIPCHK(hipStreamWaitEvent(s, d->ev_stats, 0));
    hflags = d->ws_attn_flags.as<int>();
    float* kmax_w = d->row_offsets ? d->ws_attn_flags.as<float>() + H : nullptr;
    kmax = kmax_w;
    if (kmax_w) K5CHK(ensure_zeroed(a.pref, (size_t)2 * H * 4, s));
    K5CHK(k5_launch_attn_flags(ustats + (size_t)r * Hp, ustats + H + (size_t)r * Hp, P, 2 * H, Hp, 0, d->ws_attn_flags.as<int>(),
                               d->ws_attn_cnt.as<unsigned long long>(), s, kmax_w, kmax_w ? (a.pref.as<int>() + (size_t)pref_slot * H) + (size_t)r * Hp : nullptr, nullptr, nullptr, P, 2 * H,
                               kmax_w && d->anchor));
  }
  HIPCHK(hipStreamWaitEvent(s, d->ev_gathered, 0));
  const bf16_t* qall = d->ws_u_recv.as<bf16_t>();
  K5CHK(d->ws_attn_bal.ensure(k5_attention_balance_bytes(Hp, N)));
  K5KeyCentre kcen{nullptr, nullptr, nullptr};
  if (kmax && d->anchor) {   // anchored offsets of the heads beyond the window: this rank holds all rows of its heads
    K5CHK(d->ws_attn_anchor.ensure((size_t)Hp * N * 4));
    Scope sc(d, s, "elementwise");
    K5CHK(k5_launch_attn_row_anchor(qall, qall + Dp, Hp, N, N, 2 * Dp, 2 * Dp, 0, N, kmax, d->ws_attn_anchor.as<float>(), s));
    kcen.row_anchor = d->ws_attn_anchor.as<float>();
  }
  {
    Scope sc(d, s, "attn_self");
    K5CHK(k5_launch_attention_bf16_range(qall, qall + Dp, d->ws_u_vrecv.p, d->ws_u_o.p, Hp, N, N, 2 * Dp, 2 * Dp, rows_pad, Dp, 0.f, rows_pad,
                                         (long long)Dp * rows_pad, 0, -1, 0x7fffffff, 0, nullptr, 0, s, d->ws_attn_bal.as<float>(), true, hflags,
                                         d->attn_mode, nullptr, kmax, 0, nullptr, kcen.row_anchor ? &kcen : nullptr));
  }
  if (kmax) K5CHK(k5_launch_attn_pref_update(d->ws_attn_bal.as<float>(), Hp, N, 4, (a.pref.as<int>() + (size_t)pref_slot * H) + (size_t)r * Hp, s));
  HIPCHK(hipEventRecord(d->ev_u_o, s));
  HIPCHK(hipStreamWaitEvent(cs, d->ev_u_o, 0));
  {
    Scope sc(d, cs, "comm");
    K5CHK(d->comm.all_to_all(d->ws_u_o.p, d->ws_u_orecv.p, (size_t)rows_pad * Dp * 2, cs));
  }
  HIPCHK(hipEventRecord(d->ev_u_back, cs));
  HIPCHK(hipStreamWaitEvent(s, d->ev_u_back, 0));
  {
    Scope sc(d, s, "elementwise");
    K5CHK(k5_launch_ulysses_unpack_o(d->ws_u_orecv.p, o, rows, rows_pad, D, P, s));
  }
  {
    Scope sc(d, s, "gemm");
    K5CHK(k5_launch_gemm_bf16(o, a.wo.p, a.bo.as<float>(), resid, rows, D, D, D, D, D, K5_EPI_GATE, resid, D, gate, s));
  }
  return K5_OK;
}

// keys and V^T of every visual block's cross-attention from the text stream (k5_dit::cross_kv_batched): ws_ck_all [L][blocks * D] (normalised
// keys), ws_cvt_all [blocks * D][rup(L, 8)].  The key projection is the launch it was per block (128 x 128 tiles for 256 rows); the V^T one is forced
// onto the same 128 x 128 kernel the per-block call took (57 344 weight rows would otherwise pick a 256-row kernel: another summation order).
int cross_kv_batched_run(k5_dit* d, hipStream_t s, const void* text, int L) {
  const int D = d->D, H = d->Hh, nbv = (int)d->vblocks.size();
  const int ldvt = (int)rup(L, 8);
  K5CHK(d->ws_ck_all.ensure((size_t)L * nbv * D * 2)); K5CHK(d->ws_cvt_all.ensure((size_t)nbv * D * ldvt * 2));
  {
    Scope sc(d, s, "gemm");
    K5CHK(k5_launch_gemm_bf16(text, d->cx_wk_all.p, d->cx_bk_all.as<float>(), d->ws_ck_all.p, L, nbv * D, D, D, D, nbv * D, K5_EPI_BIAS, nullptr, 0, nullptr, s, 2));
    K5CHK(k5_launch_gemm_bf16(d->cx_wv_all.p, text, d->cx_bv_all.as<float>(), d->ws_cvt_all.p, nbv * D, L, D, D, D, ldvt, K5_EPI_BIAS_M, nullptr, 0, nullptr, s, 2));
  }
  Scope sc(d, s, "elementwise");
  const int32_t hc[2] = {H, 0};   // one norm weight per block's H heads, no RoPE
  return k5_launch_rmsnorm_rope(d->ws_ck_all.p, d->cx_knorm_all.as<float>(), nullptr, nullptr, L, nbv * H, nbv * D, hc, s);
}

// kv_ready: ck (row stride ldck) / cvt already hold this block's normalised keys and V^T (cross_kv_batched_run)
int run_cross_attention(k5_dit* d, hipStream_t s, const AttnW& a, const void* h, int rows, const void* text,
                        int L, void* q, void* ck, void* cvt, void* o, void* resid, const float* gate, bool kv_ready = false, int ldck = 0) {
  const int D = d->D, H = d->Hh;
  const int ldvt = (int)rup(L, 8);
  if (!kv_ready) ldck = D;
  {
    Scope sc(d, s, "gemm");
    K5CHK(k5_launch_gemm_bf16(h, a.wq.p, a.bq.as<float>(), q, rows, D, D, D, D, D, K5_EPI_BIAS, nullptr, 0, nullptr, s));
    if (!kv_ready) {
      K5CHK(k5_launch_gemm_bf16(text, a.wk.p, a.bk.as<float>(), ck, L, D, D, D, D, D, K5_EPI_BIAS, nullptr, 0, nullptr, s));
      K5CHK(k5_launch_gemm_bf16(a.wv.p, text, a.bv.as<float>(), cvt, D, L, D, D, D, ldvt, K5_EPI_BIAS_M, nullptr, 0, nullptr, s));
    }
  }
  // RMSNorm of the queries (no RoPE in cross-attention, nn.py:330-334) is fused into the attention kernel's Q-fragment load when the
  // weight-derived bound admits the fixed-offset kernel: one pass over the (rows, D) projection less per block
  const bool fuse_qnorm = a.score_bound > 0.f && a.score_bound * K5_SOFTMAX_C <= 90.f;
  {
    Scope sc(d, s, "elementwise");
    if (!fuse_qnorm) K5CHK(k5_launch_rmsnorm_rope(q, a.norm.as<float>(), nullptr, nullptr, rows, H, D, nullptr, s));
    if (!kv_ready) K5CHK(k5_launch_rmsnorm_rope(ck, a.norm.as<float>() + 64, nullptr, nullptr, L, H, D, nullptr, s));
  }
  {
    Scope sc(d, s, "attn_cross");
    const K5QueryNorm qn{a.norm.as<float>(), nullptr, nullptr, nullptr};
    if (fuse_qnorm)
      K5CHK(k5_launch_attention_bf16_range(q, ck, cvt, o, H, rows, L, D, ldck, ldvt, D, a.score_bound, 0, 0, 0, -1, 0x7fffffff, 0, nullptr, 0, s,
                                           nullptr, false, nullptr, K5_ATTN_AUTO, nullptr, nullptr, 0, &qn));
    else
      K5CHK(k5_launch_attention_bf16_bounded(q, ck, cvt, o, H, rows, L, D, ldck, ldvt, D, a.score_bound, s));
  }
  {
    Scope sc(d, s, "gemm");
    K5CHK(k5_launch_gemm_bf16(o, a.wo.p, a.bo.as<float>(), resid, rows, D, D, D, D, D, K5_EPI_GATE, resid, D, gate, s));
  }
  return K5_OK;
}

int run_ff(k5_dit* d, hipStream_t s, const BlockW& b, const void* h, int rows, void* ff, void* resid, const float* gate, bool h8_ready = false) {
  const int D = d->D, FF = d->FF;
  if (h8_ready && !ff_fp8_in(d, b, rows)) { k5_set_error("internal: e4m3 activations handed to a bf16 feed-forward"); return K5_ERR_ARG; }
  if (ff_fp8_in(d, b, rows)) {
    // opt-in lossy path: h -> e4m3 (static scale; by the LayerNorm itself when h8_ready), FF1 on the fp8 MFMA with the GELU epilogue writing
    // e4m3, FF2 likewise with the gated-residual epilogue (gemm_fp8.hip)
    K5CHK(d->ws_h8.ensure((size_t)rows * D)); K5CHK(d->ws_ff8.ensure((size_t)rows * FF));
    if (!h8_ready) {
      Scope sc(d, s, "elementwise");
      K5CHK(k5_launch_quant_rows_fp8(h, d->ws_h8.p, nullptr, rows, D, D, D, s));
    }
    Scope sc(d, s, "gemm");
    K5CHK(k5_launch_gemm_fp8(d->ws_h8.p, b.w1_f8.p, b.s1_f8.as<float>(), d->ws_ff8.p, rows, FF, D, D, D, FF, K5_EPI_GELU, nullptr, 0, nullptr, s));
    K5CHK(k5_launch_gemm_fp8(d->ws_ff8.p, b.w2_f8.p, b.s2_f8.as<float>(), resid, rows, D, FF, FF, FF, D, K5_EPI_GATE, resid, D, gate, s));
    return K5_OK;
  }
  Scope sc(d, s, "gemm");
  K5CHK(k5_launch_gemm_bf16(h, b.w1.p, nullptr, ff, rows, FF, D, D, D, FF, K5_EPI_GELU, nullptr, 0, nullptr, s));
  K5CHK(k5_launch_gemm_bf16(ff, b.w2.p, nullptr, resid, rows, D, FF, FF, FF, D, K5_EPI_GATE, resid, D, gate, s));
  return K5_OK;
}

// forget which heads the per-row-offset softmax served badly (AttnW::pref): the memory is a speed hint for the NEXT step of the SAME sampling
// run; carried across runs it made results depend on what the handle had computed before
int reset_attn_pref(k5_dit* d, hipStream_t s, bool sync = false) {
  if (sync) HIPCHK(hipDeviceSynchronize());   // called outside any stream order (k5_dit_set_option): nothing may still be reading the flags
  for (auto& b : d->vblocks)
    if (b.self_attn.pref.p) HIPCHK(hipMemsetAsync(b.self_attn.pref.p, 0, b.self_attn.pref.bytes, s));
  if (sync) HIPCHK(hipDeviceSynchronize());
  return K5_OK;
}

int ln_mod(k5_dit* d, hipStream_t s, const void* x, const float* mod3, void* out, int rows, bool to_e4m3 = false) {
  // mod3 = [shift | scale | gate] (dit.py:36,40,64,69,74).  to_e4m3: every consumer of this h is an fp8 GEMM — the rows go to ws_h8 as e4m3
  // (the bf16 rounding, then the e4m3 one: what the separate quantisation pass made of the bf16 h), and no bf16 h is written at all
  if (to_e4m3) K5CHK(d->ws_h8.ensure((size_t)rows * d->D));
  Scope sc(d, s, "elementwise");
  return k5_launch_ln_modulate(x, mod3 + d->D, mod3, to_e4m3 ? nullptr : out, rows, d->D, d->D, d->D, s, to_e4m3 ? d->ws_h8.p : nullptr);
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
int ensure_workspaces(k5_dit* d, int N, int L) {
  const size_t D = d->D, FF = d->FF;
  const size_t Lr = rup(L, 8), Nr = rup(N, 8);
  K5CHK(d->ws_text_in.ensure((size_t)L * d->cfg.in_text_dim * 2));
  K5CHK(d->ws_text.ensure(L * D * 2)); K5CHK(d->ws_th.ensure(L * D * 2)); K5CHK(d->ws_tqk.ensure(L * 2 * D * 2));
  K5CHK(d->ws_tvt.ensure(D * Lr * 2)); K5CHK(d->ws_to.ensure(L * D * 2)); K5CHK(d->ws_tff.ensure(L * FF * 2));
  K5CHK(d->ws_pool_in.ensure((size_t)rup(d->cfg.in_text_dim2, 8) * 2)); K5CHK(d->ws_pool_lin.ensure(d->TD * 2));
  K5CHK(d->ws_pool_f32.ensure(d->TD * 4)); K5CHK(d->ws_time.ensure(16)); K5CHK(d->ws_tfeat.ensure(D * 4));
  K5CHK(d->ws_th1.ensure(d->TD * 4)); K5CHK(d->ws_temb.ensure(d->TD * 4)); K5CHK(d->ws_mod.ensure(d->mod_rows * 4));
  K5CHK(d->ws_xp.ensure((size_t)N * d->KvisPad * 2)); K5CHK(d->ws_vis.ensure(N * D * 2)); K5CHK(d->ws_h.ensure(N * D * 2));
  K5CHK(d->ws_qk.ensure((size_t)N * 2 * D * 2)); K5CHK(d->ws_vt.ensure(D * Nr * 2)); K5CHK(d->ws_o.ensure(N * D * 2));
  K5CHK(d->ws_ff.ensure((size_t)N * FF * 2)); K5CHK(d->ws_ck.ensure(L * D * 2)); K5CHK(d->ws_cvt.ensure(D * Lr * 2));
  K5CHK(d->ws_y.ensure((size_t)N * d->Fout * 2));
  K5CHK(d->ws_vcos.ensure((size_t)N * 32 * 4)); K5CHK(d->ws_vsin.ensure((size_t)N * 32 * 4));
  return K5_OK;
}

int prepare_rope(k5_dit* d, hipStream_t s, const k5_forward_args* a, int Tp, int Hp, int Wp, const int32_t* tok_perm) {
  // visual tables: recompute only when the (shape, positions, scale) key changes — step-invariant (K15)
  std::vector<int32_t> key;
  key.insert(key.end(), a->pos_t, a->pos_t + Tp); key.insert(key.end(), a->pos_h, a->pos_h + Hp);
  key.insert(key.end(), a->pos_w, a->pos_w + Wp);
  const bool same = key == d->key_vpos && d->key_shape[0] == Tp && d->key_shape[1] == Hp && d->key_shape[2] == Wp &&
                    !memcmp(d->key_scale, a->scale_factor, 12) && d->key_fractal == (tok_perm != nullptr);
  if (!same) {
    K5CHK(d->ws_pos.ensure(key.size() * 4));
    HIPCHK(hipMemcpyAsync(d->ws_pos.p, key.data(), key.size() * 4, hipMemcpyHostToDevice, s));
    const int32_t* p = d->ws_pos.as<int32_t>();
    const int n0 = d->cfg.axes_dims[0] / 2, n1 = d->cfg.axes_dims[1] / 2, n2 = d->cfg.axes_dims[2] / 2;
    K5CHK(k5_launch_rope_table(d->ws_vcos.as<float>(), d->ws_vsin.as<float>(), p, p + Tp, p + Tp + Hp, Tp, Hp, Wp, n0, n1,
                               n2, a->scale_factor[0], a->scale_factor[1], a->scale_factor[2], tok_perm, s));
    HIPCHK(hipStreamSynchronize(s));  // key vector is host memory reused below; once per shape only
    d->key_fractal = tok_perm != nullptr;
    d->key_vpos = key; d->key_shape[0] = Tp; d->key_shape[1] = Hp; d->key_shape[2] = Wp;
    memcpy(d->key_scale, a->scale_factor, 12);
  }
  return K5_OK;
}

int prepare_text_rope(k5_dit* d, hipStream_t s, const k5_text_cond& c, const float** cosT, const float** sinT) {
  // RoPE1D tables (nn.py:99-116) are step-invariant: cache per position vector (cond / null cond)
  std::vector<int32_t> key(c.text_rope_pos, c.text_rope_pos + c.text_len);
  for (auto& e : d->text_rope)
    if (e.key == key) { *cosT = e.cosT.as<float>(); *sinT = e.sinT.as<float>(); return K5_OK; }
  if (d->text_rope.size() >= 8) {
    HIPCHK(hipStreamSynchronize(s));
    for (auto& e : d->text_rope) { e.cosT.release(); e.sinT.release(); e.pos.release(); }
    d->text_rope.clear();
  }
  d->text_rope.emplace_back();
  auto& e = d->text_rope.back();
  e.key = key;
  K5CHK(e.pos.ensure(key.size() * 4)); K5CHK(e.cosT.ensure(key.size() * 32 * 4)); K5CHK(e.sinT.ensure(key.size() * 32 * 4));
  HIPCHK(hipMemcpyAsync(e.pos.p, e.key.data(), key.size() * 4, hipMemcpyHostToDevice, s));
  K5CHK(k5_launch_rope_table(e.cosT.as<float>(), e.sinT.as<float>(), e.pos.as<int32_t>(), nullptr, nullptr, c.text_len, 1,
                             1, 32, 0, 0, 1.f, 1.f, 1.f, nullptr, s));
  HIPCHK(hipStreamSynchronize(s));  // once per distinct prompt length
  *cosT = e.cosT.as<float>(); *sinT = e.sinT.as<float>();
  return K5_OK;
}

int to_bf16(k5_dit* d, hipStream_t s, const void* src, int dtype, size_t n, DevBuf& dst, const void** out) {
  (void)d;
  if (dtype == K5_BF16) { *out = src; return K5_OK; }
  if (dtype != K5_F32) { k5_set_error("text embeddings must be f32 or bf16"); return K5_ERR_UNSUPPORTED; }
  K5CHK(dst.ensure(n * 2));
  K5CHK(k5_launch_cast_f32_bf16((const float*)src, dst.p, (int64_t)n, s));
  *out = dst.p;
  return K5_OK;
}

// ---------------------------------------------------------------------------------------------
// self-tuning sequence-parallel schedule (see k5_dit::sp_autotune)
// ---------------------------------------------------------------------------------------------
struct SpCand { int mode, slices, passes; const char* name; };

// pure host logic, exported for a CPU test (k5_sp_pick_schedule): candidate c costs max over ranks of times[r * ncand + c] (a collective
// schedule is as slow as its slowest rank); the cheapest valid one wins, ties to the lowest index (= the more conservative schedule);
// a non-finite or non-positive time disqualifies the candidate on every rank alike (all ranks see the same gathered table)
int sp_pick(const float* times, int ncand, int world, const int* valid, float* cost_out) {
  int best = -1; float best_t = 0.f;
  for (int c = 0; c < ncand; ++c) {
    float t = 0.f; bool ok = !valid || valid[c];
    for (int r = 0; r < world && ok; ++r) {
      const float v = times[(size_t)r * ncand + c];
      if (!(v > 0.f) || !(v < 3.0e38f)) ok = false; else t = v > t ? v : t;
    }
    if (cost_out) cost_out[c] = ok ? t : -1.f;
    if (ok && (best < 0 || t < best_t)) { best = c; best_t = t; }
  }
  return best;
}

int sp_autotune_run(k5_dit* d, hipStream_t s, int N, int L, const NablaArgs* nabla) {
  const int P = d->sp_world, H = d->Hh, D = d->D;
  d->sp_tuned = true;                                  // once per handle; the guard below takes it back if the run does not complete
  std::vector<SpCand> cands;
  if (!nabla) {
    cands.push_back({0, 1, 1, "K / V^T all-gather"});
    if (!(d->sp_user_set & 3) && d->comm.can_exchange() && N / 64 >= 4 * P) cands.push_back({0, 2, 1, "K / V^T exchange in 2 slices"});
    if (!(d->sp_user_set & 3) && H % P == 0 && d->comm.can_exchange()) cands.push_back({1, 1, 1, "Ulysses all-to-all"});
  } else {
    cands.push_back({0, 1, 1, "all-gather, one pass over the lists"});
    if (!(d->sp_user_set & 4)) cands.push_back({0, 1, 2, "all-gather, own key blocks first (two passes)"});
  }
  const int nc = (int)cands.size();
  if (nc < 2) { d->sp_report = "{\"tuned\": false, \"reason\": \"one admissible schedule\"}"; return K5_OK; }
  const int save_mode = d->sp_mode, save_slices = d->sp_slices, save_passes = d->sp_nabla_passes, save_prof = d->profiling;
  // every exit path — the failure returns inside the candidate loop included — leaves the handle as it found it: the caller's options and profiling
  // level, no leaked events, the RoPE cache invalidated (the trial runs zeroed the tables), and "not tuned" unless the run completed
  struct Guard {
    k5_dit* d; int mode, slices, passes, prof; hipEvent_t e0 = nullptr, e1 = nullptr; bool done = false;
    ~Guard() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      d->profiling = prof;
      d->key_vpos.clear(); d->key_shape[0] = d->key_shape[1] = d->key_shape[2] = 0;
      d->nabla_hint_pending = false;
      if (!done) { d->sp_mode = mode; d->sp_slices = slices; d->sp_nabla_passes = passes; d->sp_tuned = false; }
    }
  } guard{d, save_mode, save_slices, save_passes, save_prof};
  d->profiling = 0;
  unsigned long long cnt_save[4] = {0, 0, 0, 0};       // the trial launches must not show up in the softmax-form counters (k5_dit_attn_variant_counts)
  K5CHK(ensure_zeroed(d->ws_attn_cnt, 32, s));
  HIPCHK(hipMemcpyAsync(cnt_save, d->ws_attn_cnt.p, 32, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipEventCreate(&guard.e0)); HIPCHK(hipEventCreate(&guard.e1));
  const hipEvent_t e0 = guard.e0, e1 = guard.e1;
  std::vector<float> mine(nc, -1.f);
  double comm_ms = -1.0, comm_bytes = 0.0;
  const AttnW& a = d->vblocks[0].self_attn;
  for (int c = 0; c < nc; ++c) {
    d->sp_mode = cands[c].mode; d->sp_slices = cands[c].slices; d->sp_nabla_passes = cands[c].passes;
    const int S = d->sp_slices > 1 ? d->sp_slices : 1;
    const int n_pad = ((N / 64 + P * S - 1) / (P * S)) * S * 64, tok0 = d->sp_rank * n_pad;
    const int n = N - tok0 < n_pad ? N - tok0 : n_pad;
    if ((long long)(P - 1) * n_pad >= N || n <= 0) continue;       // this slot size leaves a rank without rows: not a candidate (time stays -1 everywhere)
    K5CHK(ensure_workspaces(d, P * n_pad, L));
    K5CHK(d->ws_q.ensure((size_t)n_pad * D * 2)); K5CHK(d->ws_kfull.ensure((size_t)P * n_pad * D * 2)); K5CHK(d->ws_vtfull.ensure((size_t)P * n_pad * D * 2));
    // finite, harmless operands: activations 0.1 (bf16 0x3dcd), rotary tables and gates zero (the out projection then adds nothing)
    HIPCHK(hipMemsetD16Async((hipDeviceptr_t)d->ws_h.p, 0x3dcd, (size_t)n_pad * D, s));
    HIPCHK(hipMemsetD16Async((hipDeviceptr_t)d->ws_vis.p, 0x3dcd, (size_t)n_pad * D, s));
    HIPCHK(hipMemsetAsync(d->ws_vcos.p, 0, (size_t)N * 32 * 4, s)); HIPCHK(hipMemsetAsync(d->ws_vsin.p, 0, (size_t)N * 32 * 4, s));
    HIPCHK(hipMemsetAsync(d->ws_mod.p, 0, d->mod_rows * 4, s));
    const float* cosT = d->ws_vcos.as<float>() + (size_t)tok0 * 32; const float* sinT = d->ws_vsin.as<float>() + (size_t)tok0 * 32;
    const bool uly = d->sp_mode == 1;
    // a line per candidate BEFORE it runs: if an exchange never used on this node hangs, the log says which ("sp_autotune" = 0 / K5_SP_AUTOTUNE=0 skip the tuning)
    if (d->sp_rank == 0) fprintf(stderr, "libk5: sequence-parallel schedule: timing candidate %d of %d: %s\n", c + 1, nc, cands[c].name);
    int trial_rc = K5_OK;   // a candidate that REFUSES its arguments does so on every rank alike (argument checks only): it is disqualified, not fatal
    for (int it = 0; it < 3 && trial_rc == K5_OK; ++it) {
      if (it == 1) HIPCHK(hipEventRecord(e0, s));
      trial_rc = uly ? run_self_attention_ulysses(d, s, a, d->ws_h.p, n, n_pad, N, d->ws_o.p, cosT, sinT, d->ws_vis.p, d->ws_mod.as<float>())
                     : run_self_attention_sp(d, s, a, d->ws_h.p, n, n_pad, N, d->ws_o.p, cosT, sinT, d->ws_vis.p, d->ws_mod.as<float>(), nabla);
    }
    if (trial_rc != K5_OK) {
      if (c == 0) return trial_rc;       // the default schedule itself fails: nothing to fall back to
      HIPCHK(hipStreamSynchronize(s));
      continue;
    }
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    mine[c] = ms / 2.f;
    if (c == 0) {   // the bare exchange of candidate 0 (both all-gathers back to back, nothing else running): achieved ingress bandwidth
      HIPCHK(hipEventRecord(e0, s));
      for (int it = 0; it < 2; ++it) {
        K5CHK(d->comm.all_gather_inplace(d->ws_kfull.p, (size_t)n_pad * D, 2, s));
        K5CHK(d->comm.all_gather_inplace(d->ws_vtfull.p, (size_t)D * n_pad, 2, s));
      }
      HIPCHK(hipEventRecord(e1, s)); HIPCHK(hipEventSynchronize(e1));
      HIPCHK(hipEventElapsedTime(&ms, e0, e1));
      comm_ms = ms / 2.0; comm_bytes = 2.0 * (double)(P - 1) * n_pad * D * 2.0;
    }
  }
  // every rank's table to every rank, then the same decision everywhere
  K5CHK(d->ws_tune.ensure((size_t)P * nc * 4));
  HIPCHK(hipMemcpyAsync(d->ws_tune.as<float>() + (size_t)d->sp_rank * nc, mine.data(), (size_t)nc * 4, hipMemcpyHostToDevice, s));
  K5CHK(d->comm.all_gather_inplace(d->ws_tune.p, (size_t)nc, 4, s));
  std::vector<float> all((size_t)P * nc), cost(nc);
  HIPCHK(hipMemcpyAsync(all.data(), d->ws_tune.p, all.size() * 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  const int best = sp_pick(all.data(), nc, P, nullptr, cost.data());
  // the trial runs used the handle's workspaces with made-up operands: put back what outlives a forward
  HIPCHK(hipMemcpyAsync(d->ws_attn_cnt.p, cnt_save, 32, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));                     // cnt_save is a local
  K5CHK(reset_attn_pref(d, s));
  guard.done = true;                                   // from here on the run counts as tuned (the guard still restores profiling and drops the RoPE cache)
  d->sp_mode = save_mode; d->sp_slices = save_slices; d->sp_nabla_passes = save_passes;
  if (best < 0) { d->sp_report = "{\"tuned\": false, \"reason\": \"no candidate ran\"}"; return K5_OK; }
  // only what this run VARIED is taken from the winner (a dense run never touches "sp_nabla_passes", a NABLA run never "sp_mode" / "sp_slices"),
  // and a knob the caller set is never a varied one (the candidate lists above leave it out)
  if (!nabla) { d->sp_mode = cands[best].mode; d->sp_slices = cands[best].slices; }
  else d->sp_nabla_passes = cands[best].passes;
  char buf[256];
  std::string rep = "{\"tuned\": true, \"world\": " + std::to_string(P) + ", \"tokens\": " + std::to_string(N) + ", \"attention\": \"" + (nabla ? "nabla" : "dense") +
                    "\", \"chosen\": \"" + cands[best].name + "\", \"unit\": \"ms per block's self-attention section, max over ranks\", \"candidates\": [";
  for (int c = 0; c < nc; ++c) {
    snprintf(buf, sizeof(buf), "%s{\"name\": \"%s\", \"ms\": %.4f, \"ms_this_rank\": %.4f}", c ? ", " : "", cands[c].name, cost[c], mine[c]);
    rep += buf;
  }
  snprintf(buf, sizeof(buf), "], \"gather_bytes_in_per_block\": %.0f, \"gather_ms_alone\": %.4f, \"gather_GBps_in\": %.1f}", comm_bytes, comm_ms,
           comm_ms > 0 ? comm_bytes / comm_ms * 1e-6 : 0.0);
  rep += buf;
  d->sp_report = rep;
  if (d->sp_rank == 0) fprintf(stderr, "libk5: sequence-parallel schedule: %s\n", rep.c_str());
  return K5_OK;
}

int forward_impl(k5_dit* d, const k5_forward_args* a, const k5_text_cond& cond, float time, const float* x,
                 int x_channels, void* out_velocity, hipStream_t s, const float* tvec = nullptr, const int* step = nullptr,
                 int text_slot = -1) {
  const k5_dit_config& c = d->cfg;
  if (!d->finalized) { k5_set_error("k5_dit_forward before k5_dit_finalize"); return K5_ERR_STATE; }
  if (a->attention_type == 1) {   // NABLA workgroup size of this forward (see nabla_group_rows)
    d->nabla_grp_now = d->nabla_group_rows ? d->nabla_group_rows : 4;
    if (d->nabla_group_rows == 0 && d->h_nabla_kept && d->nabla_hint_possible > 0) {
      const unsigned long long kept = *(volatile unsigned long long*)d->h_nabla_kept;   // whatever has landed: no synchronisation
      if (kept > 0 && (double)kept < 0.5 * (double)d->nabla_hint_possible) d->nabla_grp_now = 2;
    }
    d->nabla_hint_pending = true;
  }
  if (a->attention_type != 0 && a->attention_type != 1) { k5_set_error("attention_type must be 0 (flash) or 1 (nabla)"); return K5_ERR_ARG; }
  if (c.patch_size[0] != 1 || c.patch_size[1] != 2 || c.patch_size[2] != 2) return K5_ERR_UNSUPPORTED;
  const int Tp = a->T, Hp = a->H / 2, Wp = a->W / 2;
  const int N = Tp * Hp * Wp, L = cond.text_len, D = d->D;
  const int Cin = c.visual_cond ? 2 * c.in_visual_dim + 1 : c.in_visual_dim;
  if (N <= 0 || L <= 0 || (a->H & 1) || (a->W & 1)) { k5_set_error("bad shapes"); return K5_ERR_ARG; }
  if (x_channels != Cin && x_channels != c.in_visual_dim) { k5_set_error("x_channels must be %d or %d", Cin, c.in_visual_dim); return K5_ERR_ARG; }
  const int P = d->sp_world;
  const bool sp = d->comm.active();  // a communicator (even of size 1) selects the sharded code path
  if (sp && P > 1 && !d->emulated && d->sp_autotune && !d->sp_tuned && !d->vblocks.empty() && N % 64 == 0) {
    NablaArgs tna{};
    const bool tn = a->attention_type == 1 && !(Hp % 8) && !(Wp % 8);
    if (tn) tna = NablaArgs{Tp, Hp / 8, Wp / 8, a->nabla_wT, a->nabla_wH, a->nabla_wW, a->nabla_P};
    if (a->attention_type == 0 || tn) K5CHK(sp_autotune_run(d, s, N, L, tn ? &tna : nullptr));
  }
  // token shards: whole 64-token blocks, ceil(blocks / P) per rank; the last rank takes what is left (3660 blocks over 8
  // ranks: 7 x 458 + 454).  n_pad = slot size in the gather buffers, n = rows this rank really owns.
  int n = N, n_pad = N, tok0 = 0;
  if (sp) {
    if (N % 64) { k5_set_error("sequence parallelism needs whole 64-token blocks (token count %d)", N); return K5_ERR_UNSUPPORTED; }
    const int S = d->sp_slices > 1 ? d->sp_slices : 1;   // slots are whole slices (sliced K / V^T exchange, "sp_slices")
    n_pad = ((N / 64 + P * S - 1) / (P * S)) * S * 64;
    tok0 = d->sp_rank * n_pad;
    n = N - tok0 < n_pad ? N - tok0 : n_pad;
    if ((long long)(P - 1) * n_pad >= N) { k5_set_error("sequence parallel x%d: %d token blocks leave a rank without work", P, N / 64); return K5_ERR_UNSUPPORTED; }
  }
  K5CHK(ensure_workspaces(d, sp ? P * n_pad : N, L));
  if (sp) {
    K5CHK(d->ws_q.ensure((size_t)n_pad * D * 2)); K5CHK(d->ws_kfull.ensure((size_t)P * n_pad * D * 2));
    K5CHK(d->ws_vtfull.ensure((size_t)P * n_pad * D * 2));
  }
  // NABLA: tokens are processed in fractal order (8x8 spatial tiles contiguous), utils.py:31-41,54-78
  const bool nabla = a->attention_type == 1;
  d->last_nabla = nabla;
  NablaArgs na{};
  const int32_t* perm = nullptr;
  if (nabla) {
    if ((Hp % 8) || (Wp % 8)) { k5_set_error("nabla attention needs latent H, W divisible by 16 (got %d x %d)", a->H, a->W); return K5_ERR_ARG; }
    na = NablaArgs{Tp, Hp / 8, Wp / 8, a->nabla_wT, a->nabla_wH, a->nabla_wW, a->nabla_P};
    if (d->perm_shape[0] != Tp || d->perm_shape[1] != Hp || d->perm_shape[2] != Wp) {
      std::vector<int32_t> pv((size_t)N);
      const int Hb = Hp / 8, Wb = Wp / 8;
      for (int i = 0; i < N; ++i) {
        const int b = i >> 6, r = i & 63, t = b / (Hb * Wb), hb = (b / Wb) % Hb, wb = b % Wb;
        pv[i] = (t * Hp + hb * 8 + (r >> 3)) * Wp + wb * 8 + (r & 7);
      }
      K5CHK(d->ws_perm.ensure((size_t)N * 4));
      HIPCHK(hipMemcpy(d->ws_perm.p, pv.data(), (size_t)N * 4, hipMemcpyHostToDevice));
      d->perm_shape[0] = Tp; d->perm_shape[1] = Hp; d->perm_shape[2] = Wp;
    }
    perm = d->ws_perm.as<int32_t>();
  }
  K5CHK(prepare_rope(d, s, a, Tp, Hp, Wp, perm));
  const float *tcos = nullptr, *tsin = nullptr;
  K5CHK(prepare_text_rope(d, s, cond, &tcos, &tsin));

  const float* mod = d->ws_mod.as<float>();
  // ---- before_text_transformer_blocks (dit.py:129-137) ----
  {
    Scope sc(d, s, "prologue");
    k5_dit::TextCache* tc = text_slot >= 0 ? &d->text_cache[text_slot] : nullptr;
    if (tc && tc->valid && tc->L == L) {   // the text blocks update ws_text in place: restore their input
      HIPCHK(hipMemcpyAsync(d->ws_text.p, tc->text.p, (size_t)L * D * 2, hipMemcpyDeviceToDevice, s));
      HIPCHK(hipMemcpyAsync(d->ws_pool_f32.p, tc->pool.p, (size_t)d->TD * 4, hipMemcpyDeviceToDevice, s));
    } else {
      const void* text_bf; const void* pool_bf;
      K5CHK(to_bf16(d, s, cond.text_embed, cond.text_dtype, (size_t)L * c.in_text_dim, d->ws_text_in, &text_bf));
      K5CHK(to_bf16(d, s, cond.pooled_embed, cond.text_dtype, (size_t)c.in_text_dim2, d->ws_pool_in, &pool_bf));
      K5CHK(k5_launch_gemm_bf16(text_bf, d->text_w.p, d->text_b.as<float>(), d->ws_th.p, L, D, c.in_text_dim, c.in_text_dim,
                                c.in_text_dim, D, K5_EPI_BIAS, nullptr, 0, nullptr, s));
      K5CHK(k5_launch_ln_affine(d->ws_th.p, d->text_lnw.as<float>(), d->text_lnb.as<float>(), d->ws_text.p, nullptr, L, D, s));
      K5CHK(k5_launch_gemm_bf16(pool_bf, d->pool_w.p, d->pool_b.as<float>(), d->ws_pool_lin.p, 1, d->TD, c.in_text_dim2,
                                c.in_text_dim2, c.in_text_dim2, d->TD, K5_EPI_BIAS, nullptr, 0, nullptr, s));
      K5CHK(k5_launch_ln_affine(d->ws_pool_lin.p, d->pool_lnw.as<float>(), d->pool_lnb.as<float>(), nullptr,
                                d->ws_pool_f32.as<float>(), 1, d->TD, s));
      if (tc) {
        K5CHK(tc->text.ensure((size_t)L * D * 2)); K5CHK(tc->pool.ensure((size_t)d->TD * 4));
        HIPCHK(hipMemcpyAsync(tc->text.p, d->ws_text.p, (size_t)L * D * 2, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(tc->pool.p, d->ws_pool_f32.p, (size_t)d->TD * 4, hipMemcpyDeviceToDevice, s));
        tc->valid = true; tc->L = L;
      }
    }
    K5CHK(k5_launch_time_features(time, d->ws_tfeat.as<float>(), D, s, tvec, step));
    K5CHK(k5_launch_gemv_f32(d->ws_tfeat.as<float>(), d->time_w1.as<float>(), d->time_b1.as<float>(), d->ws_th1.as<float>(),
                             d->TD, D, 0, nullptr, s));
    K5CHK(k5_launch_gemv_f32(d->ws_th1.as<float>(), d->time_w2.as<float>(), d->time_b2.as<float>(), d->ws_temb.as<float>(),
                             d->TD, d->TD, 1, d->ws_pool_f32.as<float>(), s));
    // every Modulation layer of the network in one GEMV (they all consume the same time_embed)
    K5CHK(k5_launch_gemv_f32(d->ws_temb.as<float>(), d->mod_w.as<float>(), d->mod_b.as<float>(), d->ws_mod.as<float>(),
                             (int)d->mod_rows, d->TD, 1, nullptr, s));
    K5CHK(k5_launch_patchify(x, d->ws_xp.p, a->T, a->H, a->W, x_channels, Cin, d->KvisPad, perm, s));
    K5CHK(k5_launch_gemm_bf16(d->ws_xp.as<bf16_t>() + (size_t)tok0 * d->KvisPad, d->vis_w.p, d->vis_b.as<float>(), d->ws_vis.p,
                              n, D, d->KvisPad, d->KvisPad, d->KvisPad, D, K5_EPI_BIAS, nullptr, 0, nullptr, s));
  }
  // ---- text blocks (dit.py:170-171, 33-44) ----
  for (int i = 0; i < c.num_text_blocks; ++i) {
    const BlockW& b = d->tblocks[i];
    const float* m = mod + b.mod_off;
    K5CHK(ln_mod(d, s, d->ws_text.p, m, d->ws_th.p, L));
    K5CHK(run_self_attention(d, s, b.self_attn, d->ws_th.p, L, d->ws_tqk.p, d->ws_tvt.p, d->ws_to.p, tcos, tsin, d->ws_text.p, m + 2 * D, "attn_text"));
    K5CHK(ln_mod(d, s, d->ws_text.p, m + 3 * D, d->ws_th.p, L));
    K5CHK(run_ff(d, s, b, d->ws_th.p, L, d->ws_tff.p, d->ws_text.p, m + 5 * D));
  }
  // ---- MagCache decision (magcache_utils.py:59-76), float64 like the reference's numpy scalars ----
  auto& mg = d->mag;
  bool mag_skip = false;
  const int slot = mg.cnt & 1;
  const size_t vis_elems = (size_t)n * D;
  if (mg.on) {
    if (mg.cnt >= (int)((double)mg.table.size() * mg.retention)) {
      mg.acc_ratio[slot] *= mg.table[mg.cnt];
      mg.acc_steps[slot] += 1;
      mg.acc_err[slot] += std::fabs(1.0 - mg.acc_ratio[slot]);
      if (mg.acc_err[slot] < mg.thresh && mg.acc_steps[slot] <= mg.K) mag_skip = true;
      else { mg.acc_err[slot] = 0; mg.acc_steps[slot] = 0; mg.acc_ratio[slot] = 1.0; }
    }
    if (mag_skip) {
      if (mg.res_elems[slot] != vis_elems) { k5_set_error("magcache: skip decided but no cached residual of this shape for the slot"); return K5_ERR_ARG; }
      Scope sc(d, s, "elementwise");   // visual_embed + residual (bf16 add), :78-79
      K5CHK(k5_launch_gate_sum(d->ws_vis.p, mg.residual[slot].p, mg.pm_one.as<float>(), d->ws_vis.p, n, D, s));
      mg.n_skipped++;
    } else {
      K5CHK(mg.residual[slot].ensure(vis_elems * 2));
      HIPCHK(hipMemcpyAsync(mg.residual[slot].p, d->ws_vis.p, vis_elems * 2, hipMemcpyDeviceToDevice, s));  // ori_visual_embed
      mg.n_ran++;
    }
  }
  // ---- visual blocks (dit.py:176-178, 61-79) ----
  const bool cx_all = !mag_skip && d->cross_kv_batched && c.num_visual_blocks > 1 && d->cx_wk_all.p;
  if (cx_all) K5CHK(cross_kv_batched_run(d, s, d->ws_text.p, L));
  const int Lr8 = (int)rup(L, 8);
  for (int i = 0; i < (mag_skip ? 0 : c.num_visual_blocks); ++i) {
    const BlockW& b = d->vblocks[i];
    const float* m = mod + b.mod_off;
    const float* vcos = d->ws_vcos.as<float>() + (size_t)tok0 * 32;
    const float* vsin = d->ws_vsin.as<float>() + (size_t)tok0 * 32;
    const bool ulysses = sp && d->sp_mode == 1 && !nabla && d->sp_world > 1 && d->Hh % d->sp_world == 0 && !d->emulated;
    // fp8 modes: the LayerNorm writes the e4m3 rows the projections read (no bf16 h, no quantisation pass) — "fp8_fuse_ln" = 0 keeps the two passes
    const bool h8_sa = d->fp8_fuse_ln && !ulysses && (sp ? sa_sp_fp8_in(d, b.self_attn, n, n_pad, nabla) : sa_fp8_in(d, b.self_attn, n));
    K5CHK(ln_mod(d, s, d->ws_vis.p, m, d->ws_h.p, n, h8_sa));
    if (d->profiling) ++d->prof_self_blocks;   // bench.py: FLOPs of the roofline kernel = per-block FLOPs x the blocks that RAN
    if (ulysses) {
      K5CHK(run_self_attention_ulysses(d, s, b.self_attn, d->ws_h.p, n, n_pad, N, d->ws_o.p, vcos, vsin, d->ws_vis.p, m + 2 * D, text_slot > 0 ? 1 : 0));
    } else if (sp) {
      K5CHK(run_self_attention_sp(d, s, b.self_attn, d->ws_h.p, n, n_pad, N, d->ws_o.p, vcos, vsin, d->ws_vis.p, m + 2 * D, nabla ? &na : nullptr, text_slot > 0 ? 1 : 0, h8_sa));
    } else {
      K5CHK(run_self_attention(d, s, b.self_attn, d->ws_h.p, n, d->ws_qk.p, d->ws_vt.p, d->ws_o.p, vcos, vsin, d->ws_vis.p,
                               m + 2 * D, "attn_self", nabla ? &na : nullptr, text_slot > 0 ? 1 : 0, h8_sa));
    }
    K5CHK(ln_mod(d, s, d->ws_vis.p, m + 3 * D, d->ws_h.p, n));
    if (cx_all)
      K5CHK(run_cross_attention(d, s, b.cross_attn, d->ws_h.p, n, d->ws_text.p, L, d->ws_qk.p, d->ws_ck_all.as<bf16_t>() + (size_t)i * D,
                                d->ws_cvt_all.as<bf16_t>() + (size_t)i * D * Lr8, d->ws_o.p, d->ws_vis.p, m + 5 * D, true, c.num_visual_blocks * D));
    else
      K5CHK(run_cross_attention(d, s, b.cross_attn, d->ws_h.p, n, d->ws_text.p, L, d->ws_qk.p, d->ws_ck.p, d->ws_cvt.p,
                                d->ws_o.p, d->ws_vis.p, m + 5 * D));
    const bool h8_ff = d->fp8_fuse_ln && ff_fp8_in(d, b, n);
    K5CHK(ln_mod(d, s, d->ws_vis.p, m + 6 * D, d->ws_h.p, n, h8_ff));
    K5CHK(run_ff(d, s, b, d->ws_h.p, n, d->ws_ff.p, d->ws_vis.p, m + 8 * D, h8_ff));
  }
  if (mg.on) {
    if (!mag_skip) {  // residual = visual_embed - ori_visual_embed (bf16), :84
      Scope sc(d, s, "elementwise");
      K5CHK(k5_launch_gate_sum(d->ws_vis.p, mg.residual[slot].p, mg.pm_one.as<float>() + D, mg.residual[slot].p, n, D, s));
      mg.res_elems[slot] = vis_elems;
    }
    mg.cnt += mg.stride;   // :91-100
    if (mg.cnt >= (int)mg.table.size()) {
      mg.cnt = mg.first;
      for (int j = 0; j < 2; ++j) { mg.acc_ratio[j] = 1.0; mg.acc_err[j] = 0; mg.acc_steps[j] = 0; }
    }
  }
  // ---- after_blocks / OutLayer (dit.py:149-153, nn.py:374-400) ----
  {
    Scope sc(d, s, "epilogue");
    const float* m = mod + d->out_mod_off;  // [shift | scale]
    K5CHK(k5_launch_ln_modulate(d->ws_vis.p, m + D, m, d->ws_h.p, n, D, D, D, s));
    K5CHK(k5_launch_gemm_bf16(d->ws_h.p, d->out_w.p, d->out_b.as<float>(), d->ws_y.as<bf16_t>() + (size_t)tok0 * d->Fout, n,
                              d->Fout, D, D, D, d->Fout, K5_EPI_BIAS, nullptr, 0, nullptr, s));
    // every rank needs the whole velocity: the (replicated) latent is advanced identically on all ranks
    if (sp) K5CHK(d->comm.all_gather_inplace(d->ws_y.p, (size_t)n_pad * d->Fout, 2, s));
    K5CHK(k5_launch_unpatchify(d->ws_y.p, out_velocity, Tp, Hp, Wp, c.out_visual_dim, d->Fout, perm, s));
  }
  return K5_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI: engine
// ---------------------------------------------------------------------------------------------
extern "C" int k5_dit_create(const k5_dit_config* cfg, k5_dit** out) {
  g_err[0] = 0;
  if (!cfg || !out) return K5_ERR_ARG;
  const int hd = cfg->axes_dims[0] + cfg->axes_dims[1] + cfg->axes_dims[2];
  if (hd != 64) { k5_set_error("head_dim = sum(axes_dims) must be 64 (got %d)", hd); return K5_ERR_UNSUPPORTED; }
  if (cfg->model_dim % 64 || cfg->model_dim % 8 || cfg->ff_dim % 8 || cfg->time_dim % 8 || cfg->in_text_dim % 8 ||
      cfg->in_text_dim2 % 8) {
    k5_set_error("model_dim %% 64, ff_dim/time_dim/in_text_dim(2) %% 8 must be 0"); return K5_ERR_UNSUPPORTED;
  }
  k5_dit* d = new k5_dit();
  d->cfg = *cfg;
  d->D = cfg->model_dim; d->FF = cfg->ff_dim; d->TD = cfg->time_dim; d->Hh = cfg->model_dim / 64;
  const int Cin = cfg->visual_cond ? 2 * cfg->in_visual_dim + 1 : cfg->in_visual_dim;
  d->Kvis = cfg->patch_size[0] * cfg->patch_size[1] * cfg->patch_size[2] * Cin;
  d->KvisPad = (int)rup(d->Kvis, 8);
  d->Fout = cfg->patch_size[0] * cfg->patch_size[1] * cfg->patch_size[2] * cfg->out_visual_dim;
  expected_keys(*cfg, d->expected);
  *out = d;
  return K5_OK;
}

extern "C" void k5_dit_destroy(k5_dit* d) {
  if (!d) return;
  // DevBufs are released with the process; explicit frees for long-lived hosts:
  DevBuf* all[] = {&d->time_w1, &d->time_b1, &d->time_w2, &d->time_b2, &d->text_w, &d->text_b, &d->text_lnw, &d->text_lnb,
                   &d->pool_w, &d->pool_b, &d->pool_lnw, &d->pool_lnb, &d->vis_w, &d->vis_b, &d->out_w, &d->out_b, &d->mod_w,
                   &d->mod_b, &d->ws_text_in, &d->ws_text, &d->ws_th, &d->ws_tqk, &d->ws_tvt, &d->ws_to, &d->ws_tff,
                   &d->ws_pool_in, &d->ws_pool_lin, &d->ws_pool_f32, &d->ws_time, &d->ws_tfeat, &d->ws_th1, &d->ws_temb,
                   &d->ws_mod, &d->ws_xp, &d->ws_vis, &d->ws_h, &d->ws_qk, &d->ws_vt, &d->ws_o, &d->ws_ff, &d->ws_ck,
                   &d->ws_cvt, &d->ws_y, &d->ws_vcos, &d->ws_vsin, &d->ws_pos, &d->ws_vel_c,
                   &d->ws_vel_u};
  for (DevBuf* b : all) b->release();
  for (auto& e : d->text_rope) { e.cosT.release(); e.sinT.release(); e.pos.release(); }
  d->ws_q.release(); d->ws_kfull.release(); d->ws_vtfull.release(); d->ws_attn_state.release(); d->ws_attn_bal.release(); d->ws_kc.release(); d->ws_kmeans.release(); d->ws_nabla_kept.release(); if (d->h_nabla_kept) { (void)hipHostFree(d->h_nabla_kept); d->h_nabla_kept = nullptr; } d->ws_sched.release(); d->ws_h8.release(); d->ws_ff8.release();
  d->ws_attn_stats.release(); d->ws_attn_flags.release(); d->ws_attn_cnt.release(); d->ws_attn_part.release(); d->ws_leave_sig.release();
  if (d->h_leave_sig) { (void)hipHostFree(d->h_leave_sig); d->h_leave_sig = nullptr; }
  for (auto& t : d->text_cache) { t.text.release(); t.pool.release(); }
  for (auto& kv : d->staged) kv.second.dev.release();   // a handle destroyed before finalize still holds its staged matrices
  d->mag.residual[0].release(); d->mag.residual[1].release(); d->mag.pm_one.release();
  for (auto& b : d->vblocks) { b.w1_f8.release(); b.w2_f8.release(); b.s1_f8.release(); b.s2_f8.release(); }
  if (d->graph_stream) { (void)hipStreamSynchronize(d->graph_stream); (void)hipStreamDestroy(d->graph_stream); (void)hipEventDestroy(d->ev_graph); }
  if (d->comm_stream) { (void)hipStreamSynchronize(d->comm_stream); (void)hipStreamDestroy(d->comm_stream); }
  for (hipEvent_t e : {d->ev_k, d->ev_v, d->ev_gathered, d->ev_stats, d->ev_means, d->ev_slice[0], d->ev_slice[1], d->ev_slice[2], d->ev_slice[3]}) if (e) (void)hipEventDestroy(e); d->ws_perm.release(); d->ws_nabla.release();
  if (d->comm.comm) (void)d->comm.CommDestroy(d->comm.comm);
  for (Comm* c : {&d->comm, &d->pair}) if (c->ipc) { c->ipc->close_all(); delete c->ipc; c->ipc = nullptr; }
  if (d->pair_stream) { (void)hipStreamSynchronize(d->pair_stream); (void)hipStreamDestroy(d->pair_stream); }
  for (hipEvent_t e : {d->ev_vel_ready, d->ev_vel_done, d->ev_u_o, d->ev_u_back}) if (e) (void)hipEventDestroy(e);
  if (d->pair.comm) (void)d->pair.CommDestroy(d->pair.comm);
  d->ws_vel_pair.release();
  auto rel_attn = [](AttnW& a) {
    DevBuf* bs[] = {&a.wqk, &a.wq, &a.wk, &a.wv, &a.wo, &a.bqk, &a.bq, &a.bk, &a.bv, &a.bo, &a.norm};
    for (DevBuf* b : bs) b->release();
  };
  for (auto* v : {&d->tblocks, &d->vblocks})
    for (auto& b : *v) { rel_attn(b.self_attn); rel_attn(b.cross_attn); b.w1.release(); b.w2.release(); }
  delete d;
}

extern "C" int k5_dit_load_tensor(k5_dit* d, const char* key, const void* host_ptr, int dtype, const int64_t* shape, int rank) {
  g_err[0] = 0;
  if (!d || !key || !host_ptr || rank < 1 || rank > 2) return K5_ERR_ARG;
  if (d->finalized) { k5_set_error("load_tensor after finalize"); return K5_ERR_STATE; }
  bool known = false;
  for (auto& e : d->expected) if (e == key) { known = true; break; }
  if (!known) { k5_set_error("unexpected key in state_dict: %s", key); return K5_ERR_KEY; }
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < rank; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
  if (dtype != K5_F32 && dtype != K5_BF16 && dtype != K5_F16) return K5_ERR_ARG;
  if (rank == 2) {   // a matrix: raw copy to the device (D2D for a checkpoint that is already there), packed by k5_dit_finalize
    const size_t bytes = n * (dtype == K5_F32 ? 4 : 2);
    t.dtype = dtype;
    K5CHK(t.dev.ensure(bytes));
    HIPCHK(hipMemcpy(t.dev.p, host_ptr, bytes, hipMemcpyDefault));
    d->staged[key] = std::move(t);
    return K5_OK;
  }
  t.data.resize(n);
  // hipMemcpyDefault: the source may be host memory (safetensors on CPU) or device memory
  if (dtype == K5_F32) HIPCHK(hipMemcpy(t.data.data(), host_ptr, n * 4, hipMemcpyDefault));
  else if (dtype == K5_BF16 || dtype == K5_F16) {
    std::vector<uint16_t> raw(n);
    HIPCHK(hipMemcpy(raw.data(), host_ptr, n * 2, hipMemcpyDefault));
    if (dtype == K5_BF16)
      for (size_t i = 0; i < n; ++i) { uint32_t u = (uint32_t)raw[i] << 16; memcpy(&t.data[i], &u, 4); }
    else
      for (size_t i = 0; i < n; ++i) t.data[i] = half_to_float(raw[i]);
  } else return K5_ERR_ARG;
  d->staged[key] = std::move(t);
  return K5_OK;
}

extern "C" int k5_dit_missing_keys(k5_dit* d) {
  if (!d) return -1;
  int miss = 0; std::string names;
  for (auto& e : d->expected)
    if (!d->staged.count(e)) { if (miss < 8) names += e + " "; ++miss; }
  if (miss) k5_set_error("missing %d key(s): %s", miss, names.c_str());
  return d->finalized ? 0 : miss;
}

extern "C" int k5_dit_finalize(k5_dit* d) {
  g_err[0] = 0;
  if (!d) return K5_ERR_ARG;
  if (d->finalized) return K5_OK;
  if (k5_dit_missing_keys(d)) return K5_ERR_KEY;
  const k5_dit_config& c = d->cfg;
  const size_t D = d->D, FF = d->FF, TD = d->TD;
  auto shape_is = [&](const char* k, size_t r, size_t cc) -> bool {
    const HostTensor* t = find(d, k);
    const bool ok = t && ((cc == 0 && t->shape.size() == 1 && (size_t)t->shape[0] == r) ||
                          (cc && t->shape.size() == 2 && (size_t)t->shape[0] == r && (size_t)t->shape[1] == cc));
    if (!ok) k5_set_error("shape mismatch for %s", k);
    return ok;
  };
  if (!shape_is("time_embeddings.in_layer.weight", TD, D) || !shape_is("time_embeddings.out_layer.weight", TD, TD) ||
      !shape_is("text_embeddings.in_layer.weight", D, c.in_text_dim) ||
      !shape_is("pooled_text_embeddings.in_layer.weight", TD, c.in_text_dim2) ||
      !shape_is("visual_embeddings.in_layer.weight", D, d->Kvis) || !shape_is("out_layer.out_layer.weight", d->Fout, D) ||
      !shape_is("out_layer.modulation.out_layer.weight", 2 * D, TD) ||
      !shape_is("time_embeddings.in_layer.bias", TD, 0) || !shape_is("time_embeddings.out_layer.bias", TD, 0) ||
      !shape_is("text_embeddings.in_layer.bias", D, 0) || !shape_is("text_embeddings.norm.weight", D, 0) ||
      !shape_is("text_embeddings.norm.bias", D, 0) || !shape_is("pooled_text_embeddings.in_layer.bias", TD, 0) ||
      !shape_is("pooled_text_embeddings.norm.weight", TD, 0) || !shape_is("pooled_text_embeddings.norm.bias", TD, 0) ||
      !shape_is("visual_embeddings.in_layer.bias", D, 0) || !shape_is("out_layer.out_layer.bias", d->Fout, 0))
    return K5_ERR_ARG;
  K5CHK(pack_f32(d->time_w1, find(d, "time_embeddings.in_layer.weight"), TD, D));
  K5CHK(upload_f32(d->time_b1, find(d, "time_embeddings.in_layer.bias")->data.data(), TD));
  K5CHK(pack_f32(d->time_w2, find(d, "time_embeddings.out_layer.weight"), TD, TD));
  K5CHK(upload_f32(d->time_b2, find(d, "time_embeddings.out_layer.bias")->data.data(), TD));
  K5CHK(pack_bf16(d->text_w, find(d, "text_embeddings.in_layer.weight"), D, c.in_text_dim, c.in_text_dim));
  K5CHK(upload_bias_bf16r(d->text_b, find(d, "text_embeddings.in_layer.bias")->data.data(), D));
  K5CHK(upload_f32(d->text_lnw, find(d, "text_embeddings.norm.weight")->data.data(), D));
  K5CHK(upload_f32(d->text_lnb, find(d, "text_embeddings.norm.bias")->data.data(), D));
  K5CHK(pack_bf16(d->pool_w, find(d, "pooled_text_embeddings.in_layer.weight"), TD, c.in_text_dim2, c.in_text_dim2));
  K5CHK(upload_bias_bf16r(d->pool_b, find(d, "pooled_text_embeddings.in_layer.bias")->data.data(), TD));
  K5CHK(upload_f32(d->pool_lnw, find(d, "pooled_text_embeddings.norm.weight")->data.data(), TD));
  K5CHK(upload_f32(d->pool_lnb, find(d, "pooled_text_embeddings.norm.bias")->data.data(), TD));
  K5CHK(pack_bf16(d->vis_w, find(d, "visual_embeddings.in_layer.weight"), D, d->Kvis, d->KvisPad));
  K5CHK(upload_bias_bf16r(d->vis_b, find(d, "visual_embeddings.in_layer.bias")->data.data(), D));
  K5CHK(pack_bf16(d->out_w, find(d, "out_layer.out_layer.weight"), d->Fout, D, D));
  K5CHK(upload_bias_bf16r(d->out_b, find(d, "out_layer.out_layer.bias")->data.data(), d->Fout));

  // stacked modulation
  d->mod_rows = (size_t)c.num_text_blocks * 6 * D + (size_t)c.num_visual_blocks * 9 * D + 2 * D;
  std::vector<float> mb(d->mod_rows);
  size_t off = 0;
  auto put_mod = [&](const std::string& p, size_t rows) -> bool {
    const HostTensor *w = find(d, p + ".weight"), *b = find(d, p + ".bias");
    if (!w || !b || w->shape.size() != 2 || w->count() != rows * TD || b->data.size() != rows) { k5_set_error("shape mismatch for %s", p.c_str()); return false; }
    if (pack_f32(d->mod_w, w, rows, TD, off, d->mod_rows) != K5_OK) return false;
    memcpy(mb.data() + off, b->data.data(), rows * 4);
    off += rows;
    return true;
  };
  d->tblocks.resize(c.num_text_blocks); d->vblocks.resize(c.num_visual_blocks);
  for (int i = 0; i < c.num_text_blocks; ++i) {
    const std::string p = "text_transformer_blocks." + std::to_string(i);
    d->tblocks[i].mod_off = off;
    if (!put_mod(p + ".text_modulation.out_layer", 6 * D)) return K5_ERR_ARG;
    K5CHK(pack_attn(d, p + ".self_attention", d->tblocks[i].self_attn, true));
    if (!shape_ok(d, p + ".feed_forward.in_layer.weight", FF, D) || !shape_ok(d, p + ".feed_forward.out_layer.weight", D, FF)) return K5_ERR_ARG;
    K5CHK(pack_bf16(d->tblocks[i].w1, find(d, p + ".feed_forward.in_layer.weight"), FF, D, D));
    K5CHK(pack_bf16(d->tblocks[i].w2, find(d, p + ".feed_forward.out_layer.weight"), D, FF, FF));
  }
  for (int i = 0; i < c.num_visual_blocks; ++i) {
    const std::string p = "visual_transformer_blocks." + std::to_string(i);
    d->vblocks[i].mod_off = off;
    if (!put_mod(p + ".visual_modulation.out_layer", 9 * D)) return K5_ERR_ARG;
    K5CHK(pack_attn(d, p + ".self_attention", d->vblocks[i].self_attn, true));
    K5CHK(pack_attn(d, p + ".cross_attention", d->vblocks[i].cross_attn, false));
    if (!shape_ok(d, p + ".feed_forward.in_layer.weight", FF, D) || !shape_ok(d, p + ".feed_forward.out_layer.weight", D, FF)) return K5_ERR_ARG;
    K5CHK(pack_bf16(d->vblocks[i].w1, find(d, p + ".feed_forward.in_layer.weight"), FF, D, D));
    K5CHK(pack_bf16(d->vblocks[i].w2, find(d, p + ".feed_forward.out_layer.weight"), D, FF, FF));
  }
  {   // stacked cross-attention key / value weights of the visual blocks (device-to-device copies of the packed per-block buffers)
    const size_t nbv = (size_t)c.num_visual_blocks;
    if (nbv) {
      K5CHK(d->cx_wk_all.ensure(nbv * D * D * 2)); K5CHK(d->cx_wv_all.ensure(nbv * D * D * 2));
      K5CHK(d->cx_bk_all.ensure(nbv * D * 4)); K5CHK(d->cx_bv_all.ensure(nbv * D * 4)); K5CHK(d->cx_knorm_all.ensure(nbv * 64 * 4));
      bool complete = true;
      for (size_t i = 0; i < nbv && complete; ++i) {
        const AttnW& a = d->vblocks[i].cross_attn;
        complete = a.wk.p && a.wv.p && a.bk.p && a.bv.p && a.norm.p;
      }
      if (!complete) {
        d->cross_kv_batched = 0;
        d->cx_wk_all.release(); d->cx_wv_all.release(); d->cx_bk_all.release(); d->cx_bv_all.release(); d->cx_knorm_all.release();
      }
      // the stacked copy REPLACES the per-block buffers (they become views into it) instead of doubling them — 4 nbv D^2 bytes, 411 MB on
      // 2B-Lite; the per-block path ("cross_kv_batched" = 0) reads the same memory through the views
      for (size_t i = 0; i < nbv && complete; ++i) {
        AttnW& a = d->vblocks[i].cross_attn;
        char* wk = (char*)d->cx_wk_all.p + i * D * D * 2; char* wv = (char*)d->cx_wv_all.p + i * D * D * 2;
        char* bk = (char*)d->cx_bk_all.p + i * D * 4; char* bv = (char*)d->cx_bv_all.p + i * D * 4;
        HIPCHK(hipMemcpy(wk, a.wk.p, D * D * 2, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(wv, a.wv.p, D * D * 2, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(bk, a.bk.p, D * 4, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(bv, a.bv.p, D * 4, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy((char*)d->cx_knorm_all.p + i * 64 * 4, a.norm.as<float>() + 64, 64 * 4, hipMemcpyDeviceToDevice));
        a.wk.alias(wk, D * D * 2); a.wv.alias(wv, D * D * 2); a.bk.alias(bk, D * 4); a.bv.alias(bv, D * 4);
      }
    }
  }
  d->out_mod_off = off;
  if (!put_mod("out_layer.modulation.out_layer", 2 * D)) return K5_ERR_ARG;
  K5CHK(upload_f32(d->mod_b, mb.data(), mb.size()));
  HIPCHK(hipDeviceSynchronize());   // the pack kernels read the staged device copies that are freed next
  for (auto& kv : d->staged) kv.second.dev.release();
  d->staged.clear();
  // the NABLA density hint's device counter and pinned host copy exist before any forward (no allocation inside a stream capture)
  K5CHK(d->ws_nabla_kept.ensure(8));
  if (!d->h_nabla_kept) { HIPCHK(hipHostMalloc((void**)&d->h_nabla_kept, 8, hipHostMallocDefault)); *d->h_nabla_kept = 0ull; }
  d->finalized = true;
  return K5_OK;
}

extern "C" int k5_dit_set_magcache(k5_dit* d, const double* ratio_table, int table_len, int no_cfg, double thresh, int K,
                                   double retention_ratio) {
  if (!d || !d->finalized) { k5_set_error("k5_dit_set_magcache: handle not finalized"); return K5_ERR_STATE; }
  auto& mg = d->mag;
  mg.cnt = 0; mg.n_ran = mg.n_skipped = 0;
  for (int j = 0; j < 2; ++j) { mg.acc_ratio[j] = 1.0; mg.acc_err[j] = 0; mg.acc_steps[j] = 0; mg.res_elems[j] = 0; }
  if (table_len <= 0) { mg.on = false; mg.table.clear(); return K5_OK; }
  if (!ratio_table || (table_len & 1) || K < 0) { k5_set_error("k5_dit_set_magcache: table of 2*num_steps ratios expected"); return K5_ERR_ARG; }
  mg.table.assign(ratio_table, ratio_table + table_len);
  mg.no_cfg = no_cfg != 0; mg.thresh = thresh; mg.K = K; mg.retention = retention_ratio;
  mg.first = 0; mg.stride = mg.no_cfg ? 2 : 1;
  std::vector<float> pm(2 * (size_t)d->D);
  for (int i = 0; i < d->D; ++i) { pm[i] = 1.f; pm[d->D + i] = -1.f; }
  K5CHK(mg.pm_one.ensure(pm.size() * 4));
  HIPCHK(hipMemcpy(mg.pm_one.p, pm.data(), pm.size() * 4, hipMemcpyHostToDevice));
  mg.on = true;
  return K5_OK;
}

extern "C" int k5_dit_magcache_calls(k5_dit* d, int first_call, int stride) {
  if (!d || !d->mag.on) { k5_set_error("k5_dit_magcache_calls: MagCache is not enabled on this handle"); return K5_ERR_STATE; }
  if (first_call < 0 || stride < 1 || first_call >= stride) { k5_set_error("k5_dit_magcache_calls: need 0 <= first < stride"); return K5_ERR_ARG; }
  d->mag.first = d->mag.cnt = first_call; d->mag.stride = stride;
  return K5_OK;
}

extern "C" int k5_dit_magcache_state(k5_dit* d, int* cnt, long long* n_ran, long long* n_skipped) {
  if (!d) { k5_set_error("k5_dit_magcache_state: null handle"); return K5_ERR_ARG; }
  if (cnt) *cnt = d->mag.cnt;
  if (n_ran) *n_ran = d->mag.n_ran;
  if (n_skipped) *n_skipped = d->mag.n_skipped;
  return K5_OK;
}

extern "C" int k5_dit_forward(k5_dit* d, const k5_forward_args* a, void* out_velocity, void* stream) {
  g_err[0] = 0;
  if (!d || !a || !out_velocity || !a->x) return K5_ERR_ARG;
  const int st = forward_impl(d, a, a->cond, a->time, a->x, a->x_channels, out_velocity, (hipStream_t)stream);
  return st;
}

extern "C" int k5_sample(k5_dit* d, const k5_sample_args* a, void* stream) {
  g_err[0] = 0;
  if (!d || !a || !a->latent || !a->sigmas || a->num_steps <= 0) return K5_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const k5_dit_config& c = d->cfg;
  const int64_t n = (int64_t)a->fwd.T * a->fwd.H * a->fwd.W * c.out_visual_dim;
  if (c.in_visual_dim != c.out_visual_dim) return K5_ERR_UNSUPPORTED;
  const bool cfg_on = fabsf(a->guidance_weight - 1.0f) > 1e-6f;  // generation_utils.py:63
  const bool pair = cfg_on && d->pair.active();                  // CFG-parallel: one branch here, the other on the paired handle
  if (pair) K5CHK(d->ws_vel_pair.ensure(2 * n * 2));
  else {
    K5CHK(d->ws_vel_c.ensure(n * 2));
    if (cfg_on) K5CHK(d->ws_vel_u.ensure(n * 2));
  }
  void* vel_c = pair ? d->ws_vel_pair.p : d->ws_vel_c.p;
  void* vel_u = pair ? (void*)(d->ws_vel_pair.as<bf16_t>() + n) : d->ws_vel_u.p;
  // hipGraph mode (BASELINE config 5 "hipGraph-captured step"): the per-step scalars live in device tables indexed by a
  // device-side step counter, so ONE captured step (forward(s) + CFG/Euler + counter increment) replays for every step.
  // Step 0 runs eagerly (it sizes every workspace and fills the RoPE / permutation caches), step 1 is captured, steps 1..
  // are launches of the instantiated graph.  Not with MagCache (its skip pattern changes the launch sequence per step)
  // or while profiling (events).
  // not replayed: MagCache (host decisions), profiling (events), the NABLA map tap (its destination advances on the host per launch: a replay
  // would overwrite the slots baked at capture — ADVICE r5), loopback groups (their collectives are host rendezvous).  The IPC transport IS
  // replayable: its epochs live on the device and the peers' buffers were mapped by the eager first step (ipc_comm.h)
  const bool graph = d->use_graph && !d->mag.on && !d->profiling && a->num_steps > 2 && !d->nabla_tap && !d->comm.loop && !d->pair.loop;
  std::vector<float> host_tab(2 * (size_t)a->num_steps);
  for (int i = 0; i < a->num_steps; ++i) {
    host_tab[i] = a->sigmas[i] * 1000.0f;                        // t * 1000, fp32 (:57)
    host_tab[a->num_steps + i] = a->sigmas[i + 1] - a->sigmas[i];  // torch.diff(timesteps) (:105)
  }
  const float* tvec = nullptr; const float* dtvec = nullptr; int* step = nullptr;
  hipStream_t caller = s;
  if (graph) {
    if (!d->graph_stream) {
      HIPCHK(hipStreamCreateWithFlags(&d->graph_stream, hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&d->ev_graph, hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(d->ev_graph, caller));          // everything the caller enqueued so far happens before the sampler
    s = d->graph_stream;
    HIPCHK(hipStreamWaitEvent(s, d->ev_graph, 0));
    K5CHK(d->ws_sched.ensure(host_tab.size() * 4 + 16));
    HIPCHK(hipMemcpyAsync(d->ws_sched.p, host_tab.data(), host_tab.size() * 4, hipMemcpyHostToDevice, s));
    step = reinterpret_cast<int*>(d->ws_sched.as<float>() + host_tab.size());
    HIPCHK(hipMemsetAsync(step, 0, 4, s));
    HIPCHK(hipStreamSynchronize(s));   // host_tab is a local
    tvec = d->ws_sched.as<float>(); dtvec = tvec + a->num_steps;
  }
  d->text_cache[0].valid = d->text_cache[1].valid = false;   // the prompt tensors are constant for THIS call only
  K5CHK(reset_attn_pref(d, s));                              // ... and so is what the softmax-form memory of the layers is worth
  // where the visual queries are normalised (see k5_dit::fuse_qnorm_auto): a plain one-handle dense run decides after its first step
  struct FuseGuard { k5_dit* d; ~FuseGuard() { d->fuse_now = false; d->leave_collect = false; } } fuse_guard{d};
  d->fuse_now = false; d->leave_collect = false; d->fuse_last = -1;
  const bool auto_fuse = d->fuse_qnorm_auto && !d->fuse_qnorm && !d->comm.active() && d->attn_mode == K5_ATTN_AUTO && d->row_offsets &&   // (a CFG pair's handles decide each for itself)
                         a->fwd.attention_type == 0 && a->num_steps >= 2;
  if (auto_fuse) {
    K5CHK(d->ws_leave_sig.ensure(4));
    if (!d->h_leave_sig) HIPCHK(hipHostMalloc((void**)&d->h_leave_sig, 4, hipHostMallocDefault));
    HIPCHK(hipMemsetAsync(d->ws_leave_sig.p, 0, 4, s));
    d->leave_collect = true;
  }
  auto decide_fuse = [&]() -> int {   // after step 0
    if (!auto_fuse) return K5_OK;
    d->leave_collect = false;
    *d->h_leave_sig = 1u;
    HIPCHK(hipMemcpyAsync(d->h_leave_sig, d->ws_leave_sig.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));   // once per call: the host runs a step ahead otherwise
    d->fuse_now = *d->h_leave_sig == 0u;
    d->fuse_last = d->fuse_now ? 1 : 0;
    return K5_OK;
  };
  auto one_step = [&](int i) -> int {
    const float t1000 = host_tab[i], dt = host_tab[a->num_steps + i];
    if (pair) {
      // my branch only; then the pair's velocities change hands on the side stream (fork / join by events: capturable) and both
      // handles hold [v_cond | v_uncond] — every rank of both groups applies the same update to the same numbers
      if (d->cfg_branch == 0) K5CHK(forward_impl(d, &a->fwd, a->fwd.cond, t1000, a->latent, c.in_visual_dim, vel_c, s, tvec, step, 0));
      else K5CHK(forward_impl(d, &a->fwd, a->null_cond, t1000, a->latent, c.in_visual_dim, vel_u, s, tvec, step, 1));
      Scope sc(d, s, "comm");
      HIPCHK(hipEventRecord(d->ev_vel_ready, s));
      HIPCHK(hipStreamWaitEvent(d->pair_stream, d->ev_vel_ready, 0));
      K5CHK(d->pair.all_gather_inplace(d->ws_vel_pair.p, (size_t)n, 2, d->pair_stream));
      HIPCHK(hipEventRecord(d->ev_vel_done, d->pair_stream));
      HIPCHK(hipStreamWaitEvent(s, d->ev_vel_done, 0));
    } else {
      K5CHK(forward_impl(d, &a->fwd, a->fwd.cond, t1000, a->latent, c.in_visual_dim, vel_c, s, tvec, step, 0));
      if (cfg_on) K5CHK(forward_impl(d, &a->fwd, a->null_cond, t1000, a->latent, c.in_visual_dim, vel_u, s, tvec, step, 1));
    }
    {
      Scope sc(d, s, "elementwise");
      K5CHK(k5_launch_cfg_euler(a->latent, vel_c, cfg_on ? vel_u : nullptr, a->guidance_weight, dt, n, s, dtvec, step));
    }
    if (step) K5CHK(k5_launch_step_inc(step, s));
    return K5_OK;
  };
  if (!graph) {
    for (int i = 0; i < a->num_steps; ++i) {
      K5CHK(one_step(i));
      if (i == 0) K5CHK(decide_fuse());
    }
    return K5_OK;
  }
  K5CHK(one_step(0));
  K5CHK(decide_fuse());
  hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
  HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  const int rc = one_step(1);
  const hipError_t ec = hipStreamEndCapture(s, &g);
  if (rc != K5_OK || ec != hipSuccess || !g) {
    if (g) (void)hipGraphDestroy(g);
    if (rc == K5_OK) k5_set_error("hipStreamEndCapture: %s", hipGetErrorString(ec));
    return rc != K5_OK ? rc : K5_ERR_HIP;
  }
  if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { (void)hipGraphDestroy(g); k5_set_error("hipGraphInstantiate failed"); return K5_ERR_HIP; }
  int status = K5_OK;
  for (int i = 1; i < a->num_steps && status == K5_OK; ++i)
    if (hipGraphLaunch(ge, s) != hipSuccess) { k5_set_error("hipGraphLaunch failed"); status = K5_ERR_HIP; }
  (void)hipStreamSynchronize(s);   // the executable graph must outlive its launches (this also orders the caller's stream after us)
  (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
  (void)caller;
  return status;
}

// W8A8 e4m3 feed-forward (BASELINE config 5).  Quantises W1 / W2 of every visual block per output channel on first enable.
extern "C" int k5_dit_set_fp8(k5_dit* d, int enabled) {
  g_err[0] = 0;
  if (!d || !d->finalized) { k5_set_error("k5_dit_set_fp8: handle not finalized"); return K5_ERR_STATE; }
  if (enabled < 0 || enabled > 7) { k5_set_error("k5_dit_set_fp8: a bit mask 0..7 (1 feed-forward, 2 q|k|v projections, 4 out projection)"); return K5_ERR_ARG; }
  if (enabled && ((d->D % 128) || (d->FF % 128) || d->D < 256 || d->FF < 256)) {
    k5_set_error("k5_dit_set_fp8: model_dim and ff_dim must be multiples of 128 and >= 256"); return K5_ERR_UNSUPPORTED;
  }
  const size_t D = d->D, FF = d->FF;
  auto quant = [&](const DevBuf& w, DevBuf& w8, DevBuf& sc, size_t rows, size_t K) -> int {
    if (w8.p) return K5_OK;
    K5CHK(w8.ensure(rows * K)); K5CHK(sc.ensure(rows * 4));
    return k5_launch_quant_rows_fp8(w.p, w8.p, sc.as<float>(), (int)rows, (int)K, (int)K, (int)K, nullptr);
  };
  for (auto& b : d->vblocks) {
    if (enabled & 1) { K5CHK(quant(b.w1, b.w1_f8, b.s1_f8, FF, D)); K5CHK(quant(b.w2, b.w2_f8, b.s2_f8, D, FF)); }
    if (enabled & 2) { K5CHK(quant(b.self_attn.wqk, b.self_attn.wqk8, b.self_attn.sqk, 2 * D, D)); K5CHK(quant(b.self_attn.wv, b.self_attn.wv8, b.self_attn.sv, D, D)); }
    if (enabled & 4) K5CHK(quant(b.self_attn.wo, b.self_attn.wo8, b.self_attn.so, D, D));
  }
  HIPCHK(hipDeviceSynchronize());
  d->use_fp8 = (enabled & 1) != 0;
  d->fp8_mask = enabled;
  return K5_OK;
}

extern "C" int k5_dit_set_graph(k5_dit* d, int enabled) { if (!d) return K5_ERR_ARG; d->use_graph = enabled != 0; return K5_OK; }

// ---------------------------------------------------------------------------------------------
// C ABI: sequence parallelism over RCCL
// ---------------------------------------------------------------------------------------------
extern "C" int k5_comm_unique_id(const char* rccl_lib_path, void* out128) {
  g_err[0] = 0;
  if (!out128) return K5_ERR_ARG;
  static Comm probe;  // library handle only
  K5CHK(probe.open(rccl_lib_path));
  ncclUniqueId id;
  const ncclResult_t r = probe.GetUniqueId(&id);
  if (r != ncclSuccess) { k5_set_error("ncclGetUniqueId: %s", probe.GetErrorString(r)); return K5_ERR_HIP; }
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  memcpy(out128, &id, 128);
  return K5_OK;
}

static int comm_common_init(k5_dit* d, int rank, int world) {
  d->comm.rank = rank; d->comm.world = world;
  if (const char* e = getenv("K5_SP_AUTOTUNE")) d->sp_autotune = atoi(e) != 0;   // K5_SP_AUTOTUNE=1: time the admissible exchanges at the first sharded forward (opt-in)
  if (d->comm.loop) d->sp_autotune = false;   // loopback ranks (tests of specific schedules on one GPU) tune only when asked to ("sp_autotune" = 1)
  d->sp_rank = rank; d->sp_world = world;
  HIPCHK(hipStreamCreateWithFlags(&d->comm_stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&d->ev_k, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&d->ev_v, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&d->ev_gathered, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&d->ev_stats, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&d->ev_means, hipEventDisableTiming));
  for (auto& e : d->ev_slice) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return K5_OK;
}

extern "C" int k5_dit_comm_init(k5_dit* d, const char* rccl_lib_path, int rank, int world, const void* unique_id128) {
  g_err[0] = 0;
  if (!d || !unique_id128 || world < 1 || rank < 0 || rank >= world) return K5_ERR_ARG;
  if (d->comm.active()) { k5_set_error("communicator already initialised"); return K5_ERR_STATE; }
  K5CHK(d->comm.open(rccl_lib_path));
  ncclUniqueId id;
  memcpy(&id, unique_id128, 128);
  const ncclResult_t r = d->comm.CommInitRank(&d->comm.comm, world, id, rank);
  if (r != ncclSuccess) { k5_set_error("ncclCommInitRank: %s", d->comm.GetErrorString(r)); return K5_ERR_HIP; }
  return comm_common_init(d, rank, world);
}

// CFG-parallel pair (see k5_dit::pair): a 2-rank communicator of its own; branch = this handle's rank in it (0 = conditional forward,
// 1 = unconditional).  Collective over the two handles of the pair; independent of (and initialised after) the sequence-parallel one.
static int pair_common_init(k5_dit* d, int branch) {
  d->pair.rank = branch; d->pair.world = 2; d->cfg_branch = branch;
  HIPCHK(hipStreamCreateWithFlags(&d->pair_stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&d->ev_vel_ready, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&d->ev_vel_done, hipEventDisableTiming));
  return K5_OK;
}
extern "C" int k5_dit_cfg_pair_init(k5_dit* d, const char* rccl_lib_path, int branch, const void* unique_id128) {
  g_err[0] = 0;
  if (!d || !unique_id128 || branch < 0 || branch > 1) return K5_ERR_ARG;
  if (d->pair.active()) { k5_set_error("CFG pair communicator already initialised"); return K5_ERR_STATE; }
  K5CHK(d->pair.open(rccl_lib_path));
  ncclUniqueId id;
  memcpy(&id, unique_id128, 128);
  const ncclResult_t r = d->pair.CommInitRank(&d->pair.comm, 2, id, branch);
  if (r != ncclSuccess) { k5_set_error("ncclCommInitRank (CFG pair): %s", d->pair.GetErrorString(r)); return K5_ERR_HIP; }
  return pair_common_init(d, branch);
}

// ---- loopback group: P handles of one process on one GPU as the P ranks of a sequence-parallel run (tests) ----
struct k5_loopback { LoopGroup g; };
extern "C" int k5_loopback_create(int world, k5_loopback** out) {
  g_err[0] = 0;
  if (!out || world < 1 || world > 64) return K5_ERR_ARG;
  k5_loopback* lb = new k5_loopback();
  lb->g.world = world;
  if (pthread_barrier_init(&lb->g.bar, nullptr, (unsigned)world)) { delete lb; k5_set_error("pthread_barrier_init failed"); return K5_ERR_STATE; }
  lb->g.ptr.assign(world, nullptr); lb->g.joined.assign(world, 0);
  lb->g.ready.resize(world); lb->g.pulled.resize(world);
  for (int i = 0; i < world; ++i) {
    HIPCHK(hipEventCreateWithFlags(&lb->g.ready[i], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&lb->g.pulled[i], hipEventDisableTiming));
  }
  *out = lb;
  return K5_OK;
}
extern "C" void k5_loopback_destroy(k5_loopback* lb) {
  if (!lb) return;
  for (auto e : lb->g.ready) (void)hipEventDestroy(e);
  for (auto e : lb->g.pulled) (void)hipEventDestroy(e);
  pthread_barrier_destroy(&lb->g.bar);
  delete lb;
}
extern "C" int k5_dit_comm_init_loopback(k5_dit* d, k5_loopback* lb, int rank) {
  g_err[0] = 0;
  if (!d || !lb || rank < 0 || rank >= lb->g.world) return K5_ERR_ARG;
  if (d->comm.active()) { k5_set_error("communicator already initialised"); return K5_ERR_STATE; }
  if (lb->g.joined[rank]) { k5_set_error("loopback rank %d is already taken", rank); return K5_ERR_STATE; }
  lb->g.joined[rank] = 1;
  d->comm.loop = &lb->g;
  return comm_common_init(d, rank, lb->g.world);
}

extern "C" int k5_dit_cfg_pair_init_loopback(k5_dit* d, k5_loopback* lb, int branch) {
  g_err[0] = 0;
  if (!d || !lb || lb->g.world != 2 || branch < 0 || branch > 1) return K5_ERR_ARG;
  if (d->pair.active()) { k5_set_error("CFG pair communicator already initialised"); return K5_ERR_STATE; }
  if (lb->g.joined[branch]) { k5_set_error("loopback pair rank %d is already taken", branch); return K5_ERR_STATE; }
  lb->g.joined[branch] = 1;
  d->pair.loop = &lb->g;
  return pair_common_init(d, branch);
}
extern "C" int k5_dit_cfg_branch(k5_dit* d) { return d && d->pair.active() ? d->cfg_branch : -1; }

// ---- IPC group: one PROCESS per rank, peers' buffers mapped through hipIpc handles, epoch flags in device memory (ipc_comm.h) ----
static int ipc_open(Comm& c, const char* shm_name, int rank, int world) {
  if (c.active()) { k5_set_error("communicator already initialised"); return K5_ERR_STATE; }
  k5ipc::Group* g = new k5ipc::Group();
  if (g->open(shm_name, rank, world)) { k5_set_error("%s", g->err.c_str()); g->close_all(); delete g; return K5_ERR_STATE; }
  c.ipc = g;
  return K5_OK;
}
extern "C" int k5_dit_comm_init_ipc(k5_dit* d, const char* shm_name, int rank, int world) {
  g_err[0] = 0;
  if (!d || !shm_name || world < 1 || world > k5ipc::MAXR || rank < 0 || rank >= world) return K5_ERR_ARG;
  K5CHK(ipc_open(d->comm, shm_name, rank, world));
  K5CHK(comm_common_init(d, rank, world));
  d->sp_autotune = false;   // several ranks may share one device: timings of the candidates would measure the oversubscription
  return K5_OK;
}
extern "C" int k5_dit_cfg_pair_init_ipc(k5_dit* d, const char* shm_name, int branch) {
  g_err[0] = 0;
  if (!d || !shm_name || branch < 0 || branch > 1) return K5_ERR_ARG;
  K5CHK(ipc_open(d->pair, shm_name, branch, 2));
  return pair_common_init(d, branch);
}

// Engine options (all default 0).
//   "attn_mode"       0 = softmax form per head from the data (fixed offset where |q||k'| <= 90, online max elsewhere),
//                     1 = online max everywhere (what a checkpoint with large QK-norm gains gets; bench.py --attn-online)
//   "sp_pass1_tiles"  local key tiles attended before the K/V^T gather has landed (0 = all of the rank's own tiles)
//   "attn_row_offsets" 1 (default) / 0: fixed-offset softmax with per-row offsets — heads whose bound max|q| max|k'| lies in (90, 190]
//                     stay on the fast kernel (a row whose sum underflows sends its head to the online form late); 0 = the plain <= 90 rule
//   "attn_anchor"     1 (default) / 0: heads beyond that window keep the fixed form on offsets anchored at achieved scores (the row's maximum
//                     over a sample of keys + 20; attn_row_anchor_kernel) — no underflow whatever the norms, a job whose row sum
//                     overflows falls back to the online form like an underflowing one; 0 = such heads take the online form (one-GPU path)
//   "attn_pref_reset" (write-only action) forget which heads kept falling back from the per-row-offset form (k5_sample does it at the start of
//                     every call; a caller stepping k5_dit_forward itself does it at the start of a sampling run)
//   "attn_fuse_qnorm" 0 (default): norm_qk + RoPE of the visual queries is a standalone pass; 1 = dense visual self-attention on ONE rank
//                     applies it inside the attention kernel's Q-fragment load (K5QueryNorm; needs attn_row_offsets or attn_mode 1);
//                     2 = under sequence parallelism too.  Measured neutral (elementwise -2.0 ms, attention +0.6 .. +3.4 ms per step
//                     depending on how the compiler schedules the tile loop of the extra instantiation), hence opt-in.
//   "attn_fuse_qnorm_auto" 1 (default) / 0: with attn_fuse_qnorm = 0, k5_sample on a handle without a sequence-parallel communicator decides per call:
//                     step 0 as above (standalone pass) while it records whether any head left the plain fixed-offset form; if none did, steps 1.. run as
//                     attn_fuse_qnorm = 1.  BIT-IDENTICAL either way since round 6 (the two kernels' norm + rotation arithmetic is spelled out operation
//                     for operation: tools/probes/qn_arith_probe.hip); final build, alternating on one box: 522.8-523.1 -> 516.5 ms per step
//                     (elementwise -1.4, attention -4).  "attn_fuse_qnorm_used" (read-only): the last call's decision (1 / 0; -1 = not evaluated)
//   "nabla_group_rows" 0 (default) / 2 / 4: 64-query rows per NABLA key-tile list = per attention workgroup on one GPU; 0 picks 2 (128-query
//                     workgroups) when the previous forward's first map kept less than half of its blocks, else 4; same bits either way
//   "ipc_flags_finegrained" (read-only): 1 when the IPC group's flag page is fine-grained device memory (csrc/ipc_comm.h Group::open; K5_IPC_COARSE_FLAGS=1 forces plain memory)
//   "gemm_split_tail" 0 (default) / 1 / 2 (PROCESS-wide, like K5_GEMM_SK): 1 = the ragged last round of a four-wave GEMM launch is cut along K into two
//                     aligned slices per tile on two workgroups of one XCD (csrc/gemm_bf16.hip) where at most half of the CUs would be busy, 2 = wherever
//                     a tile can be cut; 0 = whole tiles everywhere (one K order in every kernel).  Measured neutral through the engine, hence opt-in
//   "cross_kv_batched" 1 (default) / 0: the cross-attention key / V^T projections of all visual blocks as two GEMMs against stacked weights before
//                     the visual stack (they depend on the text stream only), their key norms as one launch; same bits as the per-block launches
//   "nabla_fuse_means" 1 (default: where the launch has >= 150 k threads) / 2 (always) / 0: the 64-token block means NABLA's map is built from are taken by the norm + RoPE pass itself (one read of q | k
//                     less per block, and on one GPU the keys are scaled in place: no scaled copy); same bits as with 0
//   "fp8_fuse_ln"     1 (default) / 0: under k5_dit_set_fp8 the LayerNorm in front of an e4m3 projection writes the e4m3 rows itself (no bf16 h, no
//                     quantisation pass: two passes of N x D less per block); same bits as with 0
//   "nabla_pair_frames" 1 (default) / 0: which two 64-query rows share a 128-query list — the same spatial tile of adjacent frames, or adjacent
//                     tiles of one frame (round 3); same bits, tighter unions under the sliding-tile window (k5_pair_row)
//   "sp_nabla_passes" 1 (default) / 2: NABLA under sequence parallelism walks every list in one pass after the gather, or in two — the
//                     rank's own key blocks while the other ranks' keys travel, the rest after the gather (costs 12 % of the attention
//                     in compute, emulated P = 4; pays when the exposed part of the gather is longer than that — a property of the node)
//   "sp_mode"         0 (default): every rank gathers all K / V^T (any rank count, dense and NABLA); 1: Ulysses — two all-to-alls trade token rows
//                     for heads and back (run_self_attention_ulysses; needs heads % ranks == 0 and dense attention, else the gather is used)
//   "sp_autotune"     0 (default since round 5) / 1: the first sharded forward of a handle with more than one rank times one block's self-attention section under
//                     every admissible schedule (all-gather; 2 slices; Ulysses — NABLA: one / two passes) and keeps the fastest (max over ranks; every
//                     rank takes the same decision from the gathered table).  A knob set explicitly through this call is left alone, and a run only assigns
//                     the knobs it varied.  The winner depends on timings and the schedules differ in fp32 summation order: with tuning on, bits may
//                     differ between nodes / runs.  2 = tune again.
//                     k5_dit_sp_schedule returns what was measured and chosen.
//   "sp_slices"       S in 1..4: exchange K / V^T of a block in S slices (grouped send/recv to every peer at once) and attend each
//                     slice as it lands — the gather hides behind the attention of the slices before it (dense attention; NABLA and
//                     S = 1 keep the single in-place all-gather).  Token slots become multiples of 64 S.
//   "emulate_world"   TIMING ONLY, one GPU: lay the work out as rank 0 of a P-rank group while the communicator has one rank —
//                     collectives move nothing, the other ranks' keys are never filled, RESULTS ARE GARBAGE; the handle is
//                     marked (k5_dit_get_option "emulated" = 1) so that a bench can refuse to report it as a measurement
extern "C" int k5_dit_set_option(k5_dit* d, const char* name, int value) {
  g_err[0] = 0;
  if (!d || !name) return K5_ERR_ARG;
  if (!strcmp(name, "attn_mode")) {
    if (value != K5_ATTN_AUTO && value != K5_ATTN_ONLINE) { k5_set_error("attn_mode must be 0 or 1"); return K5_ERR_ARG; }
    d->attn_mode = value; return K5_OK;
  }
  if (!strcmp(name, "sp_pass1_tiles")) { if (value < 0) return K5_ERR_ARG; d->sp_pass1_tiles = value; return K5_OK; }
  if (!strcmp(name, "sp_mode")) { if (value < 0 || value > 1) return K5_ERR_ARG; d->sp_mode = value; d->sp_user_set |= 1; return K5_OK; }
  if (!strcmp(name, "sp_autotune")) { d->sp_autotune = value != 0; if (value > 1) d->sp_tuned = false; return K5_OK; }   // 2 = tune again at the next sharded forward
  if (!strcmp(name, "attn_row_offsets")) { d->row_offsets = value != 0; return K5_OK; }
  if (!strcmp(name, "attn_anchor")) { d->anchor = value != 0; return K5_OK; }
  if (!strcmp(name, "attn_pref_reset")) return reset_attn_pref(d, nullptr, true);   // an action, not a state: the per-step path (k5_dit_forward) calls it per run
  if (!strcmp(name, "cross_kv_batched")) { if (value != 0 && value != 1) return K5_ERR_ARG; d->cross_kv_batched = value; return K5_OK; }
  if (!strcmp(name, "gemm_split_tail")) { if (value < -1 || value > 2) return K5_ERR_ARG; k5_gemm_set_stream_k_default(value); return K5_OK; }   // process-wide: the GEMM launcher has no handle
  if (!strcmp(name, "nabla_fuse_means")) { if (value < 0 || value > 2) return K5_ERR_ARG; d->nabla_fuse_means = value; return K5_OK; }
  if (!strcmp(name, "fp8_fuse_ln")) { if (value != 0 && value != 1) return K5_ERR_ARG; d->fp8_fuse_ln = value; return K5_OK; }
  if (!strcmp(name, "nabla_pair_frames")) { if (value != 0 && value != 1) return K5_ERR_ARG; d->nabla_pair_frames = value; return K5_OK; }
  if (!strcmp(name, "nabla_group_rows")) { if (value != 0 && value != 1 && value != 2 && value != 4) return K5_ERR_ARG; d->nabla_group_rows = value; return K5_OK; }
  if (!strcmp(name, "sp_nabla_passes")) { if (value < 1 || value > 2) return K5_ERR_ARG; d->sp_nabla_passes = value; d->sp_user_set |= 4; return K5_OK; }
  if (!strcmp(name, "attn_fuse_qnorm")) { if (value < 0 || value > 2) return K5_ERR_ARG; d->fuse_qnorm = value; return K5_OK; }
  if (!strcmp(name, "attn_fuse_qnorm_auto")) { if (value != 0 && value != 1) return K5_ERR_ARG; d->fuse_qnorm_auto = value; return K5_OK; }
  if (!strcmp(name, "sp_slices")) {
    if (value < 1 || value > 4) return K5_ERR_ARG;
    if (value > 1 && d->comm.comm && !d->comm.can_exchange()) { k5_set_error("sp_slices > 1 needs ncclSend / ncclRecv / ncclGroup* in the RCCL library"); return K5_ERR_STATE; }
    d->sp_slices = value; d->sp_user_set |= 2; return K5_OK;
  }
  if (!strcmp(name, "emulate_world")) {
    if (!d->comm.active() || d->comm.world != 1 || value < 1) { k5_set_error("emulate_world needs a world = 1 communicator"); return K5_ERR_STATE; }
    d->sp_world = value; d->emulated = value > 1;
    if (d->emulated) fprintf(stderr, "libk5: emulate_world=%d — timing-only layout, the results of this handle are garbage\n", value);
    return K5_OK;
  }
  k5_set_error("unknown option %s", name);
  return K5_ERR_ARG;
}
extern "C" int k5_dit_get_option(k5_dit* d, const char* name, int* value) {
  if (!d || !name || !value) return K5_ERR_ARG;
  if (!strcmp(name, "attn_mode")) *value = d->attn_mode;
  else if (!strcmp(name, "sp_pass1_tiles")) *value = d->sp_pass1_tiles;
  else if (!strcmp(name, "sp_mode")) *value = d->sp_mode;
  else if (!strcmp(name, "sp_slices")) *value = d->sp_slices;
  else if (!strcmp(name, "attn_row_offsets")) *value = d->row_offsets ? 1 : 0;
  else if (!strcmp(name, "attn_anchor")) *value = d->anchor ? 1 : 0;
  else if (!strcmp(name, "attn_fuse_qnorm")) *value = d->fuse_qnorm;
  else if (!strcmp(name, "attn_fuse_qnorm_auto")) *value = d->fuse_qnorm_auto;
  else if (!strcmp(name, "attn_fuse_qnorm_used")) *value = d->fuse_last;
  else if (!strcmp(name, "sp_nabla_passes")) *value = d->sp_nabla_passes;
  else if (!strcmp(name, "sp_autotune")) *value = d->sp_autotune ? 1 : 0;
  else if (!strcmp(name, "sp_tuned")) *value = d->sp_tuned ? 1 : 0;
  else if (!strcmp(name, "fp8_effective")) {   // the classes of k5_dit_set_fp8 that RUN in e4m3 on this handle's path — the sharded schedules keep the out
    int m = d->fp8_mask;                       // projection in bf16 (bit 4), Ulysses also the q | k | V^T projections (bit 2); a bench must label its line with THIS mask
    if (d->comm.active()) { m &= ~4; if (d->sp_mode == 1 && !d->last_nabla && d->sp_world > 1 && d->Hh % d->sp_world == 0 && !d->emulated) m &= ~2; }   // the dispatch's own predicate (forward_impl: `ulysses`), with the attention type of the LAST forward
    *value = m;
  }
  else if (!strcmp(name, "rccl_ranks")) {   // the size the RCCL communicator reports (ncclCommCount): 0 = no communicator, -1 = a loopback group / no such symbol
    *value = 0;
    if (d->comm.comm && d->comm.CommCount) { int n = 0; *value = d->comm.CommCount(d->comm.comm, &n) == ncclSuccess ? n : -1; }
    else if (d->comm.active()) *value = -1;
  }
  else if (!strcmp(name, "ipc_ranks")) *value = d->comm.ipc ? d->comm.ipc->world : 0;   // processes of the IPC group this handle is a rank of (0 = not that transport)
  else if (!strcmp(name, "ipc_pair_ranks")) *value = d->pair.ipc ? d->pair.ipc->world : 0;
  else if (!strcmp(name, "ipc_collectives")) *value = (int)((d->comm.ipc ? d->comm.ipc->collectives : 0) + (d->pair.ipc ? d->pair.ipc->collectives : 0));
  else if (!strcmp(name, "ipc_pulled_mb")) *value = (int)(((d->comm.ipc ? d->comm.ipc->bytes_pulled : 0) + (d->pair.ipc ? d->pair.ipc->bytes_pulled : 0)) >> 20);
  else if (!strcmp(name, "ipc_flags_finegrained")) *value = (d->comm.ipc && d->comm.ipc->flags_fine) ? 1 : 0;   // the IPC group's flag page is fine-grained device memory
  else if (!strcmp(name, "ipc_errors")) {   // synchronises the device: the first flag wait that ran into its time limit (0 = none; bit 31 | which << 24 | peer << 16 | epoch)
    *value = 0;
    for (Comm* c : {&d->comm, &d->pair}) if (c->ipc) { uint32_t w = 0; if (c->ipc->error_word(&w)) { k5_set_error("%s", c->ipc->err.c_str()); return K5_ERR_HIP; } if (w && !*value) *value = (int)w; }
  }
  else if (!strcmp(name, "nabla_group_rows")) *value = d->nabla_group_rows;
  else if (!strcmp(name, "nabla_pair_frames")) *value = d->nabla_pair_frames;
  else if (!strcmp(name, "fp8_fuse_ln")) *value = d->fp8_fuse_ln;
  else if (!strcmp(name, "nabla_fuse_means")) *value = d->nabla_fuse_means;
  else if (!strcmp(name, "cross_kv_batched")) *value = d->cross_kv_batched;
  else if (!strcmp(name, "gemm_split_tail")) *value = k5_gemm_stream_k_policy();
  else if (!strcmp(name, "emulate_world")) *value = d->emulated ? d->sp_world : 0;
  else if (!strcmp(name, "emulated")) *value = d->emulated ? 1 : 0;
  else { k5_set_error("unknown option %s", name); return K5_ERR_ARG; }
  return K5_OK;
}

// How many (block, head) self-attention launches took the fixed-offset / the online-max softmax since the last reset
// (synchronises the device; bench.py reports it as roofline.variant).  reset != 0 clears the counters.
extern "C" int k5_dit_attn_variant_counts(k5_dit* d, long long* fixed_heads, long long* online_heads, int reset) {
  if (!d) return K5_ERR_ARG;
  unsigned long long c[4] = {0, 0, 0, 0};
  if (d->ws_attn_cnt.p) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(c, d->ws_attn_cnt.p, 32, hipMemcpyDeviceToHost));
    if (reset) { HIPCHK(hipMemset(d->ws_attn_cnt.p, 0, 32)); d->nabla_possible = 0; }
  }
  if (fixed_heads) *fixed_heads = (long long)c[0];
  if (online_heads) *online_heads = (long long)c[1];
  return K5_OK;
}
// NABLA maps computed while profiling was on: kept / possible 64x64 blocks since the last k5_dit_attn_variant_counts reset
extern "C" int k5_dit_set_nabla_tap(k5_dit* d, void* dev_u8, long long capacity_bytes) {
  if (!d || capacity_bytes < 0 || (dev_u8 && capacity_bytes == 0)) return K5_ERR_ARG;
  d->nabla_tap = dev_u8; d->nabla_tap_cap = dev_u8 ? capacity_bytes : 0; d->nabla_tap_n = 0;
  return K5_OK;
}
extern "C" int k5_dit_nabla_tap_count(k5_dit* d, long long* maps) {
  if (!d || !maps) return K5_ERR_ARG;
  *maps = d->nabla_tap_n;
  return K5_OK;
}
extern "C" int k5_dit_nabla_block_counts(k5_dit* d, long long* kept, long long* possible) {
  if (!d) return K5_ERR_ARG;
  unsigned long long c[4] = {0, 0, 0, 0};
  if (d->ws_attn_cnt.p) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(c, d->ws_attn_cnt.p, 32, hipMemcpyDeviceToHost));
  }
  if (kept) *kept = (long long)c[2];
  if (possible) *possible = d->nabla_possible;
  return K5_OK;
}
// ... and the blocks the list-driven attention EXECUTED for them: sum of the union lists' lengths x rows per list (one GPU path) — the
// launch's union efficiency is kept / executed
extern "C" int k5_dit_nabla_executed_blocks(k5_dit* d, long long* executed) {
  if (!d || !executed) return K5_ERR_ARG;
  unsigned long long c[4] = {0, 0, 0, 0};
  if (d->ws_attn_cnt.p) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(c, d->ws_attn_cnt.p, 32, hipMemcpyDeviceToHost));
  }
  if (d->comm.active()) { k5_set_error("k5_dit_nabla_executed_blocks: the sharded path does not count its lists"); return K5_ERR_UNSUPPORTED; }
  *executed = (long long)c[3];      // summed on the device per launch as list length x rows of the group (the group size may change between forwards)
  return K5_OK;
}

// What the self-tuning sequence-parallel schedule measured and chose (JSON text; "{}" before the first sharded forward of a multi-rank handle).
// Returns the length of the text; copies at most len - 1 characters + NUL into buf (buf may be null to query the length).
extern "C" int k5_dit_sp_schedule(k5_dit* d, char* buf, int len) {
  if (!d) return K5_ERR_ARG;
  const std::string& r = d->sp_report.empty() ? std::string("{}") : d->sp_report;
  if (buf && len > 0) { const int n = (int)r.size() < len - 1 ? (int)r.size() : len - 1; memcpy(buf, r.data(), (size_t)n); buf[n] = 0; }
  return (int)(d->sp_report.empty() ? 2 : d->sp_report.size());
}
// The selection rule of that tuner as a plain function (tests): times[rank * ncand + cand] in ms (<= 0 or non-finite = did not run), valid
// (nullable) = admissible candidates; returns the chosen candidate or -1, and max-over-ranks per candidate in cost_out (nullable, -1 = out).
extern "C" int k5_sp_pick_schedule(const float* times, int ncand, int world, const int* valid, float* cost_out) {
  if (!times || ncand <= 0 || world <= 0) return -1;
  return sp_pick(times, ncand, world, valid, cost_out);
}

extern "C" int k5_dit_set_profiling(k5_dit* d, int level) { if (!d || level < 0 || level > 2) return K5_ERR_ARG; d->profiling = level; return K5_OK; }
extern "C" int k5_dit_reset_profile(k5_dit* d) { if (!d) return K5_ERR_ARG; drain_profile(d); d->prof.clear(); d->prof_self_blocks = 0; return K5_OK; }
extern "C" int k5_dit_get_profile(k5_dit* d, const char* family, double* total_ms, int64_t* launches) {
  if (!d || !family) return K5_ERR_ARG;
  drain_profile(d);
  if (!strcmp(family, "self_blocks")) {   // not a kernel family: how many visual blocks ran their self-attention under profiling
    if (total_ms) *total_ms = 0.0;
    if (launches) *launches = d->prof_self_blocks;
    return K5_OK;
  }
  auto it = d->prof.find(family);
  if (total_ms) *total_ms = it == d->prof.end() ? 0.0 : it->second.ms;
  if (launches) *launches = it == d->prof.end() ? 0 : it->second.n;
  return K5_OK;
}
