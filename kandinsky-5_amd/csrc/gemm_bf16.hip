// gemm_bf16.hip — C[M][N] = A[M][K] · W[N][K]^T (+bias) with fused epilogues, bf16 in, fp32
// accumulate on MFMA 32x32x16, bf16 out.  Both operands are K-contiguous ("NT").
//
// Replaces the autocast nn.Linear call sites of the reference DiT (kandinsky/models/nn.py:
// get_qkv :180-191/:233-244/:317-326, out_l :204-206/:282-284/:339-341, FeedForward :352-361,
// TextEmbeddings.in_layer :71, VisualEmbeddings.in_layer :96, OutLayer.out_layer :382) and fuses
// apply_gate_sum (nn.py:30-33) and nn.GELU (nn.py:356) into the epilogue.
//
// Structure (gfx950): 128x128x64 block tile, 256 threads = 4 waves (2 along M x 2 along N), each
// wave a 64x64 sub-tile as 2x2 MFMA 32x32 tiles.  The MFMA "A" operand is the W tile (rows = n)
// and the "B" operand is the A tile (rows = m), so an accumulator lane owns ONE token row m and
// 4-element runs of consecutive n: row reductions over n (RMSNorm) and adjacent-pair ops (RoPE)
// stay in-lane, and the C store is 8 B per lane.  LDS tiles are [128][64] bf16 with the 16-B
// chunk swizzle of k5_common.h (conflict-free ds_read_b128), double buffered, register staged:
// the global loads of tile k+1 are issued before the MFMAs of tile k and written to LDS after.
#include <stdlib.h>

#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "k5_common.h"
#include "k5_kernels.h"

#ifndef K5_GEMM_SK_DEFAULT
#define K5_GEMM_SK_DEFAULT 0   // split-K tail policy when neither K5_GEMM_SK nor a setter says otherwise (k5_gemm_set_stream_k_default): OFF.  Measured
                               // honestly — launches back to back for a second, both orders — the split tail is worth 0 +- 0.6 % on every projection of
                               // the model and nothing through the engine (profiles/r06_split_tail_sustained.log, r06_split_tail_engine_ab.log): a tail
                               // round costs a launch's latency chain + an epilogue (~40 us) whether its K loop is 28 K-tiles on 128 x 128 quadrants or 14
                               // on half tiles + a 256-KB hand-over.  It is correct, tested (kernel id 24) and stays as an option; off keeps one K order
                               // in every kernel (the bits of rounds 1-5)
#endif

namespace {
inline int k5_num_cu() {
  int dev = 0; hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
  return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
}

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 128;  // [128 rows][64 bf16]

struct GemmP {
  const bf16_t* A; const bf16_t* W; bf16_t* C;
  const float* bias;      // per n (or per m when BIAS_M), fp32 holding bf16-rounded values; may be null
  const bf16_t* resid;    // EPI_GATE: residual stream [M][ldr]
  const float* gate;      // EPI_GATE: per-n gate (fp32)
  int M, N, K, lda, ldw, ldc, ldr;
  int tiles_m, tiles_n;
  float alpha;            // EPI_F32: C_f32 = alpha * acc
  int causal_hw;          // EPI_F32, 4-wave kernel: > 0 = frame-causal scores, row i only needs columns < (i / hw + 1) * hw; output
                          // tiles wholly beyond that are not computed (nor written).  0 = all tiles
  int dbg;                // benchmarking experiments only (K5_GEMM_DBG); 0 in production
  // 256x256 kernel: logical tiles [0, lid_limit) only.  128x128 kernel in tail mode (tail_base >= 0): workgroup b computes
  // quadrant b & 3 of the 256x256 logical tile tail_base + b / 4 (tiles256_m/n = that grid's extent).
  int lid_limit, tail_base, tiles256_m, tiles256_n;
  unsigned long long* trace;  // -DK8_TRACE builds only
  // four-wave kernel, stream-K schedule (round 6): fp32 partial tiles [workgroup][wave][accumulator quad][lane] f32x4 + one flag per (workgroup, wave)
  // behind them; null = whole tiles per workgroup
  float* sk_ws;
  int sk_smax;            // most K slices a tail tile is cut into
};

// four consecutive output columns n .. n+3 of token row m (v = fp32 accumulators)
template <int EPI>
K5_DEV void gemm_epilogue_quad(const GemmP& p, float (&v)[4], int m, int n, float bias_m) {
  if (n >= p.N) return;
  const bool full = (n + 3 < p.N);
  if (EPI == K5_EPI_F32) {  // raw fp32 scores (VAE mid-block attention): C is float*
    float* fp = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
#pragma unroll
    for (int e = 0; e < 4; ++e) if (n + e < p.N) fp[e] = v[e] * p.alpha;
    return;
  }
  if (EPI == K5_EPI_BIAS_M) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += bias_m;
  } else if (p.bias) {
    if (full) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += b[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (n + e < p.N) v[e] += p.bias[n + e];
    }
  }
  if (EPI == K5_EPI_GELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(bf_round(v[e]));
  }
  if (EPI == K5_EPI_GATE) {
    const bf16_t* rp = p.resid + (size_t)m * p.ldr + n;
    if (full && ((p.ldr & 3) == 0)) {  // one 8-B residual load + one 16-B gate load per 4 outputs
      const u32x2 rr = *reinterpret_cast<const u32x2*>(rp);
      const f32x4 gg = *reinterpret_cast<const f32x4*>(p.gate + n);
      v[0] = __uint_as_float(rr[0] << 16) + gg[0] * bf_round(v[0]);
      v[1] = __uint_as_float(rr[0] & 0xffff0000u) + gg[1] * bf_round(v[1]);
      v[2] = __uint_as_float(rr[1] << 16) + gg[2] * bf_round(v[2]);
      v[3] = __uint_as_float(rr[1] & 0xffff0000u) + gg[3] * bf_round(v[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < p.N) v[e] = bf2f(rp[e]) + p.gate[n + e] * bf_round(v[e]);
    }
  }
  bf16_t* cp = p.C + (size_t)m * p.ldc + n;
  if (full && ((p.ldc & 3) == 0)) {
    u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    *reinterpret_cast<u32x2*>(cp) = o;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) if (n + e < p.N) cp[e] = f2bf(v[e]);
  }
}

// one 32x32 accumulator tile: this lane owns token row m and, per register group rg, the 4 consecutive
// output columns n_base + 8 rg + 4 hi .. +3
template <int EPI>
K5_DEV void gemm_epilogue_tile(const GemmP& p, const f32x16& acc, int m, int n_base, int hi) {
  if (m >= p.M) return;
  float bias_m = 0.f;
  if (EPI == K5_EPI_BIAS_M && p.bias) bias_m = p.bias[m];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = acc[4 * rg + e];
    gemm_epilogue_quad<EPI>(p, v, m, n_base + 8 * rg + 4 * hi, bias_m);
  }
}

template <int EPI>
K5_DEV void gemm_epilogue(const GemmP& p, f32x16 (&acc)[2][2], int m0, int n0, int wm, int wn, int hi, int l31) {
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      gemm_epilogue_tile<EPI>(p, acc[i][j], m0 + wm * 64 + j * 32 + l31, n0 + wn * 64 + i * 32, hi);
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmP p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
  char* sA = smem;                   // 2 buffers
  char* sW = smem + 2 * TILE_BYTES;  // 2 buffers

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, l31 = lane & 31;

  // block -> tile: XCD-contiguous ranges, then 8-row groups so that the ~64 tiles resident on
  // one XCD form an 8x8 patch sharing 8 A panels and 8 W panels in that XCD's L2.
  const int nblk = p.tiles_m * p.tiles_n;
  int lid = xcd_remap(blockIdx.x, nblk);
  constexpr int GM = 8;
  const int per_group = GM * p.tiles_n;
  const int g = lid / per_group, first_m = g * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  const int tm = first_m + (lid % per_group) % gsz;
  const int tn = (lid % per_group) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;
  if (EPI == K5_EPI_F32 && p.causal_hw > 0 && n0 >= ((min(m0 + BM, p.M) - 1) / p.causal_hw + 1) * p.causal_hw) return;

  // loader mapping: 1024 16-B chunks per operand tile, 4 per thread
  int ld_row[4], ld_c[4];
  const bf16_t* pa[4]; const bf16_t* pw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = tid + 256 * i;
    ld_row[i] = q >> 3; ld_c[i] = q & 7;
    pa[i] = p.A + (size_t)min(m0 + ld_row[i], p.M - 1) * p.lda + 8 * ld_c[i];
    pw[i] = p.W + (size_t)min(n0 + ld_row[i], p.N - 1) * p.ldw + 8 * ld_c[i];
  }
  u32x4 ra[4], rw[4];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = (k0 + 8 * ld_c[i]) < p.K;
      const u32x4 z = {0, 0, 0, 0};
      ra[i] = ok ? *reinterpret_cast<const u32x4*>(pa[i] + k0) : z;
      rw[i] = ok ? *reinterpret_cast<const u32x4*>(pw[i] + k0) : z;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = lds_swz(ld_row[i], ld_c[i]);
      *reinterpret_cast<u32x4*>(sA + buf * TILE_BYTES + off) = ra[i];
      *reinterpret_cast<u32x4*>(sW + buf * TILE_BYTES + off) = rw[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (p.K + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
    const char* cA = sA + buf * TILE_BYTES;
    const char* cW = sW + buf * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = 2 * ks + hi;
      bf16x8 fw[2], fx[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        fw[i] = *reinterpret_cast<const bf16x8*>(cW + lds_swz(wn * 64 + i * 32 + l31, c));
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fx[j] = *reinterpret_cast<const bf16x8*>(cA + lds_swz(wm * 64 + j * 32 + l31, c));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fw[i], fx[j], acc[i][j]);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  gemm_epilogue<EPI>(p, acc, m0, n0, wm, wn, hi, l31);
}

// ---------------------------------------------------------------------------------------------
// Direct-to-LDS variant (K % 64 == 0): tiles are staged with global_load_lds_dwordx4 — no staging VGPRs and no
// ds_write pass.  The LDS destination of that instruction is wave-uniform base + lane*16, so the image is
// lane-linear; the XOR swizzle is therefore applied to the per-lane GLOBAL source address (lane l of a 1-KiB
// piece fills row l>>3, slot l&7, and fetches the chunk that belongs there: c = slot ^ ((row>>1)&7)); the
// fragment reads use the same swizzle as the register-staged kernel.  One barrier per K-tile: its implied
// vmcnt(0) retires tile kt, then tile kt+1's loads are issued and fly under the whole MFMA phase.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_glds_kernel(GemmP p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
  char* sA = smem;
  char* sW = smem + 2 * TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, l31 = lane & 31;

  const int nblk = p.tiles_m * p.tiles_n;
  int lid = xcd_remap(blockIdx.x, nblk);
  constexpr int GM = 8;
  const int per_group = GM * p.tiles_n;
  const int g = lid / per_group, first_m = g * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  const int tm = first_m + (lid % per_group) % gsz;
  const int tn = (lid % per_group) / gsz;
  int m0 = tm * BM, n0 = tn * BN;
  if (p.tail_base >= 0) {   // quadrant of a 256x256 logical tile left over by the persistent kernel's whole rounds
    const int plid = p.tail_base + (int)(blockIdx.x >> 2), q = blockIdx.x & 3;
    const int pg = plid / (4 * p.tiles256_n), pfirst = pg * 4, pgsz = min(p.tiles256_m - pfirst, 4);
    m0 = (pfirst + (plid % (4 * p.tiles256_n)) % pgsz) * 256 + 128 * (q >> 1);
    n0 = ((plid % (4 * p.tiles256_n)) / pgsz) * 256 + 128 * (q & 1);
    if (m0 >= p.M || n0 >= p.N) return;
  }
  if (EPI == K5_EPI_F32 && p.causal_hw > 0 && n0 >= ((min(m0 + BM, p.M) - 1) / p.causal_hw + 1) * p.causal_hw) return;   // beyond the frame-causal limit

  // this wave stages pieces wave*4 .. wave*4+3 (8 rows x 128 B each) of both operand tiles
  const bf16_t* ga[4]; const bf16_t* gw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * (wave * 4 + i) + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    ga[i] = p.A + (size_t)min(m0 + row, p.M - 1) * p.lda + 8 * c;
    gw[i] = p.W + (size_t)min(n0 + row, p.N - 1) * p.ldw + 8 * c;
  }
  auto stage = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = (wave * 4 + i) * 1024;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(ga[i] + k0), (lds_void_t*)(sA + buf * TILE_BYTES + piece), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(gw[i] + k0), (lds_void_t*)(sW + buf * TILE_BYTES + piece), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    __syncthreads();  // vmcnt(0) + barrier: tile kt is in LDS for every wave; buf^1 is no longer being read
    if (kt + 1 < nk) stage(buf ^ 1, (kt + 1) * BK);
    const char* cA = sA + buf * TILE_BYTES;
    const char* cW = sW + buf * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = 2 * ks + hi;
      bf16x8 fw[2], fx[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        fw[i] = *reinterpret_cast<const bf16x8*>(cW + lds_swz(wn * 64 + i * 32 + l31, c));
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fx[j] = *reinterpret_cast<const bf16x8*>(cA + lds_swz(wm * 64 + j * 32 + l31, c));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fw[i], fx[j], acc[i][j]);
    }
  }
  gemm_epilogue<EPI>(p, acc, m0, n0, wm, wn, hi, l31);
}

// ---------------------------------------------------------------------------------------------
// 256x128x64 tile, 8 waves (4 along M x 2 along N, 64x64 each), THREE-stage LDS ring (3 x 48 KB) filled by
// global_load_lds with a COUNTED vmcnt: when tile kt is consumed, tile kt+1 is still in flight and tile kt+2 is issued
// right after the barrier, i.e. two K-tiles (~2 x 1024 MFMA cycles per SIMD) of prefetch distance.  Motivation
// (profiles/r01_gemm_traffic.md): with 128x128 tiles and a single tile of prefetch the FF GEMMs miss L2 for 7-18x their
// algorithmic bytes (3.5-6.6 GB per launch) and every barrier's vmcnt(0) sat on that latency.  hipcc's __syncthreads()
// would drain vmcnt(0) (an LDS-DMA is a pending LDS write), so the loop uses a raw s_barrier + explicit s_waitcnt.
// ---------------------------------------------------------------------------------------------
constexpr int K3_BM = 256, K3_STAGE = (256 + 128) * 128, K3_STAGES = 3;

template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_k3_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, l31 = lane & 31;

  const int nblk = p.tiles_m * p.tiles_n;
  int lid = xcd_remap(blockIdx.x, nblk);
  constexpr int GM = 4;  // 32 resident tiles per XCD = 4 x 8 patch of 256x128 tiles (1024 x 1024 outputs)
  const int per_group = GM * p.tiles_n;
  const int g = lid / per_group, first_m = g * GM;
  const int gsz = min(p.tiles_m - first_m, GM);
  const int tm = first_m + (lid % per_group) % gsz;
  const int tn = (lid % per_group) / gsz;
  const int m0 = tm * K3_BM, n0 = tn * BN;

  // 48 pieces (8 rows x 128 B) per stage: A pieces 0..31, W pieces 32..47; wave w stages pieces w, w+8, ..., w+40
  const bf16_t* gsrc[6]; int ldsoff[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int piece = wave + 8 * i;
    const bool isw = piece >= 32;
    const int prow = 8 * (isw ? piece - 32 : piece) + (lane >> 3);
    const int c = (lane & 7) ^ ((prow >> 1) & 7);
    gsrc[i] = isw ? p.W + (size_t)min(n0 + prow, p.N - 1) * p.ldw + 8 * c
                  : p.A + (size_t)min(m0 + prow, p.M - 1) * p.lda + 8 * c;
    ldsoff[i] = piece * 1024;   // A region [0, 32 KB), W region [32 KB, 48 KB) of the stage
  }
  auto stage = [&](int st, int k0) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(gsrc[i] + k0), (lds_void_t*)(dsm + st * K3_STAGE + ldsoff[i]), 16, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  stage(0, 0);
  if (nk > 1) stage(1, BK);
  int st = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt must have landed; tile kt+1 (6 DMA per wave) may stay in flight
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) stage(st == 0 ? 2 : st - 1, (kt + 2) * BK);   // ring slot (kt+2)%3 == (kt-1)%3: free since the barrier
    const char* cA = dsm + st * K3_STAGE;
    const char* cW = cA + 256 * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = 2 * ks + hi;
      bf16x8 fw[2], fx[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fw[i] = *reinterpret_cast<const bf16x8*>(cW + lds_swz(wn * 64 + i * 32 + l31, c));
#pragma unroll
      for (int j = 0; j < 2; ++j) fx[j] = *reinterpret_cast<const bf16x8*>(cA + lds_swz(wm * 64 + j * 32 + l31, c));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fw[i], fx[j], acc[i][j]);
    }
    st = st == 2 ? 0 : st + 1;
  }
  gemm_epilogue<EPI>(p, acc, m0, n0, wm, wn, hi, l31);
}

template <int EPI>
int launch_k3(GemmP p, hipStream_t stream) {
  static const hipError_t attr_rc = hipFuncSetAttribute((const void*)gemm_bf16_k3_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, K3_STAGES * K3_STAGE);   // set once, thread-safe (function-local static: loopback ranks launch from P host threads)
  if (attr_rc != hipSuccess) return K5_ERR_HIP;
  p.tiles_m = (p.M + K3_BM - 1) / K3_BM;
  hipLaunchKernelGGL(gemm_bf16_k3_kernel<EPI>, dim3(p.tiles_m * p.tiles_n), dim3(512), K3_STAGES * K3_STAGE, stream, p);
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------
// 256x256x64 tile, 8 waves in two groups of four (waves w and w+4 share a SIMD), ping-pong by one barrier: while one
// group issues the 8 MFMAs of a phase the other issues its LDS fragment reads and its share of the next tiles'
// global_load_lds — so each SIMD's matrix pipe always has a wave feeding it.  Per wave: 128 (n) x 64 (m) outputs =
// 8 x 4 MFMA 16x16x32 tiles (this shape sustains 1962 TFLOP/s on random operands against 1695 for 32x32x16 —
// tools/probes/mfma_peak.hip).  A K-tile (64) is four phases = (k-half, n-half): 16 MFMAs on 16 independent
// accumulators and 8 / 4 / 8 / 4 fragment reads (the m-side fragments of a k-half are kept for its second phase).
// Operand tiles are staged as k-halves: unit = 256 rows x 32 k = 16 KB, LDS row stride 64 B, 16-B chunk swizzle
// c' = c ^ ((row >> 2) & 3) (16 consecutive rows x one chunk = all 64 banks once).  One unit (2 DMA instructions per
// thread) is issued per phase, at least one K-tile before its first use:
//     ph1: W-klo(t+1)    ph2: X-khi(t+1)    ph3: W-khi(t+1)    ph4: X-klo(t+2)
// and every phase ends with s_waitcnt vmcnt(4) = "all but the two newest units have landed" followed by a barrier.
// A unit first read in phase R was issued in phase <= R-4, so it is retired by the wait that closes phase R-2 and
// published by barriers every wave has passed before phase R starts — including the other group, which runs one
// barrier behind.  Restaging (WAR): a slot's last fragment read is >= 2 phases before its next DMA is issued.
// LDS: 2 stages x 4 units x 16 KB = 128 KB -> one workgroup per CU, 2 waves per SIMD.
// ---------------------------------------------------------------------------------------------
constexpr int K8_BM = 256, K8_BN = 256;
constexpr int K8_UNIT = 16384, K8_XOFF = 65536, K8_LDS = 131072;   // W units at (2 stage + khalf) * 16 KB, X units 64 KB above
#ifdef K8_TRACE
constexpr int K8_TRACE_BYTES = 8 * 128 * 8;
#else
constexpr int K8_TRACE_BYTES = 0;
#endif

// MT = 16-row m-tiles per wave: 4 -> 256 x 256 block tile, 3 -> 192 (m) x 256 (n): same staging (the X units still carry 256
// rows, the last 64 unused), 3/4 of the MFMAs — for token-shard shapes whose 256-row tiles would fill only 2/3 of the CUs.
template <int EPI, int MT>
__global__ __launch_bounds__(512) void gemm_bf16_k8_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                       // ping-pong group
  const int wn = wave & 1, wm = ((wave >> 1) & 1) | (grp << 1);   // 2 waves along n, 4 along m
  const int hi = lane >> 5, l31 = lane & 31;

  // Persistent workgroups (one per CU): XCD x = blockIdx % 8 owns a contiguous range of logical tiles, its (up to 32)
  // workgroups walk it interleaved, so the tiles in flight on one XCD are neighbours (shared operand panels in its L2).
  const int nblk = p.lid_limit;   // logical tiles of this launch (the rest, if any, goes to the 128x128 kernel)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = (gridDim.x + 7 - xcd) >> 3;   // workgroups on this XCD
  const int q8 = nblk >> 3, r8 = nblk & 7;
  const int x_first = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, x_cnt = q8 + (xcd < r8 ? 1 : 0);
  constexpr int GM = 4;  // tile order inside a range: 4 m-tiles x all n-tiles groups
  const int per_group = GM * p.tiles_n;
  auto tile_origin = [&](int lid, int& m0, int& n0) {
    const int g = lid / per_group, first_m = g * GM;
    const int gsz = min(p.tiles_m - first_m, GM);
    m0 = (first_m + (lid % per_group) % gsz) * (64 * MT);
    n0 = ((lid % per_group) / gsz) * K8_BN;
  };

  // DMA sources: a unit is 16 pieces of 16 rows x 64 B; this wave stages pieces `wave` and `wave + 8`: rows
  // 16 wave + (lane>>2) and 128 + 16 wave + (lane>>2); LDS chunk slot lane&3 is filled from source chunk slot ^ swizzle(row).
  // Per-lane 32-bit byte offsets from the (uniform) operand bases.
  const int prow = 16 * wave + (lane >> 2);
  const int pc = (lane & 3) ^ ((prow >> 2) & 3);   // (row>>2)&3 is the same for prow + 128 q
  uint32_t ow[2], ox[2];
  auto set_offsets = [&](int m0, int n0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = 128 * q + prow;
      ow[q] = (uint32_t)min(n0 + r, p.N - 1) * (uint32_t)p.ldw * 2u + 16u * pc;
      ox[q] = (uint32_t)min(m0 + r, p.M - 1) * (uint32_t)p.lda * 2u + 16u * pc;
    }
  };
  const char* Wbase = reinterpret_cast<const char*>(p.W);
  const char* Xbase = reinterpret_cast<const char*>(p.A);
  const int kmax = p.K - BK;
  // unit = (stage, k-half kh) of the K-tile starting at k0; tail: k0 is clamped -> redundant (harmless) loads keep the
  // per-phase DMA count, hence the vmcnt arithmetic, uniform.  (The SGPR-base + VGPR-offset form of the DMA instruction,
  // via inline asm, measured SLOWER than letting the compiler form 64-bit VGPR addresses: 1062 vs 1164 TFLOP/s at 4096^3.)
  auto dma_w = [&](int stage, int kh, int k0) {
    const char* b = Wbase + 2 * (size_t)(min(k0, kmax) + 32 * kh);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(b + ow[q]), (lds_void_t*)(dsm + (2 * stage + kh) * K8_UNIT + (8 * q + wave) * 1024), 16, 0, 0);
  };
  auto dma_x = [&](int stage, int kh, int k0) {
    const char* b = Xbase + 2 * (size_t)(min(k0, kmax) + 32 * kh);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(b + ox[q]), (lds_void_t*)(dsm + K8_XOFF + (2 * stage + kh) * K8_UNIT + (8 * q + wave) * 1024), 16, 0, 0);
  };
  // prologue of a tile: all of K-tile 0 (klo units first) and X-klo of K-tile 1
  auto prologue = [&]() { dma_w(0, 0, 0); dma_x(0, 0, 0); dma_x(0, 1, 0); dma_w(0, 1, 0); dma_x(1, 0, BK); };

  // fragment read addresses (MFMA 16x16x32 operand: lane holds row lane&15, k-chunk lane>>4 of a 32-wide k-slab = one
  // 64-B unit row): one lane base per operand, unit / tile offsets are immediates
  const int l15 = lane & 15, lc = lane >> 4;
  const int fco = (lc ^ ((l15 >> 2) & 3)) << 4;
  const char* wb = dsm + (128 * wn + l15) * 64 + fco;
  const char* xb = dsm + K8_XOFF + (16 * MT * wm + l15) * 64 + fco;
  const int nk = p.K / BK;

#ifdef K8_TRACE
  unsigned long long* tr = reinterpret_cast<unsigned long long*>(dsm + K8_LDS) + wave * 128;
  int tri = 0;
#define K8_STAMP() do { if (lane == 0 && tri < 128) tr[tri] = __builtin_amdgcn_s_memtime(); ++tri; } while (0)
#else
#define K8_STAMP() do {} while (0)
#endif

  int m0, n0;
  if (slot < x_cnt) { tile_origin(x_first + slot, m0, n0); set_offsets(m0, n0); prologue(); }
  for (int ti = slot; ti < x_cnt; ti += per_xcd) {
    f32x4 acc[8][MT];   // [n-tile of 16][m-tile of 16]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 wf[4], xf[MT];

    // this tile's prologue DMAs were issued before the previous tile's epilogue stores (vmcnt also counts those stores,
    // which may retire out of order with loads -> a full drain here; the DMAs have had the whole epilogue to land)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind group 0 from here on

    // one phase = (k-half KH, n-half NH) of the K-tile in stage ST: fragment reads, this phase's DMA unit, barrier,
    // 16 MFMAs, wait, barrier
    auto phase = [&](auto STC, auto KHC, auto NHC, auto&& issue_dma) {
      constexpr int st = decltype(STC)::value, kh = decltype(KHC)::value, nh = decltype(NHC)::value;
      constexpr int uo = (2 * st + kh) * K8_UNIT;
      __builtin_amdgcn_sched_barrier(0);
      K8_STAMP();
      if (nh == 0) {
#pragma unroll
        for (int j = 0; j < MT; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(xb + uo + j * 16 * 64);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(wb + uo + (4 * nh + i) * 16 * 64);
      issue_dma();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[4 * nh + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[4 * nh + i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    auto ktile = [&](auto STC, int t) {
      constexpr int st = decltype(STC)::value;
      const int k1 = (t + 1) * BK, k2 = (t + 2) * BK;
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
      phase(STC, I0{}, I0{}, [&] { dma_w(st ^ 1, 0, k1); });
      phase(STC, I0{}, I1{}, [&] { dma_x(st ^ 1, 1, k1); });
      phase(STC, I1{}, I0{}, [&] { dma_w(st ^ 1, 1, k1); });
      phase(STC, I1{}, I1{}, [&] { dma_x(st, 0, k2); });
    };
    int t = 0;
    for (; t + 1 < nk; t += 2) {
      ktile(std::integral_constant<int, 0>{}, t);
      ktile(std::integral_constant<int, 1>{}, t + 1);
    }
    if (nk & 1) ktile(std::integral_constant<int, 0>{}, t);
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance group 1's extra barrier: every fragment read of this tile is done
    // the tail's redundant DMAs must have landed before the same slots are restaged for the next tile (and before exit)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int cm0 = m0, cn0 = n0;
    if (ti + per_xcd < x_cnt) {   // next tile's prologue flies under this tile's epilogue
      tile_origin(x_first + ti + per_xcd, m0, n0);
      set_offsets(m0, n0);
      prologue();
    }
    // epilogue indices are re-derived from a laundered thread id so that none of the address math is hoisted above the
    // K loop (where it would push the accumulators into scratch)
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int e_l15 = tid2 & 15, e_lc = (tid2 >> 4) & 3, e_wave = tid2 >> 6;
    const int e_wn = e_wave & 1, e_wm = ((e_wave >> 1) & 1) | ((e_wave >> 2) << 1);
    // accumulator tile (i, j): this lane holds token row m = 16 j + lane&15 and output columns n = 16 i + 4 (lane>>4) + 0..3
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int m = cm0 + 16 * MT * e_wm + 16 * j + e_l15;
      if (m >= p.M) continue;
      float bias_m = 0.f;
      if (EPI == K5_EPI_BIAS_M && p.bias) bias_m = p.bias[m];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        gemm_epilogue_quad<EPI>(p, v, m, cn0 + 128 * e_wn + 16 * i + 4 * e_lc, bias_m);
      }
    }
  }
#undef K8_STAMP
#ifdef K8_TRACE
  __syncthreads();
  if (blockIdx.x == 0 && p.trace) for (int i = tid; i < 8 * 128; i += 512) p.trace[i] = reinterpret_cast<unsigned long long*>(dsm + K8_LDS)[i];
#endif
}

template <int EPI, int MT>
int launch_k8_mt(GemmP p, hipStream_t stream, int num_cu, bool full_grid, bool no_tail) {
  static const hipError_t attr_rc = hipFuncSetAttribute((const void*)gemm_bf16_k8_kernel<EPI, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, K8_LDS + K8_TRACE_BYTES);   // set once, thread-safe (function-local static: loopback ranks launch from P host threads)
  if (attr_rc != hipSuccess) return K5_ERR_HIP;
  p.tiles_m = (p.M + 64 * MT - 1) / (64 * MT); p.tiles_n = (p.N + K8_BN - 1) / K8_BN;
  const int tiles = p.tiles_m * p.tiles_n;
  // Whole rounds of num_cu tiles go to the persistent kernel; a last round that would fill less than half of the CUs is
  // computed as 128x128 quadrants by the small kernel instead (1302 tiles of an N = 1792 projection: 5 rounds + 88 small
  // workgroups instead of 6 rounds).  (256-row tiles only.)
  const int full = tiles / num_cu * num_cu, rem = tiles - full;
  const bool split_tail = MT == 4 && !no_tail && !full_grid && full > 0 && rem > 0 && 2 * rem < num_cu;
  p.lid_limit = split_tail ? full : tiles;
  p.tiles256_m = p.tiles_m; p.tiles256_n = p.tiles_n;
  const int grid = full_grid ? tiles : min(p.lid_limit, num_cu);   // persistent: one workgroup per CU (128 KB of LDS each)
#ifdef K8_TRACE
  p.trace = getenv("K5_GEMM_TRACE") ? (unsigned long long*)strtoull(getenv("K5_GEMM_TRACE"), nullptr, 16) : nullptr;
#endif
  hipLaunchKernelGGL((gemm_bf16_k8_kernel<EPI, MT>), dim3(grid), dim3(512), K8_LDS + K8_TRACE_BYTES, stream, p);
  if (split_tail) {
    p.tail_base = full;
    p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL(gemm_bf16_glds_kernel<EPI>, dim3(4 * rem), dim3(256), 0, stream, p);
  }
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}

template <int EPI>
int launch_k8(GemmP p, hipStream_t stream) {
  static const int num_cu = k5_num_cu();   // one device per process (one process per GPU); initialised once, thread-safe
  if (num_cu <= 0) return K5_ERR_HIP;
  static const bool full_grid = getenv("K5_GEMM_FULLGRID") != nullptr;   // A/B: one workgroup per tile instead of per CU
  static const bool no_tail = getenv("K5_GEMM_NO_TAIL") != nullptr;      // A/B switches for benchmarking
  static const int force_mt = getenv("K5_GEMM_MT") ? atoi(getenv("K5_GEMM_MT")) : 0;
  // 192-row tiles when they need fewer (cost-weighted) rounds of the CUs than 256-row tiles: a round of 192-row tiles costs
  // 3/4; a last round that the tail split handles costs ~0.3.  An 8-GPU shard's N = 1792 projections: 217 tiles x 0.75 vs 168 x 1.
  auto cost = [&](int bm, double w, bool tail_ok) {
    const int tiles = ((p.M + bm - 1) / bm) * ((p.N + K8_BN - 1) / K8_BN);
    const int full = tiles / num_cu, rem = tiles % num_cu;
    return w * (full + (rem == 0 ? 0.0 : (tail_ok && full > 0 && 2 * rem < num_cu ? 0.3 : 1.0)));
  };
  const bool mt3 = force_mt ? force_mt == 3 : cost(192, 0.75, false) < 0.97 * cost(256, 1.0, !no_tail);
  return mt3 ? launch_k8_mt<EPI, 3>(p, stream, num_cu, full_grid, no_tail) : launch_k8_mt<EPI, 4>(p, stream, num_cu, full_grid, no_tail);
}

// ---------------------------------------------------------------------------------------------
// 256x256x64 tile, FOUR waves, one per SIMD, each a 128 x 128 quadrant = 8 x 8 MFMA 16x16x32 tiles on 64 independent
// accumulators (256 registers; with one wave per SIMD the wave owns all 512).  Against the 8-wave kernel above: 2/3 of
// the LDS fragment traffic (32 KB instead of 48 KB per 128 MFMAs), 2 barriers per K-tile instead of 8, and the
// global -> LDS stream costs no VALU at all: `buffer_load_dwordx4 ... lds` with a per-lane offset register that is
// bumped once per K-tile, per-instruction SGPR row offsets, and hardware range checking (rows past M / N read as zero).
// This is the shape the vendor library's hand-scheduled kernels use on this chip (same macro tile, same wave count);
// the schedule below is ours.  Used from 256 tiles (one full round of the CUs) up — see the dispatch at the end of the file.
// Measured (4096 x 4096 x 32768, steady state): this kernel 1100 TFLOP/s, the 8-wave kernel 1000, hipBLASLt 1350; compile-time
// ablations of THIS loop: MFMA stream alone 1860, DMA stream alone (no barriers) 0.63 ms = 13.6 TB/s L2->LDS = "1745", DMA +
// barriers without any MFMA 0.92-0.96 ms = the full kernel's time.  I.e. the matrix pipe is ~50 % idle and the global->LDS
// stream (64 KB per K-tile and CU, whose round trip under load is ~1.8 us against a 1.2 us K-tile) sets the pace; issuing the
// DMAs in a burst is worse (938-1046) than spread over the K-tile (1100), more lead does not help, a staggered K start loses.
// What decides the model's short-K shapes is the tile boundary: the epilogue issues all its loads before its stores (a load
// behind stores can only be awaited with vmcnt(0)) and the first k-step of a tile writes its accumulators (C = 0 inline), so
// nothing is zeroed or spilled: q|k 590 -> 924, out+gate 509 -> 815, FF2+gate 961 -> 1150 TFLOP/s.
//
// LDS: an operand tile (256 rows x 64 k, 128 B per row) is 32 pieces; piece d = 16 h + r holds rows 128 h + 16 i + r
// (i = 0..7) back to back, 1024 B + pad (W4_PAD, see there: 32 B since round 5), i.e. row -> (16 h + r) * W4_PAD + 128 i.  One DMA instruction
// fills one piece (lane j: row-tile i = j >> 3, 16-B chunk j & 7: eight 128-B global segments).  A fragment read (16 rows r = lane & 15 of
// row-tile i, k-chunk 4 s + lane>>4) is lane_base + 128 i + 64 s: every offset an immediate, conflict-free with the 32-B pad.
// 2 stages x 2 operands x 33 792 B = 135 168 B.
//
// Pipeline (K-tile t in stage t & 1; there are no staging registers, so a stage is free as soon as its fragments are in
// registers, and a third fragment buffer lets ALL of K-tile t+1's fragments be read during the last quarter of K-tile t):
//   top of K-tile t: barrier (everyone holds K-tile t's fragments); DMA K-tile t+2 into stage t&1, all 16 instructions early
//   m = 88:          vmcnt(16) = K-tile t+1 landed; barrier; its 32 fragment reads under the last 32 MFMAs
// The K-tile stream runs on across output tiles (persistent workgroup): the DMAs of the last two iterations already
// fetch the next tile, whose first fragments are read while the current tile's last MFMAs run; the epilogue sits between.
// ---------------------------------------------------------------------------------------------
// Piece stride.  Rounds 2-5 used 1024 + 16 B on the assumption that a ds_read_b128 is served in four groups of 16 CONSECUTIVE lanes; the hardware's groups
// are {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32) (MI355X_MICROARCH.md §LDS), in which the 16-B pad puts two lanes of every group on the same four
// banks: SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE, 8 LDS cycles per fragment read instead of 4 (profiles/r05_gemm_pmc.md).  With a 32-B pad the
// slot of lane (l15, lc) is (2 l15 + lc) mod 16: the lc = 0 lanes of a group take the even slots, the lc = 1 lanes the odd ones — conflict-free.
#ifndef W4_PIECE_PAD
#define W4_PIECE_PAD 1056
#endif
constexpr int W4_PAD = W4_PIECE_PAD, W4_OP = 32 * W4_PAD, W4_STAGE = 2 * W4_OP, W4_LDS = 2 * W4_STAGE;
// Round 5 (second half) — the 128-row form (MT = 4) runs THREE stages.  Its X image is half of the 256-row one (16 pieces), so a stage is 50 688 B (48 pieces of 1056 B: the 32-B piece pad of round 5) and
// three of them (149 760 B + the epilogue's 4 KB) fit the 160-KB LDS: K-tile t + 3 is requested during K-tile t (two K-tiles of lead instead of one).
// Same K order, same MFMA sequence: bit-identical outputs.  Measured (profiles/r05_gemm_mt4_three_stage_ab.log): +1-6 % on warm operands, -2.3 % of a
// config-1 step through the engine, where every projection meets its weights for the first time since the last step.  It is a small gain because the
// lead was a small part of the K-tile's 0.8-0.9 us: 48 KB per K-tile at the 62-72 GB/s a CU can pull out of its L2 (tools/probes/ldsdma_rate.hip) is
// 0.69 us — this form is bound by the fill rate, not by the 0.43 us of MFMAs (HISTORY.md §R5).  -DW4_MT4_NST=2 builds the two-stage form (A/B).
#ifndef W4_MT4_NST
#define W4_MT4_NST 3
#endif
constexpr int w4_stages(int mt) { return mt == 4 ? W4_MT4_NST : 2; }
constexpr int w4_stage_bytes(int mt) { return (mt == 4 && W4_MT4_NST == 3) ? W4_OP + 16 * W4_PAD : W4_STAGE; }
constexpr int w4_ops_bytes(int mt) { return w4_stages(mt) * w4_stage_bytes(mt); }
static_assert(w4_ops_bytes(4) + 4 * 1024 <= 160 * 1024 && w4_ops_bytes(8) == W4_LDS, "LDS budget of the four-wave kernel");
#ifdef W4_TRACE
constexpr int W4_TRACE_N = 192, W4_TRACE_BYTES = W4_TRACE_N * 8;   // 64 tiles x 3 stamps per workgroup
#else
constexpr int W4_TRACE_BYTES = 0;
#endif
constexpr int W4_PF_BYTES = 4 * 1024;   // EPI_GATE: per-wave bias | gate vectors of the epilogue
typedef __attribute__((address_space(3))) void w4_lds_t;

// Round 5 — the token side of the tile is a template parameter: MT = 16-row token tiles per wave = 8 (256-row workgroup tile, the shape described
// above), 6 (192 rows) or 4 (128 rows); the weight side stays 256 wide.  Small launches are decided by tile quantisation, not by the K loop: an
// 8-GPU token shard (5952 rows) of an N = 1792 projection is 168 tiles of 256 x 256 on 256 CUs but 217 of 192 x 256 — one round at 3/4 of the
// work; BASELINE config 1 (3328 rows) is 91 tiles of 256 x 256 but 182 of 128 x 256 — one round at half the work (the dispatch at the end of the
// file picks by cost).  A K-tile is 16 MT MFMAs per wave; fragment reads per MFMA grow from 16/64 to 14/48 and 12/32, the W-operand stream per
// MFMA by 4/3 and 2, which is why the big launches stay on MT = 8.  MT = 6 keeps the 256-row X image in LDS (its DMA instructions gather rows
// 16 apart, so the unused quarter cannot be skipped per instruction: 12.5 % more global -> LDS bytes) and wave wm = 1 reads rows 96..191 through
// two base registers; MT = 4 stages only the first 128-row half of the image (four X pieces per wave instead of eight).
template <int EPI, int MT = 8>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_bf16_w4_kernel(GemmP p) {
  static_assert(MT == 8 || MT == 6 || MT == 4, "token tiles per wave");
  static_assert(EPI != K5_EPI_F32 || MT == 8, "the frame-causal tile walk is written for 256-row tiles");
  constexpr int TM = 32 * MT;            // token rows of the workgroup tile
  constexpr int NM = 16 * MT, HM = 8 * MT;   // MFMAs per K-tile / per k-step and wave
  constexpr int XD = MT == 4 ? 4 : 8;    // X-operand DMA instructions per wave and K-tile
  constexpr int ND = 8 + XD;             // DMA instructions per wave and K-tile
  constexpr int NST = w4_stages(MT), STG = w4_stage_bytes(MT), OPS = NST * STG;   // LDS stages, bytes per stage, operand image (see w4_stages)
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1;
  const int l15 = lane & 15, lc = lane >> 4;

  // persistent walk over logical tiles, as in the 8-wave kernel
  const int nblk = p.lid_limit;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = (gridDim.x + 7 - xcd) >> 3;
  const int q8 = nblk >> 3, r8 = nblk & 7;
  const int x_first = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, x_cnt = q8 + (xcd < r8 ? 1 : 0);
  constexpr int GM = 4;
  const int per_group = GM * p.tiles_n;
  auto tile_origin = [&](int lid, int& m0, int& n0) {
    if (EPI == K5_EPI_F32 && p.causal_hw > 0) {
      // frame-causal scores: m-tile i keeps its first c(i) = ceil(limit(last row of the tile) / 256) n-tiles; lid indexes the kept
      // tiles row by row (a scalar walk over <= tiles_m rows, twice per 256x256 tile: noise)
      int i = 0;
      for (;; ++i) {
        const int last = min((i + 1) * K8_BM, p.M) - 1;
        const int c = min(p.tiles_n, (min(p.N, (last / p.causal_hw + 1) * p.causal_hw) + K8_BN - 1) / K8_BN);
        if (lid < c || i + 1 >= p.tiles_m) break;
        lid -= c;
      }
      m0 = i * K8_BM; n0 = lid * K8_BN;
      return;
    }
    const int g = lid / per_group, first_m = g * GM;
    const int gsz = min(p.tiles_m - first_m, GM);
    m0 = (first_m + (lid % per_group) % gsz) * TM;
    n0 = ((lid % per_group) / gsz) * K8_BN;
  };
  const int nk = p.K / BK;                     // even, >= 4 (launcher)
  // ---- the workgroup's walk as SEGMENTS (tile, first K-tile pair, end pair).  Without stream-K: whole tiles x_first + slot + r per_xcd.
  // Split-K TAIL (p.sk_ws; round 6): the whole rounds of an XCD run as before, in lockstep; the rem_x < per_xcd tiles that are left — the ragged
  // round, which costs a whole round's latency chain whoever computes it (43 of 255 us on an N = K = 1792 projection, 150 of 906 on FF2) — are cut
  // along K into S = min(per_xcd / rem_x, sk_smax) ALIGNED slices, slice s of tile t on workgroup t + s rem_x of the XCD.  The LAST slice owns the
  // tile: the helpers store their raw fp32 accumulators and raise a flag, the owner adds them to its own (slice order: deterministic) and runs the
  // epilogue.  (The owner is the workgroup with the HIGHEST index of the tile's team: a workgroup only ever waits for workgroups that were dispatched
  // before it and that wait for nobody, so the wait cannot deadlock however few CUs the launch gets — two such GEMMs of two streams side by side.)  All S rem_x workgroups start their slices together and the workgroups of one slice index walk K in lockstep, so a panel's K-tile is
  // still fetched once per XCD — the first form of this, the data-parallel + two-tile stream-K hybrid of Osama et al. (equal runs of 1 + rem_x /
  // per_xcd tiles per workgroup), put every workgroup on its own K phase and ran its region at HALF the K-tile rate (profiles/r06_streamk_*.log:
  // never faster than whole tiles, 30-45 % slower at most remainders; the operand stream of this kernel lives on the lockstep).  Helper and owner
  // sit on the SAME XCD (workgroup b runs on XCD b & 7, tools/probes/xcc_map.hip; the flag carries the hardware XCC id and a mismatch traps), so
  // the sums travel through that XCD's L2.  Possible since the accumulators are physical AGPRs the register allocator does not see (above).  The
  // sum over K of a split tile is ((s_last + s0) + s1) + ...: the last fp32 bit of some sums differs from the whole-tile schedule.
  const int np = nk >> 1;
  const bool sk_on = EPI != K5_EPI_F32 && p.sk_ws != nullptr;
  const int r_dp = x_cnt / per_xcd, rem_x = x_cnt - r_dp * per_xcd;
  int S = 1;
  if (sk_on && rem_x > 0) { S = min(min(per_xcd / rem_x, p.sk_smax), np); if (S < 2) S = 1; }
  const bool sk = S >= 2;
  const int sk_t = sk ? slot % rem_x : 0, sk_sl = sk ? slot / rem_x : 0;   // this workgroup's tail tile and K slice (if slot < S rem_x)
  const int nseg = sk ? r_dp + (slot < S * rem_x ? 1 : 0) : (slot < x_cnt ? (x_cnt - slot + per_xcd - 1) / per_xcd : 0);
  if (nseg == 0) return;
  auto seg = [&](int si, int& lid, int& kb, int& ke) {   // segment si of this workgroup: logical tile, K-tile pairs [kb, ke)
    if (sk && si >= r_dp) {
      lid = x_first + r_dp * per_xcd + sk_t;
      kb = sk_sl * np / S; ke = (sk_sl + 1) * np / S;
    } else {
      lid = x_first + si * per_xcd + slot; kb = 0; ke = np;
    }
  };
  const uint32_t ldw2 = (uint32_t)p.ldw * 2u, lda2 = (uint32_t)p.lda * 2u;
  const uint32_t vw0 = (uint32_t)(16 * (lane >> 3)) * ldw2 + (uint32_t)(lane & 7) * 16u;
  const uint32_t vx0 = (uint32_t)(16 * (lane >> 3)) * lda2 + (uint32_t)(lane & 7) * 16u;
  const int drow = 128 * (wave >> 1) + 8 * (wave & 1);   // first row of this wave's 8 DMA instructions per operand
  const int dslot = 8 * wave;                            // ... and their LDS piece
  const int drow_x = MT == 4 ? 4 * wave : drow, dslot_x = MT == 4 ? 4 * wave : dslot;   // MT = 4: the 16 pieces of the first 128 rows, 4 per wave
  uint32_t vw = vw0, vx = vx0;                           // lane offsets incl. the K advance of the DMA cursor
  __amdgpu_buffer_rsrc_t rW, rX;
  int d_si = 0, d_cnt = 0, d_len = 0;                    // DMA cursor: segment, K-tiles issued of it, K-tiles in it
  // (a staggered K start per tile — every tile walking K from its own offset, wrapping — was measured in round 2: 1, 2, 4 K-tiles per tile index
  // all LOSE 1-6 %: lockstep workgroups share L2 fills)
  auto set_dma_seg = [&](int si) {
    int lid, kb, ke, m0, n0;
    seg(si, lid, kb, ke);
    tile_origin(lid, m0, n0);
    const int rows_w = min(p.N - n0, K8_BN), rows_x = min(p.M - m0, MT == 4 ? 128 : K8_BM);
    rW = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.W) + (size_t)n0 * ldw2), 0,
                                           (int)(((uint32_t)(rows_w - 1) * (uint32_t)p.ldw + (uint32_t)p.K) * 2u), 0x00020000);
    rX = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.A) + (size_t)m0 * lda2), 0,
                                           (int)(((uint32_t)(rows_x - 1) * (uint32_t)p.lda + (uint32_t)p.K) * 2u), 0x00020000);
    d_cnt = 0; d_len = 2 * (ke - kb);
    vw = vw0 + (uint32_t)kb * (4 * BK); vx = vx0 + (uint32_t)kb * (4 * BK);     // a K-tile is 2 BK bytes of a row, a pair 4 BK
  };
#ifndef W4_AUX
#define W4_AUX 0   // cache policy of the operand DMAs (A/B macro: sc0 = 1, nt = 2, sc1 = 16 — none of them moved the stream)
#endif
#ifndef W4_AUX_W   // ... per operand (round 5: the weight panels stream through the L2 46 times per FF1 launch and evict the token panels a round would re-use)
#define W4_AUX_W W4_AUX
#endif
#ifndef W4_AUX_X
#define W4_AUX_X W4_AUX
#endif
  auto dma_w = [&](int so) {     // so: byte offset of the stage
#pragma unroll
    for (int jj = 0; jj < 8; ++jj)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (w4_lds_t*)(dsm + so + (dslot + jj) * W4_PAD), 16, vw,
                                               (uint32_t)(drow + jj) * ldw2, 0, W4_AUX_W);
  };
  auto dma_x = [&](int so) {
#pragma unroll
    for (int jj = 0; jj < XD; ++jj)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (w4_lds_t*)(dsm + so + W4_OP + (dslot_x + jj) * W4_PAD), 16, vx,
                                               (uint32_t)(drow_x + jj) * lda2, 0, W4_AUX_X);
  };
  // after a K-tile's 16 DMAs: move the cursor; past the last segment it wraps onto the same segment (harmless loads that keep
  // the vmcnt arithmetic uniform; nothing reads them)
  auto dma_advance = [&]() {
    vw += 2 * BK; vx += 2 * BK;
    if (++d_cnt == d_len) {
      if (d_si + 1 < nseg) ++d_si;
      set_dma_seg(d_si);
    }
  };

  // fragment read addresses (LDS byte addresses; one base register per operand and stage, everything else an immediate)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(w4_lds_t*)dsm;
  // X rows of wave wm: 16 MT wm + 16 j + l15.  Row q of the image sits in piece 16 (q >> 7) + (q & 15) at slot (q >> 4) & 7, so for MT = 8 / 4 one base
  // (+ 128 j) does, and for MT = 6 wave 1 (rows 96 .. 191) needs two: slots 6, 7 of the first 16 pieces (j < 2), slots 0 .. 3 of the second (j >= 2).
  uint32_t wbs[2], xbs[2], xbs2[2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    wbs[st] = lds0 + st * STG + (16 * wn + l15) * W4_PAD + lc * 16;
    const uint32_t h0 = lds0 + st * STG + W4_OP + l15 * W4_PAD + lc * 16;
    xbs2[st] = 0;
    if (MT == 8) xbs[st] = h0 + wm * (16 * W4_PAD);
    else if (MT == 6) { xbs[st] = h0 + wm * 768; xbs2[st] = h0 + wm * (16 * W4_PAD - 256); }
    else xbs[st] = h0 + wm * 512;
    asm volatile("" : "+v"(wbs[st]), "+v"(xbs[st]));   // keep them in registers: the stage offset does not fit the 16-bit immediate
    if (MT == 6) asm volatile("" : "+v"(xbs2[st]));
  }
#define W4_XB(ST, J) ((MT == 6 && (J) >= 2) ? xbs2[ST] : xbs[ST])
  // fragment registers: k-step 0 of the current K-tile, and k-step 1 in one of two buffers (the other one receives the NEXT
  // K-tile's k-step 1 while this one is in use; k-step 0 of the next K-tile goes to wf0/xf0, dead after the first 64 MFMAs)
  bf16x8 wf0[8], xf0[MT], wf1[2][8], xf1[2][MT];
  // Round 6: the accumulators are PHYSICAL AGPRs, invisible to the compiler — quad q = 8 (m-tile) + (n-tile) lives in a[4 q : 4 q + 3], named in
  // the instruction text of the MFMA statements and read back by v_accvgpr_read in the epilogue (acc_read).  Rounds 2-5 kept them as 64 C++
  // values bound to the AGPR file ("+a"): correct and fast, but ANY second consumer of all 64 quads (a stream-K helper's partial store, a second
  // epilogue) made the register allocator park 150-500 registers in scratch (HISTORY §R5).  Now a second consumer is just more asm.  The AGPR
  // budget is reserved by an empty statement that clobbers a0 .. a[32 MT - 1] (the allocation is sized by it, and the VGPR spiller never picks an
  // AGPR that some statement clobbers); tests/test_abi_and_host.py checks that the ISA holds no AGPR move outside these statements.
  if constexpr (MT == 8) asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
  else if constexpr (MT == 6) asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191");
  else asm volatile("" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");
  // quad q -> four VGPRs (the epilogue's one way to an accumulator); Q is a constant after unrolling
#define W4_ACC(DST, Q) asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]" \
                                    : "=v"(DST[0]), "=v"(DST[1]), "=v"(DST[2]), "=v"(DST[3]) : "n"(4 * (Q)), "n"(4 * (Q) + 1), "n"(4 * (Q) + 2), "n"(4 * (Q) + 3))

  // ds_read_b128 as asm: the compiler would otherwise guard every fragment read with s_waitcnt vmcnt(..) against the LDS-DMA
  // writes in flight (it cannot tell the stages apart) and serialise the prefetch.  All waits are explicit below.
#define W4_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
  // The matrix instructions are inline asm with the accumulators pinned to the AGPR file: with the builtin the register
  // allocator mixes the two files and shuffles ~500 registers per two K-tiles.  volatile asm statements keep their order, and
  // loads cannot cross them, so the source order below IS the instruction schedule: fragment reads and DMAs are placed
  // between the MFMAs by hand.  (An accumulator is revisited 64 MFMAs later: no dependent-issue hazard inside the stream.)
#define W4_MF(WF, XF, Q) asm volatile("v_mfma_f32_16x16x32_bf16 a[%2:%3], %0, %1, a[%2:%3]" :: "v"(WF[(Q) & 7]), "v"(XF[(Q) >> 3]), "n"(4 * (Q)), "n"(4 * (Q) + 3))
  // first k-step of an output tile: C = 0 as an inline constant, the accumulator is only written — so the epilogue never has
  // to create 256 zeroed registers while the old sums are still live (which made the allocator park them in scratch)
#define W4_MF0(WF, XF, Q) asm volatile("v_mfma_f32_16x16x32_bf16 a[%2:%3], %0, %1, 0" :: "v"(WF[(Q) & 7]), "v"(XF[(Q) >> 3]), "n"(4 * (Q)), "n"(4 * (Q) + 3))
#ifndef W4_DBG
#define W4_DBG 0
#endif
  constexpr int dbg = W4_DBG;   // compile-time ablations (benchmarking only): 1 no DMA, 2 no fragment reads, 4 no barriers, 8 no MFMA
  auto dma1 = [&](int so, int d) {
    if (dbg & 1) return;
    if (d < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (w4_lds_t*)(dsm + so + (dslot + d) * W4_PAD), 16, vw,
                                                        (uint32_t)(drow + d) * ldw2, 0, W4_AUX_W);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (w4_lds_t*)(dsm + so + W4_OP + (dslot_x + d - 8) * W4_PAD), 16, vx,
                                                  (uint32_t)(drow_x + d - 8) * lda2, 0, W4_AUX_X);
  };

  set_dma_seg(0);
#pragma unroll
  for (int st = 0; st < NST; ++st) { dma_w(st * STG); dma_x(st * STG); dma_advance(); }
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ND * (NST - 1)) : "memory");   // K-tile 0 has landed
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 8; ++i) { W4_RD(wf0[i], wbs[0], i * 128); W4_RD(wf1[0][i], wbs[0], i * 128 + 64); }
#pragma unroll
  for (int j = 0; j < MT; ++j) { W4_RD(xf0[j], W4_XB(0, j), j * 128); W4_RD(xf1[0][j], W4_XB(0, j), j * 128 + 64); }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // (schedule 2 re-reads the k-step-1 fragments at the start of the K-tile; the extra 16 reads here happen once per workgroup)

  // One K-tile t (stage st = t & 1) = 128 MFMAs, m = 0..127: k-step 0 (m < 64) on wf0/xf0, k-step 1 on wf1[st]/xf1[st].  ALL of
  // its fragments are in registers when it starts, so its stage is refilled from the top:
  //   before m = 0        barrier: every wave holds K-tile t's fragments -> stage st is free
  //   m = W4_DS d         DMA d (0..15) of K-tile t+2 into stage st, spread over the K-tile (a burst is worse: the wave stalls at a
  //                       VMEM instruction the texture path cannot take yet, and its SIMD's matrix pipe with it)
  //   after m = W4_BB-1   vmcnt(n) + barrier: all but this tile's own n DMAs so far, i.e. K-tile t+1 (issued a K-tile ago), have landed
  //   from m = W4_BB      a fragment read of K-tile t+1 every W4_RS MFMAs: k-step 0 into wf0/xf0, k-step 1 into wf1[st^1]/xf1[st^1]
  //   after m = 127       lgkmcnt(0)
#ifndef W4P
#define W4P 8, 80, 1   // round 5 (after the 32-B piece pad: a fragment read is 4 LDS cycles, back-to-back reads no longer saturate the LDS): reads every MFMA, barrier at m = 80 — -0.8 % per block against 8, 64, 2 (profiles/r05_gemm_piece_pad_ab.log); rounds 2-5: read spacing 1 -> 2 was +4-6 % with the conflicting layout
#endif
#ifndef W4_BB6
#define W4_BB6 40
#endif
#ifndef W4_BB4
#define W4_BB4 16
#endif
  constexpr int w4p[3] = {W4P};
  // MT = 6: 96 MFMAs, 16 DMAs 6 apart, barrier after m = 39, k-step-1 reads (14) from m = 40, k-step-0 reads from m = 68; MT = 4: 64 MFMAs, 12 DMAs 5
  // apart, barrier after m = 15, reads (12 + 12) from m = 16 and m = 40
  constexpr int W4_DS = MT == 8 ? w4p[0] : NM / ND, W4_BB = MT == 8 ? w4p[1] : (MT == 6 ? W4_BB6 : W4_BB4), W4_RS = w4p[2];   // DMA spacing, barrier position, fragment-read spacing
  static_assert(W4_DS * (ND - 1) < NM && W4_BB >= 1, "schedule does not fit");
  constexpr int W4_NB = (W4_BB + W4_DS - 1) / W4_DS < ND ? (W4_BB + W4_DS - 1) / W4_DS : ND;   // this K-tile's DMAs issued before m = W4_BB
  constexpr int NR = 8 + MT;   // fragment reads per k-step
  // NST stages (2, or 3 on the 128-row form): K-tile t sits in stage t % NST, K-tile t + NST is requested during K-tile t, and the wait below leaves
  // this K-tile's own DMAs so far AND the NST - 2 whole K-tiles requested after K-tile t + 1 in flight.  The k-step-1 fragment buffer alternates with
  // the K-tile's parity (st: compile time, nk is even so every output tile starts on buffer 0).  With two stages the LDS stage is that parity too; with
  // three it ROTATES AT RUN TIME — o_cur (scalar: the stage this K-tile's DMAs refill) and wb_nx / xb_nx (the fragment-read bases of the stage after
  // it) move on at the end of every K-tile — so the K loop stays the same straight line unrolled by two: any branch on the stage around the asm MFMA
  // stream makes the register allocator park the accumulators (286 spilled VGPRs with a three-way dispatch per K-tile).
  int o_cur = 0;
  uint32_t wb_nx = wbs[0] + STG, xb_nx = xbs[0] + STG;     // (three stages; MT = 4 reads X through one base)
  auto ktile = [&](auto STC, auto FIRSTC) {
    constexpr int st = decltype(STC)::value, nx = st ^ 1;
    constexpr bool first = decltype(FIRSTC)::value;
    const int so = NST == 2 ? st * STG : o_cur;
    const uint32_t wbn = NST == 2 ? wbs[nx] : wb_nx;
    auto xbn = [&](int j) -> uint32_t { return NST == 2 ? W4_XB(nx, j) : xb_nx; };
    if (!(dbg & 4)) asm volatile("s_barrier" ::: "memory");
    auto chunk = [&](auto BASEC) {   // 16 MFMAs at a time: a single 128-trip loop is beyond the full-unroll budget
#pragma unroll
      for (int m = decltype(BASEC)::value; m < decltype(BASEC)::value + 16; ++m) {
        if (!(dbg & 8)) {
          if (m < HM && first) W4_MF0(wf0, xf0, m);
          else if (m < HM) W4_MF(wf0, xf0, m);
          else W4_MF(wf1[st], xf1[st], m - HM);
        }
        if (m % W4_DS == 0 && m / W4_DS < ND) dma1(so, m / W4_DS);
        if (m == W4_BB - 1) {
          if (dbg & 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W4_NB + ND * (NST - 2)) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(W4_NB + ND * (NST - 2)) : "memory");
        }
        // fragment reads of K-tile t+1, W4_RS MFMAs apart (back to back they saturate the LDS: four waves x 1 KB per 16 cycles).
        // k-step 1 goes to the idle buffer and may start at the barrier; k-step 0 reuses wf0/xf0, free from m = 64
        if (!(dbg & 2)) {
          constexpr int K1_AT = W4_BB >= HM ? W4_BB + NR * W4_RS : W4_BB;                       // first k-step-1 read
          constexpr int K0_AT = W4_BB >= HM ? W4_BB : (W4_BB + NR * W4_RS > HM ? W4_BB + NR * W4_RS : HM);
          static_assert(K0_AT + (NR - 1) * W4_RS < NM && K1_AT + (NR - 1) * W4_RS < NM, "fragment reads do not fit");
          if (m >= K0_AT && m < K0_AT + NR * W4_RS && (m - K0_AT) % W4_RS == 0) {
            const int r = (m - K0_AT) / W4_RS;
            if (r < 8) W4_RD(wf0[r & 7], wbn, (r & 7) * 128);
            else W4_RD(xf0[r - 8], xbn(r - 8), (r - 8) * 128);
          }
          if (m >= K1_AT && m < K1_AT + NR * W4_RS && (m - K1_AT) % W4_RS == 0) {
            const int r = (m - K1_AT) / W4_RS;
            if (r < 8) W4_RD(wf1[nx][r & 7], wbn, (r & 7) * 128 + 64);
            else W4_RD(xf1[nx][r - 8], xbn(r - 8), (r - 8) * 128 + 64);
          }
        }
      }
    };
    chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 16>{}); chunk(std::integral_constant<int, 32>{});
    chunk(std::integral_constant<int, 48>{});
    if constexpr (MT >= 6) { chunk(std::integral_constant<int, 64>{}); chunk(std::integral_constant<int, 80>{}); }
    if constexpr (MT >= 8) { chunk(std::integral_constant<int, 96>{}); chunk(std::integral_constant<int, 112>{}); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // K-tile t+1's fragments
    dma_advance();
    if constexpr (NST == 3) {
      const bool wrap = o_cur == STG;      // the stage after o_cur is the last one -> the one after that is stage 0
      o_cur = o_cur == 2 * STG ? 0 : o_cur + STG;
      wb_nx = wrap ? wb_nx - 2 * STG : wb_nx + STG;
      xb_nx = wrap ? xb_nx - 2 * STG : xb_nx + STG;
    }
  };

#ifndef W4_SCHED
#define W4_SCHED 1
#endif
  // Schedule 2 (W4_SCHED=2): ONE fragment set.  K-tile t (stage st): its k-step-0 fragments are in registers (read at the end
  // of K-tile t-1), its k-step-1 fragments are read at the START of the tile — W's first, then barrier -> the W half of the
  // stage is free and W(t+2) streams in while X's k-step-1 fragments are read, barrier -> X(t+2) — and the k-step-0 fragments of
  // K-tile t+1 are read in the LAST quarter, after the one vmcnt + barrier that says "K-tile t+1 has landed".  Against
  // schedule 1 the landing deadline of a K-tile moves from m = 63 to m = S2_WAIT of the tile before it is used (the last DMA of a
  // tile gets ~50 % more time), for one more barrier per K-tile and 64 fewer live registers.
  constexpr int S2_R1 = 0, S2_RS1 = 2;          // W k-step-1 reads at m = 0, 2, .., 14
  constexpr int S2_B1 = 17;                     // lgkmcnt(0) + barrier after m = 17: W half of the stage is free
  constexpr int S2_DW = 18, S2_DS = 3;          // W DMAs at m = 18, 21, .., 39   (8)
  constexpr int S2_R2 = 19;                     // X k-step-1 reads at m = 19, 21, .., 33
  constexpr int S2_B2 = 41;                     // lgkmcnt(0) + barrier after m = 41: X half free
  constexpr int S2_DX = 42, S2_DXS = 7;         // X DMAs at m = 42, 49, .., 91    (8)
  constexpr int S2_WAIT = 97;                   // vmcnt(16) + barrier after m = 97: K-tile t+1 (issued a K-tile ago) landed
  constexpr int S2_R0 = 98, S2_R0S = 2;         // k-step-0 reads of K-tile t+1: W at m = 98, 100, .. 112 and X at m = 99, 101, .. 113
  static_assert(S2_DX + 7 * S2_DXS < S2_WAIT, "all 16 DMAs of the K-tile are issued before the wait (vmcnt(16) = the tile before)");
  auto ktile2 = [&](auto STC, auto FIRSTC) {
    static_assert(MT == 8, "schedule 2 is written for the 256-row tile");
    constexpr int st = decltype(STC)::value;
    constexpr bool first = decltype(FIRSTC)::value;
    auto chunk = [&](auto BASEC) {
#pragma unroll
      for (int m = decltype(BASEC)::value; m < decltype(BASEC)::value + 16; ++m) {
        if (!(dbg & 8)) {
          if (m < 64 && first) W4_MF0(wf0, xf0, m);
          else if (m < 64) W4_MF(wf0, xf0, m);
          else W4_MF(wf1[0], xf1[0], m - 64);
        }
        if (!(dbg & 2)) {
          if (m >= S2_R1 && m < S2_R1 + 8 * S2_RS1 && (m - S2_R1) % S2_RS1 == 0) W4_RD(wf1[0][(m - S2_R1) / S2_RS1], wbs[st], ((m - S2_R1) / S2_RS1) * 128 + 64);
          if (m >= S2_R2 && m < S2_R2 + 8 * S2_RS1 && (m - S2_R2) % S2_RS1 == 0) W4_RD(xf1[0][(m - S2_R2) / S2_RS1], xbs[st], ((m - S2_R2) / S2_RS1) * 128 + 64);
          if (m >= S2_R0 && m < S2_R0 + 8 * S2_R0S && (m - S2_R0) % S2_R0S == 0) W4_RD(wf0[(m - S2_R0) / S2_R0S], wbs[st ^ 1], ((m - S2_R0) / S2_R0S) * 128);
          if (m >= S2_R0 + 1 && m < S2_R0 + 1 + 8 * S2_R0S && (m - S2_R0 - 1) % S2_R0S == 0) W4_RD(xf0[(m - S2_R0 - 1) / S2_R0S], xbs[st ^ 1], ((m - S2_R0 - 1) / S2_R0S) * 128);
        }
        if (m == S2_B1 || m == S2_B2) {
          if (dbg & 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (m >= S2_DW && m < S2_DW + 8 * S2_DS && (m - S2_DW) % S2_DS == 0) dma1(st, (m - S2_DW) / S2_DS);
        if (m >= S2_DX && m < S2_DX + 8 * S2_DXS && (m - S2_DX) % S2_DXS == 0) dma1(st, 8 + (m - S2_DX) / S2_DXS);
        if (m == S2_WAIT) {
          if (dbg & 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
        }
      }
    };
    chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 16>{}); chunk(std::integral_constant<int, 32>{});
    chunk(std::integral_constant<int, 48>{}); chunk(std::integral_constant<int, 64>{}); chunk(std::integral_constant<int, 80>{});
    chunk(std::integral_constant<int, 96>{}); chunk(std::integral_constant<int, 112>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // K-tile t+1's k-step-0 fragments
    dma_advance();
  };
#if W4_SCHED == 2
#define W4_KT ktile2
#else
#define W4_KT ktile
#endif

#ifdef W4_TRACE   // tools/gemm_w4_trace.py: wall-clock (100 MHz) stamps per workgroup and tile: tile start | K loop done | epilogue done
  unsigned long long* tr_lds = reinterpret_cast<unsigned long long*>(dsm + OPS);
  int tr_i = 0;
  if (tid == 0) tr_lds[W4_TRACE_N - 2] = __builtin_amdgcn_s_memtime();   // shader-clock ticks at the first / after the last tile: the clock this kernel ran at
#define W4_STAMP() do { if (tid == 0 && tr_i < W4_TRACE_N) tr_lds[tr_i] = __builtin_amdgcn_s_memrealtime(); ++tr_i; } while (0)
#else
#define W4_STAMP() do {} while (0)
#endif
  for (int si = 0; si < nseg; ++si) {
    int s_lid, s_kb, s_ke;
    seg(si, s_lid, s_kb, s_ke);
    W4_STAMP();
    W4_KT(std::integral_constant<int, 0>{}, std::true_type{});
    W4_KT(std::integral_constant<int, 1>{}, std::false_type{});
    for (int t = 1; t < s_ke - s_kb; ++t) {
      W4_KT(std::integral_constant<int, 0>{}, std::false_type{});
      W4_KT(std::integral_constant<int, 1>{}, std::false_type{});
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the asm MFMAs are invisible to the hazard recogniser: let the last ones retire
    W4_STAMP();
    int m0, n0;
    tile_origin(s_lid, m0, n0);
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int e_l15 = tid2 & 15, e_lc = (tid2 >> 4) & 3, e_wave = tid2 >> 6;
    const int e_wn = e_wave & 1, e_wm = e_wave >> 1;
    // ---- split-K tail (see the walk above).  Helper (slice > 0): the 8 MT accumulator quads go to the workspace straight from the AGPR file, 1 KB
    // per wave and instruction, then this wave raises its flag (the owner's wave w needs only the helpers' wave w: same lane <-> output map) — no
    // epilogue.  Owner (the last slice): for every helper in slice order, wait for this wave's flag and add the helper's sums into the AGPRs quad by quad;
    // then the ordinary epilogue.  Once per workgroup and launch at most.
    // The partial sums move with PLAIN accesses: the helper's stores are write-through in its L1 and acknowledged by the XCD's L2 (vmcnt(0) before
    // the flag), the owner's CU has not touched these lines since its L1 was invalidated at the dispatch, so its loads miss the L1 and find them in
    // that L2 (-DK5_SK_SC1: agent-scope accesses instead; same time).  The flags are agent-scope atomics.
#ifdef K5_SK_SC1
#define W4_SK_SC " sc1"
#else
#define W4_SK_SC ""
#endif
    if constexpr (EPI != K5_EPI_F32) {
      if (sk && si >= r_dp) {
        const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane(e_wave);
        uint32_t sk_xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(sk_xcc));
        sk_xcc = (sk_xcc & 15u) + 1u;
        uint32_t* flags = reinterpret_cast<uint32_t*>(p.sk_ws + (size_t)gridDim.x * (4 * 64 * 256));     // behind the partial tiles (zeroed once by the launcher, and by every owner after use)
        if (sk_sl < S - 1) {   // helper
          const float* part = p.sk_ws + ((size_t)blockIdx.x * 4 + wv) * (64 * 256) + (tid2 & 63) * 4;
#pragma unroll
          for (int q = 0; q < 8 * MT; ++q)
            asm volatile("global_store_dwordx4 %0, a[%1:%2], off" W4_SK_SC :: "v"(part + (size_t)q * 256), "n"(4 * q), "n"(4 * q + 3) : "memory");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every store of this wave acknowledged by the L2, then the flag
          if ((tid2 & 63) == 0) __hip_atomic_store(flags + (size_t)blockIdx.x * 4 + wv, sk_xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
          W4_STAMP();
          continue;
        }
        for (int hs = 0; hs < S - 1; ++hs) {   // owner: the helpers in slice order
          const uint32_t hb = (uint32_t)(xcd + 8 * (sk_t + hs * rem_x));
          const float* part = p.sk_ws + ((size_t)hb * 4 + wv) * (64 * 256) + (tid2 & 63) * 4;
          uint32_t* fp = flags + (size_t)hb * 4 + wv;
          uint32_t f;
          for (;;) {
            f = __hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            f = (uint32_t)__builtin_amdgcn_readfirstlane((int)f);
            if (f != 0u) break;
            __builtin_amdgcn_s_sleep(4);
          }
          if (f != sk_xcc) __builtin_trap();   // the helper ran on another XCD: its sums are not in this L2
          if ((tid2 & 63) == 0) __hip_atomic_store(fp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumed: zero again for the next launch
#pragma unroll
          for (int q0 = 0; q0 < 8 * MT; q0 += 8) {   // eight quads (8 KB per wave) in flight at a time
            f32x4 pq[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("global_load_dwordx4 %0, %1, off" W4_SK_SC : "=v"(pq[i]) : "v"(part + (size_t)(q0 + i) * 256) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              asm volatile("" : "+v"(pq[i]));       // nothing of pq[i] is touched before the wait above
              float a4[4];
              W4_ACC(a4, q0 + i);
              a4[0] += pq[i][0]; a4[1] += pq[i][1]; a4[2] += pq[i][2]; a4[3] += pq[i][3];
              asm volatile("v_accvgpr_write_b32 a[%4], %0\n\tv_accvgpr_write_b32 a[%5], %1\n\tv_accvgpr_write_b32 a[%6], %2\n\tv_accvgpr_write_b32 a[%7], %3"
                           :: "v"(a4[0]), "v"(a4[1]), "v"(a4[2]), "v"(a4[3]), "n"(4 * (q0 + i)), "n"(4 * (q0 + i) + 1), "n"(4 * (q0 + i) + 2), "n"(4 * (q0 + i) + 3));
            }
          }
        }
        asm volatile("s_nop 4" ::: "memory");
      }
    }
    // straight-line quads (the launcher guarantees N % 4 == 0 and 4-element aligned ldc / ldr): the only predicate is
    // "inside the matrix", so there is no control flow for the allocator to park accumulators around
    // Loads first, stores after: a load that follows stores can only be awaited with vmcnt(0), i.e. after every earlier
    // store has been acknowledged.  The per-column vectors (bias, gate) are the same for all 8 token tiles: loaded once; the
    // residual rows of the gated epilogue are fetched one token tile (8 loads) at a time.
    auto epilogue = [&](auto HB) {
      constexpr bool has_bias = decltype(HB)::value;
      const int nb = n0 + 128 * e_wn + 4 * e_lc;          // + 16 i
      if constexpr (EPI == K5_EPI_F32) {                  // raw fp32 scores: C is float*, one 16-B store per quad
        float* cf = reinterpret_cast<float*>(p.C);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          const int m = m0 + 16 * MT * e_wm + 16 * j + e_l15;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int n = nb + 16 * i;
            float a4[4];
            W4_ACC(a4, 8 * j + i);
            if (m < p.M && n < p.N)
              *reinterpret_cast<f32x4*>(cf + (size_t)m * p.ldc + n) = f32x4{a4[0] * p.alpha, a4[1] * p.alpha, a4[2] * p.alpha, a4[3] * p.alpha};
          }
        }
        return;
      }
      // per-column vectors of the wave's 128 columns.  Plain / GELU / per-row bias: 8 x 4 values per lane in registers.  Gated residual:
      // bias AND gate would be 64 VGPRs next to two register sets of residual rows and the next tile's 128 fragment registers — they live
      // in this wave's 1-KB LDS slot instead (ds_read_b128 per use: the LDS is idle in the epilogue, and DS reads are not in the vmcnt queue)
      f32x4 bvec[8];
      float* vslot = reinterpret_cast<float*>(dsm + OPS + W4_TRACE_BYTES + 1024 * e_wave);   // [bias 128 | gate 128]
      if constexpr (EPI == K5_EPI_GATE) {
        const int l = tid2 & 63;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int c = min(n0 + 128 * e_wn + 2 * l + h, p.N - 1);   // columns 2 l, 2 l + 1 of the wave's 128
          vslot[2 * l + h] = has_bias ? p.bias[c] : 0.f;
          vslot[128 + 2 * l + h] = p.gate[c];
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int nc = min(nb + 16 * i, p.N - 4);          // clamped: out-of-range quads are never stored
          if (EPI != K5_EPI_BIAS_M && has_bias) bvec[i] = *reinterpret_cast<const f32x4*>(p.bias + nc);
        }
      }
      // one token tile (16 rows) of the wave's 128 x 128 quadrant: accumulators -> bias / GELU / gated residual -> bf16 -> two n-tiles paired
      // by v_permlane16_swap so that every lane stores 16 B
      // (gated residual) bias / gate quads of n-tiles 2 iq, 2 iq + 1 from the wave's LDS slot: asm reads with counted lgkmcnt waits — a plain
      // LDS load would make the compiler wait for every LDS-DMA (the next tile's K-tiles) and store in flight first — fetched one pair ahead
      const uint32_t vaddr = (uint32_t)(uintptr_t)(w4_lds_t*)(dsm + OPS + W4_TRACE_BYTES) + 1024u * (uint32_t)e_wave + 16u * (uint32_t)e_lc;
      f32x4 bq[2][2], gq[2][2];
#define W4_VEC_FETCH(IQ, B) do { W4_RD(bq[B][0], vaddr, 128 * (IQ)); W4_RD(bq[B][1], vaddr, 128 * (IQ) + 64); \
                                W4_RD(gq[B][0], vaddr, 512 + 128 * (IQ)); W4_RD(gq[B][1], vaddr, 512 + 128 * (IQ) + 64); } while (0)
      auto finish_rows = [&](int j, float bm, const u32x2 (&rr)[8]) {
        const int m = m0 + 16 * MT * e_wm + 16 * j + e_l15;
        if constexpr (EPI == K5_EPI_GATE) W4_VEC_FETCH(0, 0);
#pragma unroll
        for (int iq = 0; iq < 4; ++iq) {     // n-tiles 2 iq and 2 iq + 1 together
          if constexpr (EPI == K5_EPI_GATE) {
            if (iq == 0) W4_VEC_FETCH(1, 1);
            if (iq == 1) W4_VEC_FETCH(2, 0);
            if (iq == 2) W4_VEC_FETCH(3, 1);
            if (iq < 3) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // the fetched quads stay LIVE up to here whether or not they are used: without a bias the bias quads are dead values to the compiler, which
            // may hand their registers to something else right behind the ds_read — and the data, arriving later, lands on top of it (round 5: wrong
            // gated outputs without a bias on the 192- / 128-row tiles; the 256-row form happened to get away with it)
            asm volatile("" : "+v"(bq[iq & 1][0]), "+v"(bq[iq & 1][1]), "+v"(gq[iq & 1][0]), "+v"(gq[iq & 1][1]));
          }
          u32x2 o[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int i = 2 * iq + h;
            float v[4];
            W4_ACC(v, 8 * j + i);
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f}, gv = bv;
            if constexpr (EPI == K5_EPI_GATE) { bv = bq[iq & 1][h]; gv = gq[iq & 1][h]; }
            else if (EPI != K5_EPI_BIAS_M && has_bias) bv = bvec[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (EPI == K5_EPI_BIAS_M) v[e] += bm;
              else if (has_bias) v[e] += bv[e];
              if (EPI == K5_EPI_GELU) v[e] = bf_round(v[e]);
            }
            if (EPI == K5_EPI_GELU) { gelu_erf_x2(v[0], v[1]); gelu_erf_x2(v[2], v[3]); }   // same bits as gelu_erf, packed fp32 math
            if (EPI == K5_EPI_GATE) {
              v[0] = __uint_as_float(rr[i][0] << 16) + gv[0] * bf_round(v[0]);
              v[1] = __uint_as_float(rr[i][0] & 0xffff0000u) + gv[1] * bf_round(v[1]);
              v[2] = __uint_as_float(rr[i][1] << 16) + gv[2] * bf_round(v[2]);
              v[3] = __uint_as_float(rr[i][1] & 0xffff0000u) + gv[3] * bf_round(v[3]);
            }
            o[h] = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          }
          // lanes l and l + 16 hold columns 4 c .. 4 c + 3 and the next four of BOTH tiles: trade (tile 2 iq + 1 of the lower
          // row) for (tile 2 iq of the upper row) and each lane owns 8 consecutive columns of ONE tile -> 16-B stores, half
          // as many (the epilogue is store-issue bound)
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const auto sw = __builtin_amdgcn_permlane16_swap(o[0][d], o[1][d], false, false);
            o[0][d] = sw[0]; o[1][d] = sw[1];
          }
#ifdef W4_STORE_ABL   // timing-only A/B (tools/build_variant.sh -DW4_STORE_ABL, results WRONG): the same bytes to the same lines, but every store
          // instruction covers 8 rows x 128 B (whole cache lines) instead of 16 rows x 64 B — what a transposing epilogue would buy
          const int m_a = m0 + 16 * MT * e_wm + 16 * j + 8 * (iq >> 1) + (e_l15 >> 1);
          const int n_a = n0 + 128 * e_wn + 64 * (iq & 1) + 8 * (4 * (e_l15 & 1) + e_lc);
          if (m_a < p.M && n_a < p.N) *reinterpret_cast<u32x4*>(p.C + (size_t)m_a * p.ldc + n_a) = u32x4{o[0][0], o[0][1], o[1][0], o[1][1]};
#else
          const int n = n0 + 128 * e_wn + 16 * (2 * iq + (e_lc & 1)) + 8 * (e_lc >> 1);
          if (m < p.M && n < p.N) *reinterpret_cast<u32x4*>(p.C + (size_t)m * p.ldc + n) = u32x4{o[0][0], o[0][1], o[1][0], o[1][1]};
#endif
        }
      };
      if constexpr (EPI == K5_EPI_GATE) {
        // The residual rows of token tile j + 1 are requested BEFORE token tile j is stored (two register sets, 32 VGPRs): a load issued
        // behind stores can only be awaited together with them, i.e. after every store has been acknowledged by L2 — with one register set
        // each of the 8 token tiles paid a store acknowledgement plus a load round trip: 16.5 us of a K = 1792 tile's 57 -> 12.1 (tools/gemm_w4_trace.py;
        // a plain-store epilogue takes 5.2).  Touching the residual tile's lines ~5 us ahead, from inside the K loop (four 4-B-per-lane LDS-DMA
        // loads per wave in K-tile nk - 4), was built and measured neutral with either epilogue (out + gate 313.8 / 314.3 / 321.1 us with,
        // 309.0 / 312.2 / 315.0 without, interleaved): every workgroup of a round reaches its epilogue within a few us of the others, the
        // 67 MB of residual reads + output writes of a round then move at ~5.6 TB/s — the epilogues are HBM-bandwidth-bound BECAUSE they
        // are in lockstep, not latency-bound.
        u32x2 rr[2][8];
        auto load_rows = [&](int j, u32x2 (&dst)[8]) {
          const int m = min(m0 + 16 * MT * e_wm + 16 * j + e_l15, p.M - 1);
#pragma unroll
          for (int i = 0; i < 8; ++i) dst[i] = *reinterpret_cast<const u32x2*>(p.resid + (size_t)m * p.ldr + min(nb + 16 * i, p.N - 4));
        };
        load_rows(0, rr[0]);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          if (j + 1 < MT) load_rows(j + 1, rr[(j + 1) & 1]);
          asm volatile("" ::: "memory");
          finish_rows(j, 0.f, rr[j & 1]);
          asm volatile("" ::: "memory");
        }
      } else {
        constexpr int G = MT == 6 ? 3 : 4;        // token tiles per phase
#pragma unroll
        for (int jh = 0; jh < MT / G; ++jh) {
          float bias_m[G];
#pragma unroll
          for (int jj = 0; jj < G; ++jj)
            bias_m[jj] = (EPI == K5_EPI_BIAS_M && has_bias) ? p.bias[min(m0 + 16 * MT * e_wm + 16 * (G * jh + jj) + e_l15, p.M - 1)] : 0.f;
          const u32x2 none[8] = {};
#pragma unroll
          for (int jj = 0; jj < G; ++jj) finish_rows(G * jh + jj, bias_m[jj], none);
          asm volatile("" ::: "memory");
        }
      }
    };
    if (p.bias) epilogue(std::true_type{}); else epilogue(std::false_type{});
    // nothing may still be loading into a VGPR when the asm stream resumes (the compiler would guard the asm's outputs with
    // vmcnt waits INSIDE the loop).  Draining the stores as well measured within noise of not draining them (the next tile's
    // first K-tiles have had the whole epilogue to land either way).
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    W4_STAMP();
  }
#ifdef W4_TRACE
  if (tid == 0) tr_lds[W4_TRACE_N - 1] = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (p.trace && tid < 64)
    for (int i = tid; i < W4_TRACE_N; i += 64) p.trace[(size_t)blockIdx.x * W4_TRACE_N + i] = (i < tr_i || i >= W4_TRACE_N - 2) ? tr_lds[i] : 0ull;
#endif
#undef W4_VEC_FETCH
#undef W4_XB
#undef W4_STAMP
#undef W4_KT
#undef W4_MF
#undef W4_MF0
#undef W4_ACC
#undef W4_SK_SC
#undef W4_RD
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wrapped DMAs still write this workgroup's LDS
}

// ---------------------------------------------------------------------------------------------
// The last, partly filled round of 256x256 tiles as 128x128 QUADRANTS with a deep prefetch (round 4).  The persistent kernels hand the
// tiles of a round that would leave more than half of the CUs idle to a small kernel, one workgroup per quadrant; until round 4 that was
// gemm_bf16_glds_kernel — one K-tile of prefetch behind a vmcnt(0) + barrier — and with at most one such workgroup per CU nothing hid
// the L2 / HBM round trip: tools/gemm_w4_trace.py measured 36 / 70 / 92 / 154 us for the tails of the out, q|k, FF1 and FF2 projections
// (22 / 44 / 88 / 22 tiles; K = 1792, 1792, 1792, 7168) = 2.5 us per K-tile, 11-15 % of those GEMMs.  Here: the w4 kernel's LDS image
// (1040-B pieces: fragment reads at immediate offsets, conflict-free) in FOUR stages of 128 + 128 rows (133 120 B), three K-tiles in
// flight under counted vmcnt waits and one raw s_barrier per K-tile; four waves, one per SIMD, each 64 x 64 = 4 x 4 MFMA 16x16x32 tiles;
// the fragment reads of k-step 1 fly under the MFMAs of k-step 0.  Same accumulation order over K as the persistent kernels' tiles
// (K-tile after K-tile, k-step 0 then 1), so a tile's numbers do not depend on which kernel computed it.
// ---------------------------------------------------------------------------------------------
constexpr int Q4_OP = 16 * W4_PAD, Q4_STAGE = 2 * Q4_OP, Q4_NST = 4, Q4_LDS = Q4_NST * Q4_STAGE;

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_q4_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1, l15 = lane & 15, lc = lane >> 4;
  // quadrant b & 3 of the 256x256 logical tile tail_base + b / 4 (the tile walk of the persistent kernels: groups of 4 m-tiles)
  const int plid = p.tail_base + (int)(blockIdx.x >> 2), q = blockIdx.x & 3;
  const int pg = plid / (4 * p.tiles256_n), pfirst = pg * 4, pgsz = min(p.tiles256_m - pfirst, 4);
  const int m0 = (pfirst + (plid % (4 * p.tiles256_n)) % pgsz) * 256 + 128 * (q >> 1);
  const int n0 = ((plid % (4 * p.tiles256_n)) / pgsz) * 256 + 128 * (q & 1);
  if (m0 >= p.M || n0 >= p.N) return;

  const int nk = p.K / BK;
  const uint32_t ldw2 = (uint32_t)p.ldw * 2u, lda2 = (uint32_t)p.lda * 2u;
  // one DMA instruction = one piece r (rows 16 i + r, i = lane >> 3, 16-B chunk lane & 7); wave w issues pieces 4 w .. 4 w + 3 of both operands
  uint32_t vw = (uint32_t)(16 * (lane >> 3)) * ldw2 + (uint32_t)(lane & 7) * 16u;
  uint32_t vx = (uint32_t)(16 * (lane >> 3)) * lda2 + (uint32_t)(lane & 7) * 16u;
  const int rows_w = min(p.N - n0, 128), rows_x = min(p.M - m0, 128);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.W) + (size_t)n0 * ldw2), 0,
                                                                       (int)(((uint32_t)(rows_w - 1) * (uint32_t)p.ldw + (uint32_t)p.K) * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.A) + (size_t)m0 * lda2), 0,
                                                                       (int)(((uint32_t)(rows_x - 1) * (uint32_t)p.lda + (uint32_t)p.K) * 2u), 0x00020000);
  auto dma = [&](int stage) {   // the next K-tile of both operands into `stage` (8 instructions per wave), then advance the cursor
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (w4_lds_t*)(dsm + stage * Q4_STAGE + (4 * wave + jj) * W4_PAD), 16, vw, (uint32_t)(4 * wave + jj) * ldw2, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (w4_lds_t*)(dsm + stage * Q4_STAGE + Q4_OP + (4 * wave + jj) * W4_PAD), 16, vx, (uint32_t)(4 * wave + jj) * lda2, 0, 0);
    }
    vw += 2 * BK; vx += 2 * BK;
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)(w4_lds_t*)dsm;
  const uint32_t wb0 = lds0 + (uint32_t)l15 * W4_PAD + (uint32_t)lc * 16u + 512u * (uint32_t)wn;            // + 128 i + 64 s (+ stage)
  const uint32_t xb0 = lds0 + Q4_OP + (uint32_t)l15 * W4_PAD + (uint32_t)lc * 16u + 512u * (uint32_t)wm;    // + 128 j + 64 s (+ stage)
#define Q4_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
  f32x4 acc[4][4];   // [n-tile][m-tile]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 wf[2][4], xf[2][4];

  // three K-tiles in flight
#pragma unroll
  for (int st = 0; st < Q4_NST - 1; ++st)
    if (st < nk) dma(st);
  int st = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = nk - 1 - kt;                    // K-tiles behind this one that have been issued may stay in flight: min(2, ahead)
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");           // every wave's pieces of K-tile kt have landed; stage (kt - 1) % 4 is no longer read
    if (kt + Q4_NST - 1 < nk) dma(st == 0 ? Q4_NST - 1 : st - 1);
    const uint32_t wb = wb0 + (uint32_t)st * Q4_STAGE, xb = xb0 + (uint32_t)st * Q4_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) { Q4_RD(wf[0][i], wb, i * 128); Q4_RD(xf[0][i], xb, i * 128); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { Q4_RD(wf[1][i], wb, i * 128 + 64); Q4_RD(xf[1][i], xb, i * 128 + 64); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][i], xf[0][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][i], xf[1][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    st = st == Q4_NST - 1 ? 0 : st + 1;
  }
#undef Q4_RD
  // accumulator (n-tile i, m-tile j): lane (l15, lc) holds token row m0 + 64 wm + 16 j + l15, columns n0 + 64 wn + 16 i + 4 lc .. + 3
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + 64 * wm + 16 * j + l15;
    if (m >= p.M) continue;
    const float bias_m = (EPI == K5_EPI_BIAS_M && p.bias) ? p.bias[m] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      gemm_epilogue_quad<EPI>(p, v, m, n0 + 64 * wn + 16 * i + 4 * lc, bias_m);
    }
  }
}

// the tail launch shared by the persistent kernels: `rem` logical 256x256 tiles from `full` on, as quadrants
template <int EPI>
int launch_tail(GemmP p, hipStream_t stream, int full, int rem) {
  p.tail_base = full;
  static const bool old_tail = getenv("K5_GEMM_TAIL_V1") != nullptr;   // A/B: the one-K-tile-of-prefetch kernel of rounds 1-3
  // more quadrants than CUs (FF1: 88 tiles = 352 quadrants): the 64-KB kernel runs two of them per CU at once, 38.5 us against 52.7 for two
  // rounds of this one (133 KB); up to one round this one wins (rocprofv3, K = 1792 / 7168: 26-35 / 92 us against 28-38 / 115)
  if (old_tail || (p.K % (2 * BK)) != 0 || p.K < 4 * BK || 4 * rem > 256) {
    p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (p.N + BN - 1) / BN;
    hipLaunchKernelGGL(gemm_bf16_glds_kernel<EPI>, dim3(4 * rem), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
  }
  static const hipError_t attr_rc = hipFuncSetAttribute((const void*)gemm_bf16_q4_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, Q4_LDS);   // set once, thread-safe (function-local static: loopback ranks launch from P host threads)
  if (attr_rc != hipSuccess) return K5_ERR_HIP;
  hipLaunchKernelGGL(gemm_bf16_q4_kernel<EPI>, dim3(4 * rem), dim3(256), Q4_LDS, stream, p);
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}

// a WHOLE small GEMM as quadrants of its 256x256 logical tiles on the deep-prefetch 128x128 kernel (config-1 shapes: 91 tiles = 364 quadrants)
template <int EPI>
int launch_q4_whole(GemmP p, hipStream_t stream) {
  static const hipError_t attr_rc = hipFuncSetAttribute((const void*)gemm_bf16_q4_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, Q4_LDS);   // set once, thread-safe (function-local static: loopback ranks launch from P host threads)
  if (attr_rc != hipSuccess) return K5_ERR_HIP;
  p.tiles256_m = (p.M + 255) / 256; p.tiles256_n = (p.N + 255) / 256; p.tail_base = 0;
  hipLaunchKernelGGL(gemm_bf16_q4_kernel<EPI>, dim3(4 * p.tiles256_m * p.tiles256_n), dim3(256), Q4_LDS, stream, p);
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}

// ---- stream-K workspace: one per (device, stream) — launches on a stream are serialised, so its partial tiles and flags are never shared by two
// kernels in flight.  num_cu x 4 waves x 64 quads x 1 KB of fp32 partial sums (64 MB at 256 CUs) + the flags; allocated on first use (never inside
// a stream capture: the engine's step 0 runs eagerly on the capture stream first), kept for the life of the process.
constexpr size_t SK_WS_PER_CU = 4 * 64 * 1024, SK_FLAGS_PER_CU = 16;
std::mutex g_sk_mu;
std::map<std::pair<int, hipStream_t>, float*> g_sk_ws;
int g_sk_default = -1;                    // k5_gemm_set_stream_k: -1 = K5_GEMM_SK from the environment
thread_local int t_sk_override = -1;      // the engine's per-handle option, set around a forward on the calling thread

int sk_policy() {
  static const int env = getenv("K5_GEMM_SK") ? atoi(getenv("K5_GEMM_SK")) : K5_GEMM_SK_DEFAULT;
  if (t_sk_override >= 0) return t_sk_override;
  return g_sk_default >= 0 ? g_sk_default : env;
}

float* sk_workspace(hipStream_t stream, int num_cu) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_sk_mu);
  auto it = g_sk_ws.find({dev, stream});
  if (it != g_sk_ws.end()) return it->second;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
  void* ptr = nullptr;
  const size_t bytes = (size_t)num_cu * (SK_WS_PER_CU + SK_FLAGS_PER_CU);
  if (hipMalloc(&ptr, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (hipMemset((char*)ptr + (size_t)num_cu * SK_WS_PER_CU, 0, (size_t)num_cu * SK_FLAGS_PER_CU) != hipSuccess) { (void)hipFree(ptr); return nullptr; }
  g_sk_ws[{dev, stream}] = (float*)ptr;
  return (float*)ptr;
}

template <int EPI, int MT>
int launch_w4_mt(GemmP p, hipStream_t stream, int num_cu, bool no_tail, int sk_mode) {
  static const hipError_t attr_rc = hipFuncSetAttribute((const void*)gemm_bf16_w4_kernel<EPI, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, w4_ops_bytes(MT) + W4_TRACE_BYTES + W4_PF_BYTES);   // set once, thread-safe (function-local static: loopback ranks launch from P host threads)
  if (attr_rc != hipSuccess) return K5_ERR_HIP;
#ifdef W4_TRACE
  p.trace = getenv("K5_GEMM_TRACE") ? (unsigned long long*)strtoull(getenv("K5_GEMM_TRACE"), nullptr, 16) : nullptr;
#endif
  p.tiles_m = (p.M + 32 * MT - 1) / (32 * MT); p.tiles_n = (p.N + K8_BN - 1) / K8_BN;
  const int tiles = p.tiles_m * p.tiles_n;
  const int full = tiles / num_cu * num_cu, rem = tiles - full;
  // stream-K (sk_mode 1: where the ragged round leaves at least K5_GEMM_SK_IDLE percent of the CUs idle; 2: wherever it applies): the tiles do not
  // divide over the CUs and there is at least one per CU (a run is then at least a tile long: a tile is cut in at most two); the kernel's flag
  // layout is sized by the GRID, which is all CUs; the XCD ownership it relies on (workgroup b on XCD b & 7) needs whole groups of 8
  static const int sk_idle = getenv("K5_GEMM_SK_IDLE") ? atoi(getenv("K5_GEMM_SK_IDLE")) : 50;
  static const int sk_smax = getenv("K5_GEMM_SK_S") ? atoi(getenv("K5_GEMM_SK_S")) : 2;   // 2 / 4 / 8 slices: within 1 % of each other (the owner adds S - 1 partial tiles one after the other)
  p.sk_ws = nullptr; p.sk_smax = sk_smax < 2 ? 2 : (sk_smax > 16 ? 16 : sk_smax);
  if (sk_mode > 0 && EPI != K5_EPI_F32 && (rem != 0 || sk_mode == 3) && tiles >= num_cu && (num_cu & 7) == 0 && (sk_mode >= 2 || 100 * (num_cu - rem) >= sk_idle * num_cu))
    p.sk_ws = sk_workspace(stream, num_cu);
  if (p.sk_ws) {
    p.lid_limit = tiles;
    p.tiles256_m = p.tiles_m; p.tiles256_n = p.tiles_n;
    hipLaunchKernelGGL((gemm_bf16_w4_kernel<EPI, MT>), dim3(num_cu), dim3(256), w4_ops_bytes(MT) + W4_TRACE_BYTES + W4_PF_BYTES, stream, p);
    return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
  }
  const bool split_tail = MT == 8 && !no_tail && full > 0 && rem > 0 && 2 * rem < num_cu;   // see launch_k8_mt
  p.lid_limit = split_tail ? full : tiles;
  p.tiles256_m = p.tiles_m; p.tiles256_n = p.tiles_n;
  hipLaunchKernelGGL((gemm_bf16_w4_kernel<EPI, MT>), dim3(min(p.lid_limit, num_cu)), dim3(256), w4_ops_bytes(MT) + W4_TRACE_BYTES + W4_PF_BYTES, stream, p);
  if (hipGetLastError() != hipSuccess) return K5_ERR_HIP;
  if (split_tail) return launch_tail<EPI>(p, stream, full, rem);
  return K5_OK;
}

// Which token-tile height: rounds of the CUs x the measured relative cost of a round (tools/gemm_block_shapes.py, profiles/r05_gemm_block_shapes_mt.log:
// N = K = 1792, one round: 40-43 us at 256 rows, 34-37 at 192, 25-27 at 128 — a K-tile is paced by the global -> LDS round trip of a two-stage
// pipeline, not by its MFMA count, so a shorter tile costs more than its share); a ragged last round of 256-row tiles costs nearly a whole one
// even when the quadrant kernel takes it over (82 us for 1 round + 73 tiles).  Ties go to the taller tile.  K5_GEMM_MT=4/6/8 forces one (A/B).
inline int w4_pick_mt(int M, int N, int num_cu, bool no_tail, int force_mt = 0, double* cost_out = nullptr) {
  static const int force_env = getenv("K5_GEMM_MT") ? atoi(getenv("K5_GEMM_MT")) : 0;
  const int force = force_mt ? force_mt : force_env;
  const int tn = (N + K8_BN - 1) / K8_BN;
  const double rel[3] = {1.0, 0.85, 0.63};
  const int mts[3] = {8, 6, 4};
  int best = 8; double best_cost = 1e30;
  for (int v = 0; v < 3; ++v) {
    const int tiles = ((M + 32 * mts[v] - 1) / (32 * mts[v])) * tn;
    const int full = tiles / num_cu, rem = tiles % num_cu;
    double c = rel[v] * (full + (rem == 0 ? 0.0 : (mts[v] == 8 && !no_tail && full > 0 && 2 * rem < num_cu ? 0.85 : 1.0)));
    if (force == mts[v]) c = -1.0;
    if (c < best_cost - 1e-9) { best_cost = c; best = mts[v]; }
  }
  if (cost_out) *cost_out = best_cost;
  return best;
}

template <int EPI>
int launch_w4(GemmP p, hipStream_t stream, int num_cu, bool no_tail, int mt, int sk_mode) {
  switch (mt) {
    case 4: return launch_w4_mt<EPI, 4>(p, stream, num_cu, no_tail, sk_mode);
    case 6: return launch_w4_mt<EPI, 6>(p, stream, num_cu, no_tail, sk_mode);
    default: return launch_w4_mt<EPI, 8>(p, stream, num_cu, no_tail, sk_mode);
  }
}

}  // namespace

// stream-K policy of the four-wave kernel: 0 = whole tiles (the schedule of rounds 2-5, one K order everywhere: bit-identical across kernels and
// tile heights), 1 = stream-K where the ragged round idles enough CUs, 2 = wherever it applies; -1 = back to K5_GEMM_SK / the built-in default
void k5_gemm_set_stream_k_default(int mode) { g_sk_default = mode < 0 ? -1 : (mode > 2 ? 2 : mode); }
void k5_gemm_set_stream_k_thread(int mode) { t_sk_override = mode < 0 ? -1 : (mode > 2 ? 2 : mode); }
int k5_gemm_stream_k_policy() { return sk_policy(); }

// Host launcher (C++ linkage, used by the C-ABI layer in k5_api.hip and by the engine).
int k5_launch_gemm_bf16(const void* A, const void* W, const float* bias, void* C, int M, int N, int K,
                        int lda, int ldw, int ldc, int epi, const void* resid, int ldr,
                        const float* gate, hipStream_t stream, int force_kernel, int force_mt) {
  if (M <= 0 || N <= 0 || K <= 0) return K5_ERR_ARG;
  if (force_mt != 0 && force_mt != 4 && force_mt != 6 && force_mt != 8) return K5_ERR_ARG;
  if ((K & 7) || (lda & 7) || (ldw & 7)) return K5_ERR_ALIGN;  // 16-B aligned rows
  if (epi == K5_EPI_GATE && (!resid || !gate)) return K5_ERR_ARG;
  GemmP p;
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.C = (bf16_t*)C; p.bias = bias;
  p.resid = (const bf16_t*)resid; p.gate = gate;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr;
  p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
  p.alpha = 1.f; p.causal_hw = 0; p.trace = nullptr; p.lid_limit = 0; p.tail_base = -1; p.tiles256_m = p.tiles256_n = 0; p.sk_ws = nullptr; p.sk_smax = 2;
  // force_kernel 14 / 24: the four-wave kernel WITH stream-K (where the ragged round idles enough CUs / wherever it applies); 4: without; 0: the policy
  const int sk_mode = force_kernel == 14 ? 1 : (force_kernel == 24 ? 2 : (force_kernel == 34 ? 3 : (force_kernel != 0 ? 0 : sk_policy())));   // 34: even when the tiles divide (A/B of the walk itself)
  if (force_kernel == 14 || force_kernel == 24 || force_kernel == 34) force_kernel = 4;
  static const int dbg = getenv("K5_GEMM_DBG") ? atoi(getenv("K5_GEMM_DBG")) : 0;
  p.dbg = dbg;
  const dim3 grid(p.tiles_m * p.tiles_n), block(256);
  static const int force_env = getenv("K5_GEMM_V1") ? atoi(getenv("K5_GEMM_V1")) : 0;  // A/B switch for benchmarking
  const int force_v1 = force_kernel ? force_kernel : force_env;   // force_kernel: a caller that needs ONE kernel whatever the shape (same numbering)
  // K5_GEMM_V1=3 selects the 256x128 3-stage counted-vmcnt variant.  Measured (round 1, model shapes): within +-3 % of the
  // 128x128 direct-to-LDS kernel (ff2 844 vs 827, ff1 698 vs 719 TFLOP/s) -> L2-miss latency is not the limiter; not default.
  // default for the model's large projections: the 256x256 two-group ping-pong kernel (K5_GEMM_V1=2 keeps the 128x128 one)
  // ... once its 256x256 tiles fill at least half of the CUs (measured crossover, tools/gemm_small.py: 91-112 tiles lose to the
  // 128x128 kernel by 5-10 %, 42 tiles by 40 %; 168 tiles win by 15 %)
  const long long tiles256 = (long long)((M + 255) / 256) * ((N + 255) / 256);
  // the 4-wave kernel from one full round of 256x256 tiles up (measured on one box, interleaved runs, TFLOP/s 4-wave vs 8-wave:
  // q|k 903 vs 841, V^T 983 vs 943, out+gate 814 vs 708, FF1+GELU 1004 vs 946, FF2+gate 1163 vs 1070, 4096^3 1240 vs 1174;
  // 4-GPU token shards (329 tiles) +6-10 %; 8-GPU shards (168 tiles) lose 3-8 % to the 8-wave kernel's 192-row tile option,
  // which therefore keeps the range below 256 tiles).  K5_GEMM_V1=4 / 8 force one of them.
  const bool w4_ok = (K % (2 * BK)) == 0 && K >= 4 * BK && M >= 512 && N >= 128 && !(N & 7)   // (N = 128: the VAE's 256 -> 128 shortcut at full resolution, 6.7 M rows — half of the
                                                                                                    // 256-wide weight tile multiplies zeros the buffer range check supplies, and it is still twice the 128 x 128 kernel's rate)
                     && !(ldc & 7) && (epi != K5_EPI_GATE || !(ldr & 3));
  // Round 5: with the token-tile height chosen per launch (w4_pick_mt: 256 / 192 / 128 rows) the four-wave kernel also takes the launches below one
  // round of 256 x 256 tiles — 8-GPU token shards (168 such tiles -> 217 of 192 rows) and BASELINE config 1 (91 -> 182 of 128 rows) — from
  // K5_GEMM_W4_MIN (default 96) tiles of the chosen height up (measured, profiles/r05_gemm_block_shapes_*.log; until round 4 those ran on the
  // 8-wave kernel / the 128 x 128 kernel).
  static const int num_cu = k5_num_cu();   // one device per process (one process per GPU); initialised once, thread-safe
  if (num_cu <= 0) return K5_ERR_HIP;
  static const bool no_tail = getenv("K5_GEMM_NO_TAIL") != nullptr;
  static const int w4_min = getenv("K5_GEMM_W4_MIN") ? atoi(getenv("K5_GEMM_W4_MIN")) : 96;
  const int mt_pick = w4_ok ? w4_pick_mt(M, N, num_cu, no_tail, force_mt) : 8;
  const long long tiles_pick = (long long)((M + 32 * mt_pick - 1) / (32 * mt_pick)) * ((N + 255) / 256);
  if (w4_ok && (force_v1 == 4 || (force_v1 == 0 && tiles_pick >= w4_min))) {
    switch (epi) {
      case K5_EPI_BIAS: return launch_w4<K5_EPI_BIAS>(p, stream, num_cu, no_tail, mt_pick, sk_mode);
      case K5_EPI_BIAS_M: return launch_w4<K5_EPI_BIAS_M>(p, stream, num_cu, no_tail, mt_pick, sk_mode);
      case K5_EPI_GELU: return launch_w4<K5_EPI_GELU>(p, stream, num_cu, no_tail, mt_pick, sk_mode);
      case K5_EPI_GATE: return launch_w4<K5_EPI_GATE>(p, stream, num_cu, no_tail, mt_pick, sk_mode);
      default: return K5_ERR_ARG;
    }
  }
  if ((K % BK) == 0 && K >= 2 * BK && M >= 512 && N >= 256 && (force_v1 == 8 || (force_v1 == 0 && tiles256 >= 128))) {
    switch (epi) {
      case K5_EPI_BIAS: return launch_k8<K5_EPI_BIAS>(p, stream);
      case K5_EPI_BIAS_M: return launch_k8<K5_EPI_BIAS_M>(p, stream);
      case K5_EPI_GELU: return launch_k8<K5_EPI_GELU>(p, stream);
      case K5_EPI_GATE: return launch_k8<K5_EPI_GATE>(p, stream);
      default: return K5_ERR_ARG;
    }
  }
  if (w4_ok && force_v1 == 5) {   // A/B: the whole GEMM on the deep-prefetch quadrant kernel
    switch (epi) {
      case K5_EPI_BIAS: return launch_q4_whole<K5_EPI_BIAS>(p, stream);
      case K5_EPI_BIAS_M: return launch_q4_whole<K5_EPI_BIAS_M>(p, stream);
      case K5_EPI_GELU: return launch_q4_whole<K5_EPI_GELU>(p, stream);
      case K5_EPI_GATE: return launch_q4_whole<K5_EPI_GATE>(p, stream);
      default: return K5_ERR_ARG;
    }
  }
  if ((K % BK) == 0 && M >= 512 && force_v1 == 3) {
    switch (epi) {
      case K5_EPI_BIAS: return launch_k3<K5_EPI_BIAS>(p, stream);
      case K5_EPI_BIAS_M: return launch_k3<K5_EPI_BIAS_M>(p, stream);
      case K5_EPI_GELU: return launch_k3<K5_EPI_GELU>(p, stream);
      case K5_EPI_GATE: return launch_k3<K5_EPI_GATE>(p, stream);
      default: return K5_ERR_ARG;
    }
  }
  if ((K % BK) == 0 && force_v1 != 1) {
    switch (epi) {
      case K5_EPI_BIAS: hipLaunchKernelGGL(gemm_bf16_glds_kernel<K5_EPI_BIAS>, grid, block, 0, stream, p); break;
      case K5_EPI_BIAS_M: hipLaunchKernelGGL(gemm_bf16_glds_kernel<K5_EPI_BIAS_M>, grid, block, 0, stream, p); break;
      case K5_EPI_GELU: hipLaunchKernelGGL(gemm_bf16_glds_kernel<K5_EPI_GELU>, grid, block, 0, stream, p); break;
      case K5_EPI_GATE: hipLaunchKernelGGL(gemm_bf16_glds_kernel<K5_EPI_GATE>, grid, block, 0, stream, p); break;
      default: return K5_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
  }
  switch (epi) {
    case K5_EPI_BIAS: hipLaunchKernelGGL(gemm_bf16_kernel<K5_EPI_BIAS>, grid, block, 0, stream, p); break;
    case K5_EPI_BIAS_M: hipLaunchKernelGGL(gemm_bf16_kernel<K5_EPI_BIAS_M>, grid, block, 0, stream, p); break;
    case K5_EPI_GELU: hipLaunchKernelGGL(gemm_bf16_kernel<K5_EPI_GELU>, grid, block, 0, stream, p); break;
    case K5_EPI_GATE: hipLaunchKernelGGL(gemm_bf16_kernel<K5_EPI_GATE>, grid, block, 0, stream, p); break;
    default: return K5_ERR_ARG;
  }
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}

// C_f32[M][N] = alpha * A[M][K] . W[N][K]^T   (fp32 output; attention scores of the VAE mid block)
// causal_hw > 0: frame-causal scores — row i is only read at columns < (i / causal_hw + 1) * causal_hw (k5_launch_causal_softmax
// with the same hw), so output tiles wholly beyond that limit are neither computed nor written (15 of 25 frame pairs at T = 5).
int k5_launch_gemm_bf16_f32out(const void* A, const void* W, float* C, int M, int N, int K, int lda, int ldw, int ldc,
                               float alpha, int causal_hw, hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || causal_hw < 0) return K5_ERR_ARG;
  if ((K & 7) || (lda & 7) || (ldw & 7)) return K5_ERR_ALIGN;
  GemmP p;
  p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.C = (bf16_t*)C; p.bias = nullptr; p.resid = nullptr; p.gate = nullptr;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = 0; p.alpha = alpha; p.dbg = 0; p.trace = nullptr; p.lid_limit = 0; p.tail_base = -1; p.tiles256_m = p.tiles256_n = 0; p.sk_ws = nullptr; p.sk_smax = 2;
  p.causal_hw = causal_hw;
  // the 4-wave persistent kernel from one round of kept 256x256 tiles up (same conditions as k5_launch_gemm_bf16)
  const int tm = (M + K8_BM - 1) / K8_BM, tn = (N + K8_BN - 1) / K8_BN;
  long long kept = 0;
  for (int i = 0; i < tm; ++i) {
    const int last = std::min((i + 1) * K8_BM, M) - 1;
    kept += causal_hw > 0 ? std::min(tn, (int)((std::min((long long)N, ((long long)last / causal_hw + 1) * causal_hw) + K8_BN - 1) / K8_BN)) : tn;
  }
  static const int force_v1 = getenv("K5_GEMM_V1") ? atoi(getenv("K5_GEMM_V1")) : 0;
  if ((K % (2 * BK)) == 0 && K >= 4 * BK && M >= 512 && N >= 256 && !(N & 3) && !(ldc & 3) && kept >= 256 && kept < (1ll << 30) &&
      (force_v1 == 0 || force_v1 == 4)) {
    static const hipError_t attr_rc = hipFuncSetAttribute((const void*)gemm_bf16_w4_kernel<K5_EPI_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);   // set once, thread-safe (function-local static: loopback ranks launch from P host threads)
    if (attr_rc != hipSuccess) return K5_ERR_HIP;
    static const int num_cu = k5_num_cu();   // one device per process (one process per GPU); initialised once, thread-safe
    if (num_cu <= 0) return K5_ERR_HIP;
    p.tiles_m = tm; p.tiles_n = tn; p.lid_limit = (int)kept; p.tiles256_m = tm; p.tiles256_n = tn;
    hipLaunchKernelGGL((gemm_bf16_w4_kernel<K5_EPI_F32>), dim3(std::min((int)kept, num_cu)), dim3(256), W4_LDS, stream, p);
    return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
  }
  p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
  const dim3 grid(p.tiles_m * p.tiles_n), block(256);
  if ((K % BK) == 0) hipLaunchKernelGGL(gemm_bf16_glds_kernel<K5_EPI_F32>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(gemm_bf16_kernel<K5_EPI_F32>, grid, block, 0, stream, p);
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}
