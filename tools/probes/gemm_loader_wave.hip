// Can the global -> LDS stream be taken OFF the MFMA wave?  (round 5; HISTORY.md §R5 "what paces a K-tile")
//
// The production four-wave GEMM (csrc/gemm_bf16.hip) runs one wave per SIMD and that wave issues its own `buffer_load ... lds` pieces between its
// MFMAs; a piece costs its wave ~60 cycles of issue (MI355X_MICROARCH.md) and a 16x16x32 MFMA holds the pipe for 16, so the pipe idles behind every
// piece.  This probe runs the 128-row form's tile — 256 weight rows x 128 token rows per workgroup, 64-wide K-tiles, the production LDS image
// (1040-B pieces, conflict-free fragment reads), three stages — in two ways:
//   NL = 0   the four MFMA waves issue the DMAs themselves (spread over their MFMAs), as the production kernel does
//   NL = 4   four MORE waves (one per SIMD next to the MFMA wave) do nothing but the DMAs; the MFMA waves only read fragments and multiply.  A
//            kernel's waves share one register allocation, so this needs the MFMA wave inside 256 registers: 128 accumulators (the 128-row form)
//            + two single-k-step fragment sets (96) + addresses; the 256-row form (256 accumulators) cannot do this.
//   NL = 2   two loader waves (24 pieces each)
// One barrier per K-tile (at the k-step 0 -> 1 boundary): before it the loaders have waited for K-tile t+1 and the MFMA waves hold K-tile t's last
// fragments; after it the loaders refill stage t % 3 with K-tile t+3 and the MFMA waves read K-tile t+1's first fragments under k-step 1's MFMAs.
// Output: plain bf16 store of C = X . W^T (checked against a host reference on sampled entries), us per launch, us per K-tile, TFLOP/s.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gemm_loader_wave tools/probes/gemm_loader_wave.hip && /tmp/gemm_loader_wave [M N K]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t bf16_t;
typedef __attribute__((address_space(3))) void lds_t;

constexpr int PAD = 1040, OPW = 32 * PAD, OPX = 16 * PAD, STG = OPW + OPX, NST = 3, LDS_BYTES = NST * STG;
constexpr int BK = 64;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct P { const bf16_t* W; const bf16_t* X; bf16_t* C; int M, N, K; };

#define RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define MF(WF, XF, Q) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[(Q) & 7][(Q) >> 3]) : "v"(WF[(Q) & 7]), "v"(XF[(Q) >> 3]))
#define MF0(WF, XF, Q) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[(Q) & 7][(Q) >> 3]) : "v"(WF[(Q) & 7]), "v"(XF[(Q) >> 3]))

template <int NL>
__global__ __launch_bounds__(256 + 64 * NL) __attribute__((amdgpu_waves_per_eu(NL ? 2 : 1, NL ? 2 : 1))) void gemm_probe(P p) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = p.N / 256;
  const int m0 = (blockIdx.x / tiles_n) * 128, n0 = (blockIdx.x % tiles_n) * 256;
  const int nk = p.K / BK;
  const uint32_t ld2 = (uint32_t)p.K * 2u;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_t*)dsm;

  // ---- DMA side: piece d of an operand = 8 row-tiles x 128 B; piece index 16 h + r holds rows 128 h + 16 i + r.  W: 32 pieces, X: 16.
  // 48 pieces per K-tile are dealt round-robin to the ND issuing waves (loader waves, or the MFMA waves themselves when NL = 0).
  constexpr int NDW = NL ? NL : 4;               // waves that issue DMAs
  constexpr int PER = 48 / NDW;                  // pieces per issuing wave and K-tile
  const int dw = NL ? wave - 4 : wave;           // index among the issuing waves
  const uint32_t v0 = (uint32_t)(16 * (lane >> 3)) * ld2 + (uint32_t)(lane & 7) * 16u;
  __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.W) + (size_t)n0 * ld2), 0, (int)(256u * ld2), 0x00020000);
  __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.X) + (size_t)m0 * ld2), 0, (int)(128u * ld2), 0x00020000);
  uint32_t vk = v0;                              // lane offset incl. the K advance of the DMA cursor
  // issuing wave dw takes W pieces PW dw .. PW dw + PW - 1 (j < PW) and X pieces PX dw .. (j >= PW): which operand is a compile-time property of j, so
  // no branch sits between the MFMAs of the NL = 0 form (a branch around the asm MFMA stream makes the allocator park the accumulators)
  constexpr int PW = 32 / NDW, PX = 16 / NDW;
  auto dma_mine = [&](int so, int j) {           // my j-th piece (j = 0 .. PER-1) of the cursor's K-tile into the stage at byte offset so
    if (j < PW) {
      const int q = PW * dw + j, h = q >> 4, r = q & 15;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_t*)(dsm + so + q * PAD), 16, vk, (uint32_t)(128 * h + r) * ld2, 0, 0);
    } else {
      const int r = PX * dw + (j - PW);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (lds_t*)(dsm + so + OPW + r * PAD), 16, vk, (uint32_t)r * ld2, 0, 0);
    }
  };

  if (NL && wave >= 4) {
    // ================= loader wave =================
#pragma unroll
    for (int st = 0; st < NST; ++st) {
#pragma unroll
      for (int j = 0; j < PER; ++j) dma_mine(st * STG, j);
      vk += 2 * BK;
    }
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PER) : "memory");   // K-tile 0
    asm volatile("s_barrier" ::: "memory");                            // P0
    int so = 0;
    for (int t = 0; t < nk; ++t) {
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");      // K-tile t+1 has landed (only K-tile t+2's pieces may be in flight)
      asm volatile("s_barrier" ::: "memory");                          // M_t: stage t % 3 is free
#pragma unroll
      for (int j = 0; j < PER; ++j) dma_mine(so, j);                   // K-tile t+3 (past the end: beyond the buffer range -> zeros, nothing reads them)
      vk += 2 * BK;
      so = so == 2 * STG ? 0 : so + STG;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ================= MFMA wave =================
  const int wn = wave & 1, wm = wave >> 1;
  const int l15 = lane & 15, lc = lane >> 4;
  const uint32_t wb0 = lds0 + (16 * wn + l15) * PAD + lc * 16;           // + 128 i + 64 s (+ stage)
  const uint32_t xb0 = lds0 + OPW + l15 * PAD + lc * 16 + wm * 512;      // + 128 j + 64 s (+ stage)
  bf16x8 wf[2][8], xf[2][4];
  f32x4 acc[8][4];
  if (NL == 0) {
#pragma unroll
    for (int st = 0; st < NST; ++st) {
#pragma unroll
      for (int j = 0; j < PER; ++j) dma_mine(st * STG, j);
      vk += 2 * BK;
    }
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PER) : "memory");
  }
  asm volatile("s_barrier" ::: "memory");                                // P0: K-tile 0 is in stage 0
#pragma unroll
  for (int i = 0; i < 8; ++i) RD(wf[0][i], wb0, i * 128);
#pragma unroll
  for (int j = 0; j < 4; ++j) RD(xf[0][j], xb0, j * 128);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  uint32_t wb = wb0, xb = xb0;     // fragment bases of the CURRENT K-tile's stage
  int so = 0;                      // (NL = 0) the stage the current K-tile sits in = the one refilled after M_t
  auto ktile = [&](auto FIRSTC) {
    constexpr bool first = decltype(FIRSTC)::value;
    const uint32_t wbn = so == 2 * STG ? wb - 2 * STG : wb + STG, xbn = so == 2 * STG ? xb - 2 * STG : xb + STG;   // next K-tile's stage
    // k-step 0 on set 0; k-step-1 fragments of this K-tile into set 1, a read after every second MFMA
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      if (first) MF0(wf[0], xf[0], m); else MF(wf[0], xf[0], m);
      if ((m & 1) == 0 && m / 2 < 12) {
        const int r = m / 2;
        if (r < 8) RD(wf[1][r], wb, r * 128 + 64); else RD(xf[1][r - 8], xb, (r - 8) * 128 + 64);
      }
    }
    if (NL == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(PER) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");                              // M_t
    // k-step 1 on set 1; k-step-0 fragments of the NEXT K-tile into set 0; (NL = 0) this wave's pieces of K-tile t+3 into the freed stage
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      MF(wf[1], xf[1], m);
      if ((m & 1) == 0 && m / 2 < 12) {
        const int r = m / 2;
        if (r < 8) RD(wf[0][r], wbn, r * 128); else RD(xf[0][r - 8], xbn, (r - 8) * 128);
      }
      if (NL == 0 && (m & 1) == 1 && m / 2 < PER) dma_mine(so, m / 2);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (NL == 0) vk += 2 * BK;
    wb = wbn; xb = xbn;
    so = so == 2 * STG ? 0 : so + STG;
  };
  ktile(std::true_type{});
  for (int t = 1; t < nk; ++t) ktile(std::false_type{});
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  // plain bf16 store: lane holds columns n0 + 128 wn + 16 i + 4 lc .. + 3 of token m0 + 64 wm + 16 j + l15
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + 64 * wm + 16 * j + l15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = n0 + 128 * wn + 16 * i + 4 * lc;
      const f32x4 v = acc[i][j];
      uint32_t lo, hi;
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(v[0]), "v"(v[1]));
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v[2]), "v"(v[3]));
      if (m < p.M) *reinterpret_cast<uint2*>(p.C + (size_t)m * p.N + n) = uint2{lo, hi};
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ---------------------------------------------------------------------------------------------
// XV: the token operand does not go through the LDS at all.  Whoever issues them, the LDS-DMA pieces of this tile move at 53-61 GB/s per CU (NL = 0 / 2 / 4
// above; the production kernels; the same with 182 or 256 CUs streaming) — the path, not the issue slot, is the limit.  Here only the WEIGHT tile (32 of
// the 48 KB per K-tile) comes by `buffer_load ... lds`; every MFMA wave loads its own 64 x 64 token fragments straight into registers
// (`global_load_dwordx4`: lane = token row l15, 16-B k-chunk lc — exactly the B-operand layout), two K-tiles deep, refilled in place after their last
// use.  The token rows are then fetched twice per workgroup (both wn waves; the second hits the L1), but on the other memory path.
//   per wave and K-tile: 8 W pieces (DMA) + 8 token fragments (VGPR) + 16 W fragment reads (LDS) for 64 MFMAs
// Queue order per K-tile t: A_t (k-step-1 token fragments of K-tile t+1, 4 loads, under k-step 0), then after the barrier B_t (k-step-0 token fragments of
// K-tile t+2, 4 loads) and C_t (W pieces of K-tile t+3, 8 DMAs): vmcnt(16) at the barrier = everything up to A_{t-1} and C_{t-2}; vmcnt(24) at the end of
// the tile = everything up to B_{t-1}.
constexpr int XV_STG = OPW, XV_LDS = NST * XV_STG;
#define GLD(DST, VOFF, SBASE, OFF) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(DST) : "v"(VOFF), "s"(SBASE), "n"(OFF))
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_probe_xv(P p) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = p.N / 256;
  const int m0 = (blockIdx.x / tiles_n) * 128, n0 = (blockIdx.x % tiles_n) * 256;
  const int nk = p.K / BK;
  const uint32_t ld2 = (uint32_t)p.K * 2u;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_t*)dsm;
  const int wn = wave & 1, wm = wave >> 1;
  const int l15 = lane & 15, lc = lane >> 4;
  const uint32_t v0 = (uint32_t)(16 * (lane >> 3)) * ld2 + (uint32_t)(lane & 7) * 16u;
  __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.W) + (size_t)n0 * ld2), 0, (int)(256u * ld2), 0x00020000);
  uint32_t vk = v0;                              // W DMA cursor (lane offset incl. K advance)
  auto dma_w = [&](int so, int j) {              // my j-th W piece (pieces 8 wave .. 8 wave + 7) of the cursor's K-tile
    const int q = 8 * wave + j, h = q >> 4, r = q & 15;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_t*)(dsm + so + q * PAD), 16, vk, (uint32_t)(128 * h + r) * ld2, 0, 0);
  };
  // token fragments: row m0 + 64 wm + 16 j + l15, bytes 128 kt + 64 s + 16 lc .. + 15
  const char* xb_[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) xb_[j] = reinterpret_cast<const char*>(p.X) + (size_t)(m0 + 64 * wm + 16 * j) * ld2;
  uint32_t xo = (uint32_t)l15 * ld2 + (uint32_t)lc * 16u;     // lane offset; the loads below add the K-tile they fetch
  bf16x8 wf[2][8], xv[2][2][4];
  f32x4 acc[8][4];
  const uint32_t wb0 = lds0 + (16 * wn + l15) * PAD + lc * 16;

  // prologue: W K-tiles 0, 1, 2; token fragments of K-tiles 0 (both k-steps) and 1 (k-step 0)
#pragma unroll
  for (int st = 0; st < NST; ++st) {
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_w(st * XV_STG, j);
    vk += 2 * BK;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { GLD(xv[0][0][j], xo, xb_[j], 0); GLD(xv[0][1][j], xo, xb_[j], 64); }
#pragma unroll
  for (int j = 0; j < 4; ++j) GLD(xv[1][0][j], xo, xb_[j], 128);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i) RD(wf[0][i], wb0, i * 128);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  uint32_t wb = wb0;
  int so = 0;
  // xo_t: lane offset of the CURRENT K-tile t
  auto ktile = [&](auto CC, auto FIRSTC) {
    constexpr int c = decltype(CC)::value;
    constexpr bool first = decltype(FIRSTC)::value;
    const uint32_t wbn = so == 2 * XV_STG ? wb - 2 * XV_STG : wb + XV_STG;
    // k-step 0: MFMAs on wf[0] / xv[c][0]; W k-step-1 fragments into wf[1]; A_t: token fragments (k-step 1) of K-tile t+1 into xv[c^1][1]
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      if (first) MF0(wf[0], xv[c][0], m); else MF(wf[0], xv[c][0], m);
      if ((m & 1) == 0 && m / 2 < 8) RD(wf[1][m / 2], wb, (m / 2) * 128 + 64);
      if (m >= 16 && (m & 3) == 1 && (m - 17) / 4 < 4) GLD(xv[c ^ 1][1][(m - 17) / 4], xo, xb_[(m - 17) / 4], 128 + 64);
    }
    asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    // k-step 1: MFMAs on wf[1] / xv[c][1]; W k-step-0 fragments of K-tile t+1 into wf[0]; B_t: token fragments (k-step 0) of K-tile t+2 into xv[c][0];
    // C_t: W pieces of K-tile t+3 into the freed stage
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      MF(wf[1], xv[c][1], m);
      if ((m & 1) == 0 && m / 2 < 8) RD(wf[0][m / 2], wbn, (m / 2) * 128);
      if (m < 8 && (m & 1) == 1) GLD(xv[c][0][m / 2], xo, xb_[m / 2], 256);      // B_t first (m = 1, 3, 5, 7) ...
      if (m >= 8 && (m - 8) % 3 == 0) dma_w(so, (m - 8) / 3);                     // ... then C_t (m = 8, 11, .., 29): the queue order the waits assume
    }
    asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
    vk += 2 * BK; xo += 2 * BK;
    wb = wbn;
    so = so == 2 * XV_STG ? 0 : so + XV_STG;
  };
  ktile(std::integral_constant<int, 0>{}, std::true_type{});
  ktile(std::integral_constant<int, 1>{}, std::false_type{});
  for (int t = 2; t < nk; t += 2) {
    ktile(std::integral_constant<int, 0>{}, std::false_type{});
    ktile(std::integral_constant<int, 1>{}, std::false_type{});
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + 64 * wm + 16 * j + l15;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = n0 + 128 * wn + 16 * i + 4 * lc;
      const f32x4 v = acc[i][j];
      uint32_t lo, hi;
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(v[0]), "v"(v[1]));
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v[2]), "v"(v[3]));
      if (m < p.M) *reinterpret_cast<uint2*>(p.C + (size_t)m * p.N + n) = uint2{lo, hi};
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static inline bf16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
static inline float bf2f(bf16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

// timing loop and check around a launch statement (a macro, not a function: a __global__ function named inside a lambda / through a function pointer
// does not resolve to the kernel handle hipFuncSetAttribute / hipLaunchKernelGGL want)
void report(const char* name, int grid, int block, double us, const P& p, const std::vector<bf16_t>& hW, const std::vector<bf16_t>& hX) {
  const double flop = 2.0 * p.M * p.N * p.K;
  std::vector<bf16_t> hC((size_t)p.M * p.N);
  CHECK(hipMemcpy(hC.data(), p.C, hC.size() * 2, hipMemcpyDeviceToHost));
  double worst = 0; int bad = 0;
  uint64_t s = 12345;
  for (int c = 0; c < 2000; ++c) {
    s = s * 6364136223846793005ull + 1442695040888963407ull; const int m = (int)((s >> 33) % p.M);
    s = s * 6364136223846793005ull + 1442695040888963407ull; const int n = (int)((s >> 33) % p.N);
    double ref = 0;
    for (int k = 0; k < p.K; ++k) ref += (double)bf2f(hX[(size_t)m * p.K + k]) * bf2f(hW[(size_t)n * p.K + k]);
    const double got = bf2f(hC[(size_t)m * p.N + n]), err = fabs(got - ref);
    if (err > 0.02 + 0.01 * fabs(ref)) ++bad;
    if (err > worst) worst = err;
  }
  printf("%-5s %5d workgroups x %3d threads  %8.1f us per launch  %6.3f us per K-tile  %7.1f TFLOP/s  check: %d of 2000 off (worst abs err %.4f)\n",
         name, grid, block, us, us / (p.K / BK), flop / us / 1e6, bad, worst);
}
#define RUN(KERN, NAME, BLOCK, LDS) do { \
    const int grid = (p.M / 128) * (p.N / 256); \
    CHECK(hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); \
    CHECK(hipMemset(p.C, 0, (size_t)p.M * p.N * 2)); \
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(KERN, dim3(grid), dim3(BLOCK), LDS, 0, p); \
    CHECK(hipDeviceSynchronize()); \
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); \
    CHECK(hipEventRecord(e0)); \
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(KERN, dim3(grid), dim3(BLOCK), LDS, 0, p); \
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); \
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); \
    report(NAME, grid, BLOCK, ms * 1e3 / iters, p, hW, hX); } while (0)

int main(int argc, char** argv) {
  P p; p.M = argc > 1 ? atoi(argv[1]) : 3328; p.N = argc > 2 ? atoi(argv[2]) : 1792; p.K = argc > 3 ? atoi(argv[3]) : 7168;
  const int iters = argc > 4 ? atoi(argv[4]) : 30;
  if (p.M % 128 || p.N % 256 || p.K % 64 || p.K < 256) { printf("M %% 128, N %% 256, K %% 64\n"); return 1; }
  std::vector<bf16_t> hW((size_t)p.N * p.K), hX((size_t)p.M * p.K);
  uint64_t s = 99;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((int)((s >> 40) & 0xffff) - 32768) / 32768.f; };
  for (auto& v : hW) v = f2bf(rnd() * 0.1f);
  for (auto& v : hX) v = f2bf(rnd());
  bf16_t *dW, *dX, *dC;
  CHECK(hipMalloc(&dW, hW.size() * 2)); CHECK(hipMalloc(&dX, hX.size() * 2 + 4096));   // (XV prefetches two K-tiles past the last row's end)
  CHECK(hipMalloc(&dC, (size_t)p.M * p.N * 2));
  CHECK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dX, hX.data(), hX.size() * 2, hipMemcpyHostToDevice));
  p.W = dW; p.X = dX; p.C = dC;
  printf("C[%d][%d] = X[%d][%d] . W[%d][%d]^T, 128 x 256 tiles, three 49 920-B stages\n", p.M, p.N, p.M, p.K, p.N, p.K);
  for (int rep = 0; rep < 2; ++rep) {
    RUN(gemm_probe<0>, "NL=0", 256, LDS_BYTES);
    RUN(gemm_probe<4>, "NL=4", 512, LDS_BYTES);
    RUN(gemm_probe<2>, "NL=2", 384, LDS_BYTES);
    RUN(gemm_probe_xv, "XV", 256, XV_LDS);
  }
  return 0;
}
