// conv3d_w4.hip — the causal 3x3x3 convolution of conv3d.hip on the 4-wave 256-row structure of gemm_bf16.hip
// (gemm_bf16_w4_kernel): one persistent workgroup per CU, one wave per SIMD, accumulators pinned to the AGPR file, operand
// tiles by `buffer_load_dwordx4 ... lds`, MFMA 16x16x32 and ds_read_b128 as a hand-ordered asm stream.
//
// Replaces the same reference code as conv3d.hip (HunyuanVideoCausalConv3d, kandinsky/models/vae.py:125-163, with the nearest
// upsample of HunyuanVideoUpsampleCausal3D vae.py:187-205 folded into the gather) for the layers that fill the machine:
// Cin a multiple of 128, Cout = 128 (256 x 128 tile) or a multiple of 256 (256 x 256 tile), at least one round of tiles.
//
// Implicit GEMM: M = output positions, N = Cout, K = 27 taps x Cin (tap-major, as the packed weights [Cout][27][Cin]).
// A K-tile is (tap, 64-channel slab).  The weight operand is a plain K-contiguous matrix (as in the GEMM).  The activation
// operand is a GATHER: LDS piece d holds the 8 output positions {16 i + (d & 15)} of a 128-row half, and every lane of a DMA
// instruction carries its own byte offset  pos(row, tap) * Cin * 2 + 16 * (lane & 7)  — recomputed per tap (replicate
// padding, causal in T, nearest upsample map: clamps, so there is no zero fill), while the 64-channel slab rides in the
// instruction's SGPR offset.  One lane serves 8 CONSECUTIVE output positions (one per DMA instruction), whose packed
// (t, h, w) coordinates are derived once per output tile.
#include <type_traits>

#include <stdlib.h>

#include "k5_common.h"
#include "k5_kernels.h"

namespace {

constexpr int CBK = 64;
constexpr int CW_PAD = 1040;   // 8 rows x 128 B + 16 B: see gemm_bf16.hip (conflict-free ds_read_b128 at immediate offsets)
typedef __attribute__((address_space(3))) void cw_lds_t;

struct ConvW4P {
  const bf16_t* X; const bf16_t* W; bf16_t* C;
  const float* bias; const bf16_t* resid;
  int Ts, Hs, Ws, To, Ho, Wo, up_t, up_s;
  int Cin, Cout, M, ldc, ldr, tiles_m, tiles_n;
  unsigned x_bytes;
  int hb, wb;          // halo kernel: 8 x 32 output patches per frame (Ho / 8, Wo / 32)
  float* quad_stats;   // STATS: [2 tiles_m][Cout / 4][2] fp32 = (sum, sum of squares) of the STORED bf16 outputs over the 128 rows of
                       // half an m-tile, per 4 consecutive channels — the GroupNorm that consumes this tensor sums them per group
};

// Epilogue of one 256 x BN output tile held in the accumulators (vae.py:274): bf16(acc + bias), or bf16(bf16(acc + bias) + residual);
// loads before stores, 16-byte stores.  rowm(r) = the output row (position index) of tile row r, >= p.M for a row past the end;
// mtile = the tile's index along M (STATS: its two 128-row halves are statistics blocks 2 mtile, 2 mtile + 1).
template <int NTW, bool RESID, bool STATS, class RowM>
__device__ __forceinline__ void conv_w4_epilogue(const ConvW4P& p, f32x4 (&acc)[NTW][8], int n0, int mtile, RowM rowm) {
  int tid2 = threadIdx.x;
  asm volatile("" : "+v"(tid2));
  const int e_l15 = tid2 & 15, e_lc = (tid2 >> 4) & 3, e_wave = tid2 >> 6;
  const int e_wn = e_wave & 1, e_wm = e_wave >> 1;
  // epilogue (vae.py:274): bf16(acc + bias), or bf16(bf16(acc + bias) + residual); loads before stores, 16-byte stores
  const int nb = n0 + 16 * NTW * e_wn + 4 * e_lc;          // + 16 i
  f32x4 bvec[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) bvec[i] = *reinterpret_cast<const f32x4*>(p.bias + nb + 16 * i);
  float gs[NTW], gq[NTW];   // STATS: this lane's sums over its 8 rows, per n-tile (its 4 channels of that tile)
#pragma unroll
  for (int i = 0; i < NTW; ++i) { gs[i] = 0.f; gq[i] = 0.f; }
#pragma unroll
  for (int jh = 0; jh < 4; ++jh) {
    u32x2 rr[2][NTW];
    if (RESID) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int m = min(rowm(128 * e_wm + 16 * (2 * jh + jj) + e_l15), p.M - 1);
#pragma unroll
        for (int i = 0; i < NTW; ++i) rr[jj][i] = *reinterpret_cast<const u32x2*>(p.resid + (size_t)m * p.ldr + nb + 16 * i);
      }
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * jh + jj;
      const int m = rowm(128 * e_wm + 16 * j + e_l15);
#pragma unroll
      for (int iq = 0; iq < NTW / 2; ++iq) {
        u32x2 o[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int i = 2 * iq + h;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + bvec[i][e];
          if (RESID) {
            v[0] = __uint_as_float(rr[jj][i][0] << 16) + bf_round(v[0]);
            v[1] = __uint_as_float(rr[jj][i][0] & 0xffff0000u) + bf_round(v[1]);
            v[2] = __uint_as_float(rr[jj][i][1] << 16) + bf_round(v[2]);
            v[3] = __uint_as_float(rr[jj][i][1] & 0xffff0000u) + bf_round(v[3]);
          }
          o[h] = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          if (STATS && m < p.M) {        // of the values as stored (bf16), which is what the next GroupNorm reads
            const float r0 = __uint_as_float(o[h][0] << 16), r1 = __uint_as_float(o[h][0] & 0xffff0000u);
            const float r2 = __uint_as_float(o[h][1] << 16), r3 = __uint_as_float(o[h][1] & 0xffff0000u);
            gs[i] += (r0 + r1) + (r2 + r3);
            gq[i] = fmaf(r3, r3, fmaf(r2, r2, fmaf(r1, r1, fmaf(r0, r0, gq[i]))));
          }
        }
#pragma unroll
        for (int d = 0; d < 2; ++d) {   // lanes l / l + 16 trade halves: each lane owns 8 consecutive channels of one tile
          const auto sw = __builtin_amdgcn_permlane16_swap(o[0][d], o[1][d], false, false);
          o[0][d] = sw[0]; o[1][d] = sw[1];
        }
        const int n = n0 + 16 * NTW * e_wn + 16 * (2 * iq + (e_lc & 1)) + 8 * (e_lc >> 1);
        if (m < p.M) *reinterpret_cast<u32x4*>(p.C + (size_t)m * p.ldc + n) = u32x4{o[0][0], o[0][1], o[1][0], o[1][1]};
      }
    }
    asm volatile("" ::: "memory");
  }
  if (STATS) {
    // the 16 lanes of a DPP row are the 16 rows of a token tile for one channel quad: four row-local steps leave the row sum
    // in every lane; lanes 0 / 16 / 32 / 48 store their quad's pair.  Fixed order: deterministic.
    auto row_sum = [](float x) __attribute__((always_inline)) {
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xb1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4e, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xf, 0xf, true));   // row_half_mirror
      x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xf, 0xf, true));   // row_mirror
      return x;
    };
    float* qs = p.quad_stats + ((size_t)(2 * mtile + e_wm) * (p.Cout >> 2) + ((n0 + 16 * NTW * e_wn) >> 2) + e_lc) * 2;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      const float a = row_sum(gs[i]), b = row_sum(gq[i]);
      if (e_l15 == 0) *reinterpret_cast<f32x2*>(qs + 8 * i) = f32x2{a, b};
    }
  }
}

// NTW = 16-channel n-tiles per wave: 8 -> 256 x 256 tile (waves 2 x 2, each 128 x 128), 4 -> 256 x 128 tile (each 128 x 64)
template <int NTW, bool RESID, bool STATS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3d_w4_kernel(ConvW4P p) {
  constexpr int BN = 32 * NTW;
  constexpr int WP = BN / 8;                                   // weight pieces per K-tile (8 rows each)
  constexpr int W_OP = WP * CW_PAD, X_OP = 32 * CW_PAD, STAGE = W_OP + X_OP;
  constexpr int HALF = 8 * NTW, TOT = 16 * NTW;                // MFMAs per k-step / per K-tile and wave
  constexpr int NDW = WP / 4, ND = NDW + 8;                    // DMA instructions per wave and K-tile: weights, then positions
  constexpr int NR = NTW + 8;                                  // fragment reads per k-step
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1;
  const int l15 = lane & 15, lc = lane >> 4;

  // persistent walk: XCD x owns a contiguous range of logical tiles (n fastest), its workgroups walk it interleaved
  const int nblk = p.tiles_m * p.tiles_n;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = (gridDim.x + 7 - xcd) >> 3;
  const int q8 = nblk >> 3, r8 = nblk & 7;
  const int x_first = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, x_cnt = q8 + (xcd < r8 ? 1 : 0);
  auto tile_origin = [&](int lid, int& m0, int& n0) __attribute__((always_inline)) { m0 = (lid / p.tiles_n) * 256; n0 = (lid % p.tiles_n) * BN; };
  if (slot >= x_cnt) return;

  const int cpt = p.Cin / CBK;                 // K-tiles per tap (even)
  const int nk = 27 * cpt;                     // even
  const uint32_t ldw2 = (uint32_t)(27 * p.Cin) * 2u, cin2 = (uint32_t)p.Cin * 2u;
  const uint32_t vw0 = (uint32_t)(16 * (lane >> 3)) * ldw2 + (uint32_t)(lane & 7) * 16u;
  const int wrow0 = NTW == 8 ? 128 * (wave >> 1) + 8 * (wave & 1) : NDW * wave;   // first weight row / LDS piece of this wave's DMAs
  const int wslot0 = NDW * wave;
  const int xslot0 = 8 * wave;                                                    // position pieces (h = wave >> 1, r = 8 (wave & 1) + jj)
  // chunk swizzle against the ds_read_b128 lane groups: the pieces of rows r = lane&15 in 4..11 hold
  // their 16-B chunks pairwise swapped.  Measured 0.5 conflict cycles per LDS cycle without it; the 256 x 128 tile reads a
  // fragment in three of every four MFMA slots, so here the LDS matters.
  const uint32_t sw_lo = (wave & 1) ? 16u : 0u, sw_hi = 16u - sw_lo;            // position pieces r = 8 (wave & 1) + jj
  const uint32_t xchunk_lo = ((uint32_t)(lane & 7) * 16u) ^ sw_lo, xchunk_hi = ((uint32_t)(lane & 7) * 16u) ^ sw_hi;
  const uint32_t ww_lo = NTW == 8 ? sw_lo : ((wave == 1 || wave == 2) ? 16u : 0u);   // weight pieces: r = 8 (wave & 1) + jj, or 4 wave + jj
  const uint32_t ww_hi = NTW == 8 ? sw_hi : ww_lo;
  uint32_t vw = vw0;
  uint32_t vxo[8];                             // per DMA instruction: byte offset of this lane's position at the current tap
  uint32_t pk[8];                              // its packed output coordinates  t << 24 | h << 12 | w
  __amdgpu_buffer_rsrc_t rW;
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (int)p.x_bytes, 0x00020000);
  int d_ti = slot, d_cnt = 0, d_tap = 0, d_cc = 0;   // DMA cursor: tile, K-tiles issued, tap, channel slab
  auto set_tap = [&](int tap) __attribute__((always_inline)) {
    const int dt = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      int tu = max((int)(pk[jj] >> 24) + dt - 2, 0);                       // causal: 2 frames of replicate pad in front
      int hu = min(max((int)((pk[jj] >> 12) & 0xfff) + dh - 1, 0), p.Ho - 1);
      int wu = min(max((int)(pk[jj] & 0xfff) + dw - 1, 0), p.Wo - 1);
      if (p.up_t == 2) tu = tu == 0 ? 0 : 1 + ((tu - 1) >> 1);             // frame 0 is not repeated in time (vae.py:190-199)
      if (p.up_s == 2) { hu >>= 1; wu >>= 1; }
      vxo[jj] = (uint32_t)((tu * p.Hs + hu) * p.Ws + wu) * cin2 + (jj < 4 ? xchunk_lo : xchunk_hi);
    }
  };
  auto set_dma_tile = [&](int ti) __attribute__((always_inline)) {
    int m0, n0;
    tile_origin(x_first + ti, m0, n0);
    rW = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.W) + (size_t)n0 * ldw2), 0, (int)((uint32_t)BN * ldw2), 0x00020000);
    const int mb = m0 + 128 * (wave >> 1) + 16 * (lane >> 3) + 8 * (wave & 1);   // + jj: 8 consecutive output positions
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int m = min(mb + jj, p.M - 1);
      const int wo = m % p.Wo, th = m / p.Wo;
      pk[jj] = (uint32_t)(th / p.Ho) << 24 | (uint32_t)(th % p.Ho) << 12 | (uint32_t)wo;
    }
    d_cnt = 0; d_tap = 0; d_cc = 0; vw = vw0;
    set_tap(0);
  };
  auto dma1 = [&](int stage, int d) __attribute__((always_inline)) {
    if (d < NDW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (cw_lds_t*)(dsm + stage * STAGE + (wslot0 + d) * CW_PAD), 16, vw ^ (d < 4 ? ww_lo : ww_hi),
                                                          (uint32_t)(wrow0 + d) * ldw2, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (cw_lds_t*)(dsm + stage * STAGE + W_OP + (xslot0 + d - NDW) * CW_PAD), 16, vxo[d - NDW],
                                                  (uint32_t)d_cc * (2 * CBK), 0, 0);
  };
  // after a K-tile's DMAs: next channel slab / tap / output tile (past the last tile it wraps onto the same one: harmless
  // loads that keep the vmcnt arithmetic uniform)
  auto dma_advance = [&]() __attribute__((always_inline)) {
    vw += 2 * CBK;
    if (++d_cc == cpt) { d_cc = 0; if (++d_tap < 27) set_tap(d_tap); }
    if (++d_cnt == nk) {
      if (d_ti + per_xcd < x_cnt) d_ti += per_xcd;
      set_dma_tile(d_ti);
    }
  };

  const uint32_t lds0 = (uint32_t)(uintptr_t)(cw_lds_t*)dsm;
  uint32_t wbs[2], xbs[2];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    const int lcs = lc ^ (((l15 + 4) >> 3) & 1);
    wbs[st] = lds0 + st * STAGE + ((NTW == 8 ? 16 * wn : 0) + l15) * CW_PAD + lcs * 16 + (NTW == 8 ? 0 : NTW * wn * 128);
    xbs[st] = lds0 + st * STAGE + W_OP + (16 * wm + l15) * CW_PAD + lcs * 16;
    asm volatile("" : "+v"(wbs[st]), "+v"(xbs[st]));
  }
  bf16x8 wf0[NTW], xf0[8], wf1[2][NTW], xf1[2][8];
  f32x4 acc[NTW][8];   // [n-tile][m-tile]; written (not accumulated) by the first k-step of every output tile

#define CW_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define CW_MF(WF, XF, Q) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[(Q) % NTW][(Q) / NTW]) : "v"(WF[(Q) % NTW]), "v"(XF[(Q) / NTW]))
#define CW_MF0(WF, XF, Q) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[(Q) % NTW][(Q) / NTW]) : "v"(WF[(Q) % NTW]), "v"(XF[(Q) / NTW]))

  set_dma_tile(slot);
#pragma unroll
  for (int d = 0; d < ND; ++d) dma1(0, d);
  dma_advance();
#pragma unroll
  for (int d = 0; d < ND; ++d) dma1(1, d);
  dma_advance();
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ND) : "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < NTW; ++i) { CW_RD(wf0[i], wbs[0], i * 128); CW_RD(wf1[0][i], wbs[0], i * 128 + 64); }
#pragma unroll
  for (int j = 0; j < 8; ++j) { CW_RD(xf0[j], xbs[0], j * 128); CW_RD(xf1[0][j], xbs[0], j * 128 + 64); }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // One K-tile (stage st) = TOT MFMAs, m = 0..TOT-1 (schedule: see gemm_bf16_w4_kernel):
  //   top: barrier (every wave holds this K-tile's fragments -> its stage may be refilled)
  //   m = DS d: DMA d of K-tile t+2;  after m = HALF-1: vmcnt(NB) + barrier (K-tile t+1 landed)
  //   second half: the 2 NR fragment reads of K-tile t+1 (k-step 0 into wf0/xf0, k-step 1 into the idle buffer), spread out
  constexpr int DS = TOT / 16;
  constexpr int NB = (HALF + DS - 1) / DS < ND ? (HALF + DS - 1) / DS : ND;
  static_assert(NTW == 8 ? HALF + 2 * (2 * NR - 1) < TOT : HALF + (2 * NR - 1) + (2 * NR - 1) / 3 < TOT, "fragment reads do not fit");
  auto ktile = [&](auto STC, auto FIRSTC) __attribute__((always_inline)) {
    constexpr int st = decltype(STC)::value;
    constexpr bool first = decltype(FIRSTC)::value;
    asm volatile("s_barrier" ::: "memory");
    auto chunk = [&](auto BASEC) __attribute__((always_inline)) {
#pragma unroll
      for (int m = decltype(BASEC)::value; m < decltype(BASEC)::value + 16; ++m) {
        if (m < HALF && first) CW_MF0(wf0, xf0, m);
        else if (m < HALF) CW_MF(wf0, xf0, m);
        else CW_MF(wf1[st], xf1[st], m - HALF);
        if (m % DS == 0 && m / DS < ND) dma1(st, m / DS);
        if (m == HALF - 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NB) : "memory");
        {   // fragment read r of K-tile t+1 in this slot?  (NTW = 8: every second slot of the second half; NTW = 4: three of four)
          const int sl = m - HALF;
          const bool has = sl >= 0 && (NTW == 8 ? (sl & 1) == 0 : (sl & 3) != 3);
          const int r = NTW == 8 ? sl / 2 : sl - sl / 4;
          if (has && r < 2 * NR) {
            const int rr = r % NR;
            if (r < NR) { if (rr < NTW) CW_RD(wf0[rr % NTW], wbs[st ^ 1], (rr % NTW) * 128); else CW_RD(xf0[(rr - NTW) & 7], xbs[st ^ 1], ((rr - NTW) & 7) * 128); }
            else { if (rr < NTW) CW_RD(wf1[st ^ 1][rr % NTW], wbs[st ^ 1], (rr % NTW) * 128 + 64); else CW_RD(xf1[st ^ 1][(rr - NTW) & 7], xbs[st ^ 1], ((rr - NTW) & 7) * 128 + 64); }
          }
        }
      }
    };
    chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 16>{}); chunk(std::integral_constant<int, 32>{});
    chunk(std::integral_constant<int, 48>{});
    if (TOT > 64) {
      chunk(std::integral_constant<int, (TOT > 64 ? 64 : 0)>{}); chunk(std::integral_constant<int, (TOT > 64 ? 80 : 0)>{});
      chunk(std::integral_constant<int, (TOT > 64 ? 96 : 0)>{}); chunk(std::integral_constant<int, (TOT > 64 ? 112 : 0)>{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    dma_advance();
  };

  for (int ti = slot; ti < x_cnt; ti += per_xcd) {
    ktile(std::integral_constant<int, 0>{}, std::true_type{});
    ktile(std::integral_constant<int, 1>{}, std::false_type{});
    for (int t = 2; t < nk; t += 2) {
      ktile(std::integral_constant<int, 0>{}, std::false_type{});
      ktile(std::integral_constant<int, 1>{}, std::false_type{});
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the asm MFMAs are invisible to the hazard recogniser
    int m0, n0;
    tile_origin(x_first + ti, m0, n0);
    conv_w4_epilogue<NTW, RESID, STATS>(p, acc, n0, m0 >> 8, [&](int row) __attribute__((always_inline)) { return m0 + row; });
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): nothing may still be loading into a VGPR when the asm stream resumes
  }
#undef CW_MF
#undef CW_MF0
#undef CW_RD
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wrapped DMAs still write this workgroup's LDS
}

template <int NTW, bool RESID, bool STATS>
int launch_conv_w4(const ConvW4P& p, int num_cu, hipStream_t stream) {
  constexpr int LDS = 2 * ((32 * NTW / 8) * CW_PAD + 32 * CW_PAD);
  static const hipError_t attr_rc = hipFuncSetAttribute((const void*)conv3d_w4_kernel<NTW, RESID, STATS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);   // once, thread-safe
  if (attr_rc != hipSuccess) return K5_ERR_HIP;
  const int tiles = p.tiles_m * p.tiles_n;
  hipLaunchKernelGGL((conv3d_w4_kernel<NTW, RESID, STATS>), dim3(min(tiles, num_cu)), dim3(256), LDS, stream, p);
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same convolution with the ACTIVATION operand staged once per (frame tap, 64-channel slab) as an LDS halo tile.
//
// conv3d_w4_kernel re-gathers its 256 positions from L2 for every one of the 27 taps, and its pace is set by that L2 -> LDS stream
// (refreshing 2 of the 8 position pieces per K-tile instead of all of them: 92.1 -> 80.8 ms over the convs of
// one decode tile — the bound of what this kernel can gain).  Here an output tile is an 8 x 32 patch of one frame.  For a frame tap
// dt and a channel slab, its (8 + 2) x (32 + 2) input positions — replicate-clamped at the frame's edges, the nearest upsample
// folded into the source address — are loaded ONCE (340 rows x 128 B, 44 one-KB DMA pieces, 11 per wave) and serve the nine
// (dh, dw) taps: a K-tile's activation stream drops from 32 KB to 4.9 KB.  The K walk is therefore (dt, slab, (dh, dw)) instead of
// (tap, slab); the weight operand, the MFMA / fragment-read / barrier schedule and the epilogue are those of the kernel above.
//
// LDS: two weight stages (as above) and two halo buffers; group g = (dt, slab) computes from buffer g & 1 while the halo of group
// g + 1 arrives in the other one (DMAs in the second halves of the group's first three K-tiles).  Halo row R = 34 hr + wr sits at
// byte 128 R with its eight 16-B chunks XOR-swizzled by R & 6: a fragment read takes 16 CONSECUTIVE rows from an arbitrary start
// (the tap shift), and with that key the four lane groups ds_read_b128 is serviced in (MI355X_MICROARCH.md: rows {0-3, 12-15} at
// chunk c together with rows 4-11 at chunk c + 1) touch 16 distinct 16-B bank slots for every start row — rows r and r + 8 share
// parity and key but sit on opposite sides of the c / c + 1 split.  The price: a lane's fragment addresses depend on the tap through
// the key, so they are recomputed per K-tile (4 patch rows x 7 VALU instructions) instead of being base + immediate.
//
// Range: that of the kernel above and Ho % 8 == 0, Wo % 32 == 0 (every decoder level of the 768 x 512 clips; the statistics
// blocks are then the same 2 M / 256 as above).
constexpr int HALO_W = 34, HALO_ROWS = 340, HALO_PIECES = 11 /* per wave */, HALO_BUF = 4 * HALO_PIECES * 1024;

template <int NTW, int S, bool RESID, bool STATS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3d_halo_kernel(ConvW4P p) {
  constexpr int BN = 32 * NTW;
  constexpr int WP = BN / 8;
  constexpr int W_OP = WP * CW_PAD, H0 = S * W_OP;             // weight stage (S of them); byte offset of halo buffer 0
  static_assert(H0 % 128 == 0 && HALO_BUF % 128 == 0, "halo rows are 128-byte aligned");
  constexpr int HALF = 8 * NTW, TOT = 16 * NTW;
  constexpr int NDW = WP / 4;                                  // weight DMA instructions per wave and K-tile
  constexpr int NR = NTW + 8;
  extern __shared__ __attribute__((aligned(128))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1;
  const int l15 = lane & 15, lc = lane >> 4;

  const int nblk = p.tiles_m * p.tiles_n;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = (gridDim.x + 7 - xcd) >> 3;
  const int q8 = nblk >> 3, r8 = nblk & 7;
  const int x_first = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, x_cnt = q8 + (xcd < r8 ? 1 : 0);
  if (slot >= x_cnt) return;

  const int cpt = p.Cin / CBK;                 // slabs (even); groups per output tile: 3 cpt
  const uint32_t ldw2 = (uint32_t)(27 * p.Cin) * 2u, cin2 = (uint32_t)p.Cin * 2u;
  const uint32_t vw0 = (uint32_t)(16 * (lane >> 3)) * ldw2 + (uint32_t)(lane & 7) * 16u;
  const int wrow0 = NTW == 8 ? 128 * (wave >> 1) + 8 * (wave & 1) : NDW * wave;
  const int wslot0 = NDW * wave;
  const uint32_t sw_lo = (wave & 1) ? 16u : 0u, sw_hi = 16u - sw_lo;
  const uint32_t ww_lo = NTW == 8 ? sw_lo : ((wave == 1 || wave == 2) ? 16u : 0u);
  const uint32_t ww_hi = NTW == 8 ? sw_hi : ww_lo;
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rW;

  // weight cursor: the K-tile whose weights are issued next = (tile w_ti, frame tap w_dt, slab w_sl, in-plane tap w_i)
  int w_ti = slot, w_dt = 0, w_sl = 0, w_i = 0;
  uint32_t vw = vw0;
  auto set_w_tile = [&](int ti) __attribute__((always_inline)) {
    const int n0 = ((x_first + ti) % p.tiles_n) * BN;
    rW = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.W) + (size_t)n0 * ldw2), 0, (int)((uint32_t)BN * ldw2), 0x00020000);
  };
  auto dma_w = [&](int stage, int d) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (cw_lds_t*)(dsm + stage * W_OP + (wslot0 + d) * CW_PAD), 16, vw ^ (d < 4 ? ww_lo : ww_hi),
                                             (uint32_t)(wrow0 + d) * ldw2, 0, 0);
  };
  auto advance_w = [&]() __attribute__((always_inline)) {   // past the last tile it wraps onto the same one: harmless loads, uniform vmcnt arithmetic
    if (++w_i == 9) {
      w_i = 0;
      if (++w_sl == cpt) {
        w_sl = 0;
        if (++w_dt == 3) {
          w_dt = 0;
          if (w_ti + per_xcd < x_cnt) w_ti += per_xcd;
          set_w_tile(w_ti);
        }
      }
    }
    vw = vw0 + ((uint32_t)((w_dt * 9 + w_i) * p.Cin + w_sl * CBK)) * 2u;
  };

  // halo cursor: the group whose halo is issued next = (tile h_ti: frame h_t, patch origin (h_h0, h_w0) - 1, frame tap h_dt, slab h_sl)
  int h_ti = slot, h_dt = 0, h_sl = 0, h_t = 0, h_h0 = 0, h_w0 = 0;
  auto set_halo_tile = [&](int ti) __attribute__((always_inline)) {
    const int mt = (x_first + ti) / p.tiles_n;
    const int wbi = mt % p.wb, hbi = (mt / p.wb) % p.hb;
    h_t = mt / (p.wb * p.hb);
    h_h0 = 8 * hbi - 1; h_w0 = 32 * wbi - 1;
  };
  auto halo_soff = [&]() __attribute__((always_inline)) {
    int tu = max(h_t + h_dt - 2, 0);                           // causal: 2 frames of replicate pad in front
    if (p.up_t == 2) tu = tu == 0 ? 0 : 1 + ((tu - 1) >> 1);   // frame 0 is not repeated in time (vae.py:190-199)
    return (uint32_t)(tu * p.Hs * p.Ws) * cin2 + (uint32_t)h_sl * (2 * CBK);
  };
  uint32_t h_soff = 0;
  const int up_sh = p.up_s == 2 ? 1 : 0;
  const int hrow0 = tid >> 3;                                                    // halo row of this lane in piece `wave` (+ 32 per further piece)
  const uint32_t hchunk = (uint32_t)(((lane & 7) ^ ((lane >> 3) & 6)) << 4);     // its chunk under the row's swizzle key (R & 6: the same in every piece)
  // piece pi of this wave = halo rows 8 (wave + 4 pi) .. + 7: the lane's (clamped, upsample-mapped) source position is derived at issue
  // (13 VALU instructions; eleven address registers per lane would not fit beside the 192 fragment registers of the 256 x 256 tile)
  auto dma_h = [&](int pi) __attribute__((always_inline)) {
    int r0 = hrow0;
    asm volatile("" : "+v"(r0));                            // derived here, every time: hoisted out of the loop these are 22 registers
    const int Rc = min(r0 + 32 * pi, HALO_ROWS - 1);        // the last piece's 12 spare rows repeat row 339
    const int hr = (Rc * 241) >> 13, wr = Rc - HALO_W * hr; // Rc / 34 for Rc < 1000
    const int hu = min(max(h_h0 + hr, 0), p.Ho - 1) >> up_sh, wu = min(max(h_w0 + wr, 0), p.Wo - 1) >> up_sh;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (cw_lds_t*)(dsm + H0 + (h_sl & 1) * HALO_BUF + (wave + 4 * pi) * 1024), 16,
                                             (uint32_t)(hu * p.Ws + wu) * cin2 + hchunk, h_soff, 0, 0);
  };
  auto advance_halo = [&]() __attribute__((always_inline)) {
    if (++h_sl == cpt) {
      h_sl = 0;
      if (++h_dt == 3) {
        h_dt = 0;
        if (h_ti + per_xcd < x_cnt) h_ti += per_xcd;
        set_halo_tile(h_ti);
      }
    }
    h_soff = halo_soff();
  };

  // fragment addresses: a lane reads its 16 halo rows per patch row at  hb + 128 R + 16 ((lc + 4 ks) ^ (R & 6)),  R = its row at tap
  // (0, 0) + 34 dh + dw — recomputed for every K-tile (the key moves with the tap), piecewise BETWEEN the MFMAs of the K-tile before the
  // reads that use it (left to the end of the K-tile the 28 instructions are exposed: the MFMA stream has drained by then).
  const uint32_t lds0 = (uint32_t)(uintptr_t)(cw_lds_t*)dsm;
  uint32_t hb_cur = lds0 + H0, hb_nxt = lds0 + H0 + HALO_BUF;   // halo buffer of the current / the next group
  const int xr0 = (4 * wm) * HALO_W + l15;     // halo row of this lane's position in patch row 4 wm at tap (0, 0)
  uint32_t xa[4][2];                           // [patch row of the wave][k-step]: byte address of the lane's 16-B chunk
  auto xa_part = [&](int xr, int jh, int ks, int tapi, uint32_t hb) __attribute__((always_inline)) {
    const uint32_t R = (uint32_t)(xr + HALO_W * jh + (tapi / 3) * HALO_W + (tapi % 3));
    xa[jh][ks] = hb + (R << 7) + ((((uint32_t)lc << 4) ^ ((R & 6u) << 4)) ^ (ks ? 64u : 0u));
  };

  // weight stages: K-tile t sits in stage t % S; s_fill = the stage refilled during the current K-tile (its own), wrd = the lane's
  // read base in the stage of the NEXT K-tile
  uint32_t wb0 = lds0 + ((NTW == 8 ? 16 * wn : 0) + l15) * CW_PAD + (lc ^ (((l15 + 4) >> 3) & 1)) * 16 + (NTW == 8 ? 0 : NTW * wn * 128);
  asm volatile("" : "+v"(wb0));
  int s_fill = 0;
  uint32_t wrd = wb0;
  auto advance_stage = [&]() __attribute__((always_inline)) {
    s_fill = s_fill + 1 == S ? 0 : s_fill + 1;
    wrd = wb0 + (uint32_t)((s_fill + 1 == S ? 0 : s_fill + 1) * W_OP);
  };
  bf16x8 wf0[NTW], xf0[8], wf1[2][NTW], xf1[2][8];
  f32x4 acc[NTW][8];

#define CW_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define CW_MF(WF, XF, Q) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[(Q) % NTW][(Q) / NTW]) : "v"(WF[(Q) % NTW]), "v"(XF[(Q) / NTW]))

  // prologue: weights of K-tiles 0 and 1, the halo of group 0, the fragments of K-tile 0
  set_w_tile(slot);
  set_halo_tile(slot);
  h_soff = halo_soff();
#pragma unroll
  for (int st = 0; st < S; ++st) {
#pragma unroll
    for (int d = 0; d < NDW; ++d) dma_w(st, d);
    advance_w();
  }
#pragma unroll
  for (int pi = 0; pi < HALO_PIECES; ++pi) dma_h(pi);
  advance_halo();
#pragma unroll
  for (int jh = 0; jh < 4; ++jh) { xa_part(xr0, jh, 0, 0, hb_cur); xa_part(xr0, jh, 1, 0, hb_cur); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < NTW; ++i) { CW_RD(wf0[i], wrd, i * 128); CW_RD(wf1[0][i], wrd, i * 128 + 64); }
#pragma unroll
  for (int j = 0; j < 8; ++j) { CW_RD(xf0[j], xa[j >> 1][0], (j & 1) * 2048); CW_RD(xf1[0][j], xa[j >> 1][1], (j & 1) * 2048); }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  wrd = wb0 + (uint32_t)((S > 1 ? 1 : 0) * W_OP);
  if (S >= 3) {   // one-barrier schedule: K-tile 0 reads K-tile 1's k-step-1 fragments in its first half
#pragma unroll
    for (int jh = 0; jh < 4; ++jh) xa_part(xr0, jh, 1, 1, hb_cur);
  }

  // One K-tile = TOT MFMAs, m = 0..TOT-1; PAR = its parity (fragment double buffer), I = its in-plane tap (position in the group):
  //   top: barrier (every wave holds this K-tile's fragments -> its weight stage may be refilled)
  //   first half, m = DS d: weight DMA d of K-tile t+S;  after m = HALF-1: vmcnt(VM) + barrier — K-tile t+1's weights have landed:
  //     VM = everything issued after them = the weights of K-tiles t+2-S .. t and the halo pieces of K-tiles t+1-S .. t-1
  //   second half: the 2 NR fragment reads of K-tile t+1, spread out; in its free slots the halo pieces of the next group
  //     (4, 4, 3 pieces in the group's K-tiles 0, 1, 2)
  constexpr int DS = TOT / 16;
  constexpr bool ONEBAR = S >= 3;
  static_assert(NDW * DS <= HALF, "weight DMAs belong to the first half");
  static_assert(!ONEBAR || (NTW == 4 && 2 * NR <= HALF), "the one-barrier schedule is laid out for the 256 x 128 tile");
  auto ktile = [&](auto PARC, auto IC) __attribute__((always_inline)) {
    constexpr int par = decltype(PARC)::value, I = decltype(IC)::value;
    constexpr int NH = I < 2 ? 4 : I == 2 ? 3 : 0, HB = 4 * I;
    auto nhs = [](int i) constexpr { i = (i + 9) % 9; return i < 2 ? 4 : i == 2 ? 3 : 0; };
    constexpr int HSUM = (S > 1 ? nhs(I - 1) : 0) + (S > 2 ? nhs(I - 2) : 0) + (S > 3 ? nhs(I - 3) : 0);
    constexpr int VM = (ONEBAR ? S - 2 : S - 1) * NDW + HSUM;
    static_assert(S <= 4 && VM < 64, "vmcnt is a 6-bit field");
    if (ONEBAR) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(VM) : "memory");
    else asm volatile("s_barrier" ::: "memory");
    int xr = xr0;
    asm volatile("" : "+v"(xr));   // derived in this K-tile: hoisted per tap the rows are 36 registers (and spills)
    const uint32_t hb1 = I + 1 < 9 ? hb_cur : hb_nxt, hb2 = I + 2 < 9 ? hb_cur : hb_nxt;   // halo buffer of K-tile t+1 / t+2
    auto rd = [&](int r) __attribute__((always_inline)) {   // fragment read r of K-tile t+1: k-step r / NR, weights then positions
      const int rr = r % NR, j = (rr - NTW) & 7;
      if (r < NR) { if (rr < NTW) CW_RD(wf0[rr % NTW], wrd, (rr % NTW) * 128); else CW_RD(xf0[j], xa[j >> 1][0], (j & 1) * 2048); }
      else { if (rr < NTW) CW_RD(wf1[par ^ 1][rr % NTW], wrd, (rr % NTW) * 128 + 64); else CW_RD(xf1[par ^ 1][j], xa[j >> 1][1], (j & 1) * 2048); }
    };
    auto chunk = [&](auto BASEC) __attribute__((always_inline)) {
#pragma unroll
      for (int m = decltype(BASEC)::value; m < decltype(BASEC)::value + 16; ++m) {
        if (m < HALF) CW_MF(wf0, xf0, m);
        else CW_MF(wf1[par], xf1[par], m - HALF);
        if (m % DS == 0 && m / DS < NDW) dma_w(s_fill, m / DS);
        const int sl = m - HALF;
        if (!ONEBAR) {
          if (m == HALF - 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(VM) : "memory");
          // second half: reads in every second slot (NTW = 8) / three of four (NTW = 4), halo pieces in slots without one
          const bool has = sl >= 0 && (NTW == 8 ? (sl & 1) == 0 : (sl & 3) != 3);
          const int r = NTW == 8 ? sl / 2 : sl - sl / 4;
          if (has && r < 2 * NR) rd(r);
          if (NTW == 8 ? (sl >= 0 && (sl & 15) == 1 && sl / 16 < NH) : (sl >= 0 && (sl & 7) == 3 && sl / 8 < NH)) dma_h(HB + (NTW == 8 ? sl / 16 : sl / 8));
          // the addresses of K-tile t+1 (both k-steps), in eight slots of the first half
          if (m < HALF && m % (HALF / 8) == HALF / 16) xa_part(xr, (m / (HALF / 8)) >> 1, (m / (HALF / 8)) & 1, (I + 1) % 9, hb1);
        } else {
          // one barrier per K-tile: the weights of K-tile t+1 landed before it, so its fragments are read across the WHOLE K-tile —
          // k-step 1 (into the idle half of the double buffer) in the odd slots of the first half, k-step 0 (wf0 / xf0 are free once the
          // first half has issued) in those of the second; the last read is 8 MFMAs before the end
          const int hs = m < HALF ? m : sl;                    // slot within the half
          if ((hs & 1) && hs / 2 < NR) rd((m < HALF ? NR : 0) + hs / 2);
          if (sl >= 0 && (sl & 7) == 2 && sl / 8 < NH) dma_h(HB + sl / 8);
          // addresses: k-step 0 of K-tile t+1 in the first half (read in the second), k-step 1 of K-tile t+2 in the second half
          // (read in the next K-tile's first half; this K-tile's k-step-1 reads have been issued by then)
          if (m < HALF && (m & 3) == 2 && m / 4 < 4) xa_part(xr, m / 4, 0, (I + 1) % 9, hb1);
          if (sl >= 0 && (sl & 7) == 4) xa_part(xr, sl / 8, 1, (I + 2) % 9, hb2);
        }
      }
    };
    chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 16>{}); chunk(std::integral_constant<int, 32>{});
    chunk(std::integral_constant<int, 48>{});
    if (TOT > 64) {
      chunk(std::integral_constant<int, (TOT > 64 ? 64 : 0)>{}); chunk(std::integral_constant<int, (TOT > 64 ? 80 : 0)>{});
      chunk(std::integral_constant<int, (TOT > 64 ? 96 : 0)>{}); chunk(std::integral_constant<int, (TOT > 64 ? 112 : 0)>{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    advance_w();
    advance_stage();
    if (I == 2) advance_halo();
    if (I == 8) { const uint32_t t = hb_cur; hb_cur = hb_nxt; hb_nxt = t; }
  };
  auto group = [&](auto PARC) __attribute__((always_inline)) {   // nine K-tiles; PARC = the parity of the first one
    using P0 = decltype(PARC); using P1 = std::integral_constant<int, P0::value ^ 1>;
    ktile(P0{}, std::integral_constant<int, 0>{}); ktile(P1{}, std::integral_constant<int, 1>{}); ktile(P0{}, std::integral_constant<int, 2>{});
    ktile(P1{}, std::integral_constant<int, 3>{}); ktile(P0{}, std::integral_constant<int, 4>{}); ktile(P1{}, std::integral_constant<int, 5>{});
    ktile(P0{}, std::integral_constant<int, 6>{}); ktile(P1{}, std::integral_constant<int, 7>{}); ktile(P0{}, std::integral_constant<int, 8>{});
  };

  // every MFMA accumulates: the accumulators start at zero and the epilogue leaves them at zero (256 v_accvgpr_write per 27 Cin / 64
  // K-tiles; a write-only first K-tile would be a second copy of the group code below)
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();
  for (int ti = slot; ti < x_cnt; ti += per_xcd) {
    for (int gp = 0; gp < 3 * cpt; gp += 2) {   // the halo of group g + 1 arrives during the first three K-tiles of group g
      group(std::integral_constant<int, 0>{});
      group(std::integral_constant<int, 1>{});
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the asm MFMAs are invisible to the hazard recogniser
    const int lid = x_first + ti;
    const int mt = lid / p.tiles_n, n0 = (lid % p.tiles_n) * BN;
    const int wbi = mt % p.wb, hbi = (mt / p.wb) % p.hb, tt = mt / (p.wb * p.hb);
    const int morg = (tt * p.Ho + 8 * hbi) * p.Wo + 32 * wbi;
#ifndef K5_CONV_ABLATE_EPI   // timing ablation (wrong results): no epilogue
    conv_w4_epilogue<NTW, RESID, STATS>(p, acc, n0, mt, [&](int row) __attribute__((always_inline)) { return morg + (row >> 5) * p.Wo + (row & 31); });
#endif
    zero_acc();
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): nothing may still be loading into a VGPR when the asm stream resumes
  }
#undef CW_MF
#undef CW_RD
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wrapped DMAs still write this workgroup's LDS
}

template <int NTW, bool RESID, bool STATS>
int launch_conv_halo(const ConvW4P& p, int num_cu, hipStream_t stream) {
  constexpr int S = NTW == 8 ? 2 : 4;   // weight stages: what fits beside the two halo buffers
  constexpr int LDS = S * (32 * NTW / 8) * CW_PAD + 2 * HALO_BUF;
  static_assert(LDS <= 160 * 1024, "LDS");
  static const hipError_t attr_rc = hipFuncSetAttribute((const void*)conv3d_halo_kernel<NTW, S, RESID, STATS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);   // once, thread-safe
  if (attr_rc != hipSuccess) return K5_ERR_HIP;
  const int tiles = p.tiles_m * p.tiles_n;
  hipLaunchKernelGGL((conv3d_halo_kernel<NTW, S, RESID, STATS>), dim3(min(tiles, num_cu)), dim3(256), LDS, stream, p);
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}

}  // namespace

// Same contract as k5_launch_conv3d_bf16 (conv3d.hip); returns K5_ERR_UNSUPPORTED when the shape is outside this kernel's
// range (the caller then uses the 128 x 128 kernel).
// quad_stats (may be null): [2 ceil(M / 256)][Cout / 4][2] fp32, filled with the per-128-row (sum, sum of squares) of the stored
// outputs per channel quad — GroupNorm statistics without another pass over the tensor (k5_launch_groupnorm_bf16_quads).
int k5_launch_conv3d_w4(const void* X, const void* W, const float* bias, void* out, int Ts, int Hs, int Ws, int Cin, int Cout,
                        int up_t, int up_s, int ldc, const void* resid, int ldr, float* quad_stats, hipStream_t stream) {
  if (Cin <= 0 || (Cin % 128) || !(Cout == 128 || (Cout % 256) == 0) || (ldc & 7) || (resid && (ldr & 3)) || !bias) return K5_ERR_UNSUPPORTED;
  ConvW4P p;
  p.X = (const bf16_t*)X; p.W = (const bf16_t*)W; p.C = (bf16_t*)out; p.bias = bias; p.resid = (const bf16_t*)resid;
  p.Ts = Ts; p.Hs = Hs; p.Ws = Ws;
  p.To = up_t == 2 ? 2 * Ts - 1 : Ts; p.Ho = up_s * Hs; p.Wo = up_s * Ws;
  p.up_t = up_t; p.up_s = up_s; p.Cin = Cin; p.Cout = Cout; p.ldc = ldc; p.ldr = ldr;
  const long long M = (long long)p.To * p.Ho * p.Wo, xb = (long long)Ts * Hs * Ws * Cin * 2;
  if (M > 0x7fffffffLL || xb >= 0xffffffffLL || p.Ho > 4095 || p.Wo > 4095 || p.To > 255) return K5_ERR_UNSUPPORTED;
  p.M = (int)M; p.x_bytes = (unsigned)xb;
  const int bn = Cout == 128 ? 128 : 256;
  p.tiles_m = (p.M + 255) / 256; p.tiles_n = Cout / bn;
  static const int num_cu = [] { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1; return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; }();   // once, thread-safe
  if (num_cu <= 0) return K5_ERR_HIP;
  // from 5/8 of a round up (measured on the tile shapes of the tiling policy, tools/vae_shapes.py: (5,64,96) 103.7 -> 102.2 ms,
  // (6,52,84) 92.6 -> 90.9, (5,64,64) 70.0 -> 68.5, (5,32,32) 21.3 -> 20.2; round 1 required a whole round); below that the
  // 128 x 128 tiles fill the chip better.  K5_CONV_MIN_FILL8 = n: n/8 of a round (A/B switch for benchmarking)
  static const int min_fill8 = getenv("K5_CONV_MIN_FILL8") ? atoi(getenv("K5_CONV_MIN_FILL8")) : 5;
  if ((long long)p.tiles_m * p.tiles_n * 8 < (long long)num_cu * min_fill8) return K5_ERR_UNSUPPORTED;
  p.quad_stats = quad_stats;
  k5_conv3d_set_last_kind(quad_stats ? K5_CONV_KIND_W4_STATS : K5_CONV_KIND_W4);
  // the halo-tile form where the output frames split into whole 8 x 32 patches (K5_CONV_HALO=0: A/B switch for benchmarking)
  static const bool halo_ok = !(getenv("K5_CONV_HALO") && atoi(getenv("K5_CONV_HALO")) == 0);
  p.hb = p.Ho / 8; p.wb = p.Wo / 32;
  if (halo_ok && (p.Ho % 8) == 0 && (p.Wo % 32) == 0) {
    if (quad_stats) {
      if (Cout == 128) return resid ? launch_conv_halo<4, true, true>(p, num_cu, stream) : launch_conv_halo<4, false, true>(p, num_cu, stream);
      return resid ? launch_conv_halo<8, true, true>(p, num_cu, stream) : launch_conv_halo<8, false, true>(p, num_cu, stream);
    }
    if (Cout == 128) return resid ? launch_conv_halo<4, true, false>(p, num_cu, stream) : launch_conv_halo<4, false, false>(p, num_cu, stream);
    return resid ? launch_conv_halo<8, true, false>(p, num_cu, stream) : launch_conv_halo<8, false, false>(p, num_cu, stream);
  }
  if (quad_stats) {
    if (Cout == 128) return resid ? launch_conv_w4<4, true, true>(p, num_cu, stream) : launch_conv_w4<4, false, true>(p, num_cu, stream);
    return resid ? launch_conv_w4<8, true, true>(p, num_cu, stream) : launch_conv_w4<8, false, true>(p, num_cu, stream);
  }
  if (Cout == 128) return resid ? launch_conv_w4<4, true, false>(p, num_cu, stream) : launch_conv_w4<4, false, false>(p, num_cu, stream);
  return resid ? launch_conv_w4<8, true, false>(p, num_cu, stream) : launch_conv_w4<8, false, false>(p, num_cu, stream);
}
