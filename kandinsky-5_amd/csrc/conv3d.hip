// conv3d.hip — HunyuanVideo causal 3x3x3 convolution as an implicit GEMM on MFMA (gfx950).
//
// Replaces HunyuanVideoCausalConv3d (kandinsky/models/vae.py:125-163: replicate pad W(1,1) H(1,1) T(2,0) then
// nn.Conv3d k=3, autocast bf16) and, folded into the operand gather, the nearest-neighbour resize of
// HunyuanVideoUpsampleCausal3D (vae.py:187-205: frame 0 x(2,2), frames 1.. x(2,2,2)).
//
// Activations are CHANNELS-LAST bf16 [T][H][W][C] (C a multiple of 64), so the im2col row of an output position
// is 27 taps x C contiguous channels: a K-tile of 64 channels of one tap is ONE contiguous 128-B segment of the
// input.  The kernel is the direct-to-LDS GEMM of gemm_bf16.hip (128x128x64 tile, global_load_lds_dwordx4 with the
// swizzle applied to the per-lane source address) whose A-row pointers are re-derived per tap:
//     (to,ho,wo) --tap--> clamp (replicate pad, causal in T) --upsample map--> (ts,hs,ws) --> X + pos*C + c0.
// Weights are pre-packed [Cout][27][Cin] (tap-major) so the W operand is a plain K-contiguous row.
// Epilogues: bias, or bias + residual add (resnet skip, vae.py:274: bf16(bf16(acc+bias) + residual)).
#include <stdlib.h>

#include "k5_common.h"
#include "k5_kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 128;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

struct ConvP {
  const bf16_t* X; const bf16_t* W; bf16_t* C;
  const float* bias; const bf16_t* resid;
  int Ts, Hs, Ws;     // source grid
  int To, Ho, Wo;     // output grid (= source, its upsampled size, or its strided size)
  int Hu, Wu;         // clamp extent of the gather = up_s * (Hs, Ws)
  int up_t, up_s;     // nearest-upsample factors folded into the gather (1 or 2)
  int st_t, st_s;     // output stride (encoder downsample, vae.py:208-227): source = out * stride + tap - pad; 1 with an upsample
  int Cin, Cout, M, ldc, ldr;
  int tiles_m, tiles_n;
};

template <bool RESID>
__global__ __launch_bounds__(256) void conv3d_kernel(ConvP p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
  char* sA = smem;
  char* sW = smem + 2 * TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, hi = lane >> 5, l31 = lane & 31;

  const int nblk = p.tiles_m * p.tiles_n;
  const int lid = xcd_remap(blockIdx.x, nblk);
  const int tm = lid / p.tiles_n, tn = lid % p.tiles_n;  // n fastest: the (few) Cout tiles of one position block are neighbours
  const int m0 = tm * BM, n0 = tn * BN;

  // this wave stages pieces wave*4 .. wave*4+3 (8 rows x 128 B); per row: output coordinates + swizzled chunk
  int ro_t[4], ro_h[4], ro_w[4], rc[4];
  const bf16_t* gw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * (wave * 4 + i) + (lane >> 3);
    rc[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    const int m = min(m0 + row, p.M - 1);
    ro_w[i] = m % p.Wo;
    const int th = m / p.Wo;
    ro_h[i] = th % p.Ho;
    ro_t[i] = th / p.Ho;
    gw[i] = p.W + (size_t)min(n0 + row, p.Cout - 1) * (27 * p.Cin) + rc[i];
  }
  const bf16_t* ga[4];
  auto set_tap = [&](int tap) {
    const int dt = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int tu = max(ro_t[i] * p.st_t + dt - 2, 0);              // causal: 2 frames of replicate pad in front
      int hu = min(max(ro_h[i] * p.st_s + dh - 1, 0), p.Hu - 1);   // Hu, Wu: extent of the (virtually upsampled) source grid
      int wu = min(max(ro_w[i] * p.st_s + dw - 1, 0), p.Wu - 1);
      if (p.up_t == 2) tu = tu == 0 ? 0 : 1 + ((tu - 1) >> 1); // frame 0 is not repeated in time (vae.py:190-199)
      if (p.up_s == 2) { hu >>= 1; wu >>= 1; }
      ga[i] = p.X + ((size_t)(tu * p.Hs + hu) * p.Ws + wu) * p.Cin + rc[i];
    }
  };
  auto stage = [&](int buf, int c0, int kw) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = (wave * 4 + i) * 1024;
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(ga[i] + c0), (lds_void_t*)(sA + buf * TILE_BYTES + piece), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void_t*)(gw[i] + kw), (lds_void_t*)(sW + buf * TILE_BYTES + piece), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int cpt = p.Cin / BK;       // K-tiles per tap
  const int nk = 27 * cpt;
  set_tap(0);
  stage(0, 0, 0);
  int tap = 0, cc = 0;              // position of the tile being PREFETCHED next
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    __syncthreads();
    if (kt + 1 < nk) {
      if (++cc == cpt) { cc = 0; ++tap; set_tap(tap); }
      stage(buf ^ 1, cc * BK, (kt + 1) * BK);
    }
    const char* cA = sA + buf * TILE_BYTES;
    const char* cW = sW + buf * TILE_BYTES;
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
  }

  // epilogue: lane owns output position m, 4 consecutive channels per register group
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + wm * 64 + j * 32 + l31;
    if (m >= p.M) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int n = n0 + wn * 64 + i * 32 + 8 * rg + 4 * hi;
        if (n >= p.Cout) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * rg + e] + ((n + e < p.Cout) ? p.bias[n + e] : 0.f);
        if (RESID) {
          const bf16_t* rp = p.resid + (size_t)m * p.ldr + n;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.Cout) v[e] = bf2f(rp[e]) + bf_round(v[e]);
        }
        bf16_t* cp = p.C + (size_t)m * p.ldc + n;
        if (n + 3 < p.Cout && (p.ldc & 3) == 0) {
          u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *reinterpret_cast<u32x2*>(cp) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (n + e < p.Cout) cp[e] = f2bf(v[e]);
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------
// Cout = 3 (the decoder's conv_out, 128 -> 3 at full resolution, vae.py:689).  As an implicit-GEMM tile it spends 125 of 128 output
// columns on zeros and gathers every input row 27 times through L2 (46 GB per launch, 6.2 ms per tile of the 5 s clip).  Here the
// convolution is split the other way round:
//     Y[pixel][(o, kh, kw)] = sum_c X[pixel][c] W[o][kt, kh, kw][c]     a GEMM with K = Cin only and N = 27 (padded to 32), on MFMA,
//     out[p][o] = sum_kt sum_kh,kw Y_kt[p + (kh - 1, kw - 1)][(o, kh, kw)]   a 27-term shift-and-add per output, through LDS.
// A workgroup owns 8 x 32 output pixels of one frame; per source frame kt it multiplies the 10 x 34 replicate-clamped input patch
// (340 pixels = 22 m-tiles of 16, A fragments straight from global memory: 64 contiguous bytes per pixel and k-step) by the 27
// weight rows of that kt (B fragments in registers), writes Y (fp32, 352 x 33 floats) to LDS, and every thread adds up its own
// pixel's 27 entries (consecutive lanes = consecutive pixels = 33 floats apart: conflict-free).  Each input element is read
// 3 x 1.33 times instead of 27; the MFMA work is 0.26 TFLOP instead of 6 padded.
// (Two VALU formulations were measured first: v_dot2c_f32_bf16 / v_pk_fma_f32 with scalar weights, 5.3-7.1 ms per launch — bound by
// the scalar weight loads, 20 KB of weights do not stay in the 16-KB scalar cache.)
// ---------------------------------------------------------------------------------------------
template <int KS>   // Cin = 32 KS
__global__ __launch_bounds__(256) void conv3d_out3_kernel(ConvP p) {
  constexpr int PW = 34, PH = 10, NPIX = PW * PH, MT = (NPIX + 15) / 16, YS = 33;
  __shared__ float Y[MT * 16 * YS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lc = lane >> 4;
  const int tiles_w = (p.Wo + 31) / 32, tiles_h = (p.Ho + 7) / 8;
  const int tw = blockIdx.x % tiles_w, th = (blockIdx.x / tiles_w) % tiles_h, t = blockIdx.x / (tiles_w * tiles_h);
  const int h0 = th * 8, w0 = tw * 32;
  const int gr = tid >> 5, gc = tid & 31;             // this thread's output pixel in the gather phase
  float acc[3] = {0.f, 0.f, 0.f};
  for (int kt = 0; kt < 3; ++kt) {
    const int ts = max(t + kt - 2, 0);                // causal: two replicated frames in front (vae.py:141-147)
    bf16x8 bfr[2][KS];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = 16 * j + l15;                     // column (o, kh, kw) = (n / 9, n % 9 / 3, n % 3); 27..31 are padding
      const bf16_t* wp = p.W + ((size_t)(n < 27 ? n / 9 : 0) * 27 + 9 * kt + (n < 27 ? n % 9 : 0)) * p.Cin + 8 * lc;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bfr[j][ks] = *reinterpret_cast<const bf16x8*>(wp + 32 * ks);
        if (n >= 27) bfr[j][ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
    bf16x8 a[(MT + 3) / 4][KS];
#pragma unroll
    for (int i = 0; i < (MT + 3) / 4; ++i) {
      const int mt = wave + 4 * i;
      const int pix = min(16 * mt + l15, NPIX - 1), rr = pix / PW, cc = pix - rr * PW;
      const int hs = min(max(h0 - 1 + rr, 0), p.Ho - 1), ws = min(max(w0 - 1 + cc, 0), p.Wo - 1);
      const bf16_t* xp = p.X + (((size_t)ts * p.Ho + hs) * p.Wo + ws) * p.Cin + 8 * lc;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[i][ks] = *reinterpret_cast<const bf16x8*>(xp + 32 * ks);
    }
    if (kt) __syncthreads();                          // the previous frame's Y has been gathered
#pragma unroll
    for (int i = 0; i < (MT + 3) / 4; ++i) {
      const int mt = wave + 4 * i;
      if (mt < MT) {
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], bfr[0][ks], d0, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], bfr[1][ks], d1, 0, 0, 0);
        }
        float* yp = Y + (16 * mt + 4 * lc) * YS + l15;   // D: lane l15 = column, registers = rows (pixels) 4 lc .. 4 lc + 3
#pragma unroll
        for (int e = 0; e < 4; ++e) { yp[e * YS] = d0[e]; yp[e * YS + 16] = d1[e]; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float* yq = Y + ((gr + kh) * PW + gc + kw) * YS + 3 * kh + kw;
        acc[0] += yq[0]; acc[1] += yq[9]; acc[2] += yq[18];
      }
  }
  const int h = h0 + gr, w = w0 + gc;
  if (h < p.Ho && w < p.Wo) {
    bf16_t* cp = p.C + ((size_t)(t * p.Ho + h) * p.Wo + w) * p.ldc;
#pragma unroll
    for (int o = 0; o < 3; ++o) cp[o] = f2bf(acc[o] + p.bias[o]);
  }
}

}  // namespace

// X [Ts][Hs][Ws][Cin] bf16 ; W [Cout][27][Cin] bf16 ; bias fp32 [Cout] ; out [To*Ho*Wo][ldc] bf16
// (To,Ho,Wo) = (Ts,Hs,Ws) scaled by the folded nearest upsample: Ho = up_s*Hs, Wo = up_s*Ws, To = up_t==2 ? 2*Ts-1 : Ts.
static thread_local int g_conv_last_kind = -1;
int k5_conv3d_last_kind() { return g_conv_last_kind; }
void k5_conv3d_set_last_kind(int kind) { g_conv_last_kind = kind; }

int k5_launch_conv3d_bf16(const void* X, const void* W, const float* bias, void* out, int Ts, int Hs, int Ws, int Cin,
                          int Cout, int up_t, int up_s, int ldc, const void* resid, int ldr, hipStream_t stream) {
  return k5_launch_conv3d_bf16_strided(X, W, bias, out, Ts, Hs, Ws, Cin, Cout, up_t, up_s, 1, 1, ldc, resid, ldr, stream);
}

// The same with an output stride (st_t, st_s in {1, 2}; only without upsample): HunyuanVideoDownsampleCausal3D (vae.py:208-227,
// encoder).  Output grid: To = (Ts - 1) / st_t + 1, Ho = (Hs - 1) / st_s + 1, Wo likewise (pad (1,1),(1,1),(2,0), kernel 3, padding 0).
int k5_launch_conv3d_bf16_strided(const void* X, const void* W, const float* bias, void* out, int Ts, int Hs, int Ws, int Cin,
                                  int Cout, int up_t, int up_s, int st_t, int st_s, int ldc, const void* resid, int ldr, hipStream_t stream) {
  if (Ts <= 0 || Hs <= 0 || Ws <= 0 || Cout <= 0 || !bias) return K5_ERR_ARG;
  if (Cin <= 0 || (Cin % 64)) return K5_ERR_ALIGN;
  if ((up_t != 1 && up_t != 2) || (up_s != 1 && up_s != 2)) return K5_ERR_ARG;
  if ((st_t != 1 && st_t != 2) || (st_s != 1 && st_s != 2) || ((st_t > 1 || st_s > 1) && (up_t > 1 || up_s > 1))) return K5_ERR_ARG;
  const bool strided = st_t > 1 || st_s > 1;
  static const int force = getenv("K5_CONV_V1") ? atoi(getenv("K5_CONV_V1")) : 0;   // A/B: 1 = always the 128 x 128 kernel below
  if (force != 1 && !strided) {
    const int r = k5_launch_conv3d_w4(X, W, bias, out, Ts, Hs, Ws, Cin, Cout, up_t, up_s, ldc, resid, ldr, nullptr, stream);
    if (r != K5_ERR_UNSUPPORTED) return r;
  }
  ConvP p;
  p.X = (const bf16_t*)X; p.W = (const bf16_t*)W; p.C = (bf16_t*)out; p.bias = bias; p.resid = (const bf16_t*)resid;
  p.Ts = Ts; p.Hs = Hs; p.Ws = Ws;
  p.To = up_t == 2 ? 2 * Ts - 1 : Ts; p.Ho = up_s * Hs; p.Wo = up_s * Ws;
  p.Hu = p.Ho; p.Wu = p.Wo; p.st_t = st_t; p.st_s = st_s;
  if (strided) { p.To = (Ts - 1) / st_t + 1; p.Ho = (Hs - 1) / st_s + 1; p.Wo = (Ws - 1) / st_s + 1; }
  p.up_t = up_t; p.up_s = up_s; p.Cin = Cin; p.Cout = Cout; p.ldc = ldc; p.ldr = ldr;
  const long long M = (long long)p.To * p.Ho * p.Wo;
  if (M > 0x7fffffffLL) return K5_ERR_UNSUPPORTED;
  p.M = (int)M;
  static const bool out3_ok = !(getenv("K5_CONV_OUT3") && atoi(getenv("K5_CONV_OUT3")) == 0);   // A/B switch for benchmarking
  if (out3_ok && force != 1 && Cout == 3 && (Cin == 64 || Cin == 128) && !strided && up_t == 1 && up_s == 1 && !resid) {
    const long long nwg = (long long)p.To * ((p.Ho + 7) / 8) * ((p.Wo + 31) / 32);
    if (nwg > 0x7fffffffLL) return K5_ERR_UNSUPPORTED;
    if (Cin == 128) hipLaunchKernelGGL(conv3d_out3_kernel<4>, dim3((unsigned)nwg), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(conv3d_out3_kernel<2>, dim3((unsigned)nwg), dim3(256), 0, stream, p);
    k5_conv3d_set_last_kind(K5_CONV_KIND_OUT3);
    return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
  }
  p.tiles_m = (p.M + BM - 1) / BM; p.tiles_n = (Cout + BN - 1) / BN;
  const dim3 grid(p.tiles_m * p.tiles_n), block(256);
  k5_conv3d_set_last_kind(K5_CONV_KIND_TILE128);
  if (resid) hipLaunchKernelGGL(conv3d_kernel<true>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(conv3d_kernel<false>, grid, block, 0, stream, p);
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}
