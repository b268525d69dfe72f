// vae_ops.hip — HBM-bound kernels of the HunyuanVideo VAE decoder on channels-last bf16 activations [M][C].
//   GroupNorm(32, eps 1e-6) (+SiLU)   vae.py:246-263, 351-355 (attention group_norm), 672-673
//       two-level statistics: per-block fp32 partial sums -> double-precision combine -> mean / rstd per group,
//       then one fused normalise*gamma+beta(+SiLU) pass writing bf16 (the fp32 GroupNorm output of autocast is
//       only ever consumed by a bf16 conv / linear).
//   frame-causal softmax              prepare_causal_attention_mask vae.py:110-122 + diffusers Attention softmax
//   layout converters                 (C,T,H,W) fp32 latent -> [M][Cpad] bf16 ; [M][3] bf16 -> (3,T,H,W) bf16
//   blend_t / blend_v / blend_h       vae.py:908-936 (eager bf16 arithmetic: every op rounds)
#include "k5_common.h"
#include "k5_kernels.h"

namespace {

constexpr int GN_ROWS = 512;  // rows per statistics block

// partial[blk][g] = (sum, sumsq) over GN_ROWS rows x (C/G) channels.  Deterministic: per-thread partials go to LDS
// and each group is summed by one thread in a fixed order (no float atomics -> bit-reproducible decodes).
__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ partial, int M,
                                                         int C, int ldx, int G) {
  __shared__ float ps[256][4];  // per thread: (sum, sumsq) for its first group, (sum, sumsq) for its second group
  const int tid = threadIdx.x;
  const int nch = C >> 3, cg = C / G;       // 16-B chunks per row; channels per group
  const int r0 = blockIdx.x * GN_ROWS, r1 = min(r0 + GN_ROWS, M);
  // thread -> fixed chunk column (tid % nch), strides over rows: its 8 channels lie in <= 2 groups
  const int rows_par = 256 / nch > 0 ? 256 / nch : 1;
  const int ch = tid % nch, rsub = tid / nch;
  float s[2] = {0.f, 0.f}, q[2] = {0.f, 0.f};
  if (rsub < rows_par) {
    // a chunk's 8 channels lie in ONE group (cg >= 8) or are the two 4-channel groups of cg == 4: sum the halves separately
    float slo = 0.f, shi = 0.f, qlo = 0.f, qhi = 0.f;
    for (int rb = r0 + rsub; rb < r1; rb += 4 * rows_par) {   // four independent loads in flight; summation order unchanged
      u32x4 raw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = rb + u * rows_par;
        raw[u] = r < r1 ? *reinterpret_cast<const u32x4*>(x + (size_t)r * ldx + 8 * ch) : u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = (j & 1) ? __uint_as_float(raw[u][j >> 1] & 0xffff0000u) : __uint_as_float(raw[u][j >> 1] << 16);
          if (j < 4) { slo += v; qlo += v * v; } else { shi += v; qhi += v * v; }
        }
    }
    if (cg == 4) { s[0] = slo; q[0] = qlo; s[1] = shi; q[1] = qhi; }
    else { s[0] = slo + shi; q[0] = qlo + qhi; }
  }
  ps[tid][0] = s[0]; ps[tid][1] = q[0]; ps[tid][2] = s[1]; ps[tid][3] = q[1];
  __syncthreads();
  if (tid < G) {
    float ss = 0.f, qq = 0.f;
    const int used = min(256, rows_par * nch);
    for (int t = 0; t < used; ++t) {
      const int tg0 = (8 * (t % nch)) / cg;
      if (tg0 == tid) { ss += ps[t][0]; qq += ps[t][1]; }
      else if (cg < 8 && tg0 + 1 == tid) { ss += ps[t][2]; qq += ps[t][3]; }
    }
    partial[((size_t)blockIdx.x * G + tid) * 2] = ss;
    partial[((size_t)blockIdx.x * G + tid) * 2 + 1] = qq;
  }
}

// one 256-thread block per group: strided double-precision sums over the per-block partials, LDS tree
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stats, int nblk,
                                                          int G, double count, float eps) {
  __shared__ double rs[256], rq[256];
  const int g = blockIdx.x, tid = threadIdx.x;
  double s = 0.0, q = 0.0;
  for (int b = tid; b < nblk; b += 256) { s += partial[((size_t)b * G + g) * 2]; q += partial[((size_t)b * G + g) * 2 + 1]; }
  rs[tid] = s; rq[tid] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { rs[tid] += rs[tid + o]; rq[tid] += rq[tid + o]; }
    __syncthreads();
  }
  if (tid == 0) {
    const double mean = rs[0] / count;
    double var = rq[0] / count - mean * mean;
    if (var < 0) var = 0;
    stats[2 * g] = (float)mean;
    stats[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// statistics from the producing conv's epilogue: quads [nblk][C / 4][2] = (sum, sum of squares) per 128 rows and 4 consecutive
// channels; group g owns quads g cg / 4 .. (g + 1) cg / 4 - 1 of every block.  Same double-precision tree as above.
constexpr int GNQ_SPLIT = 16;   // workgroups per group in the first level
// level 1: workgroup (g, s) folds slice s of the group's nblk * cg/4 quad entries in double precision -> part[(g * GNQ_SPLIT + s) * 2]
__global__ __launch_bounds__(256) void gn_fold_quads_kernel(const float* __restrict__ quads, double* __restrict__ part, int nblk, int C, int cg) {
  __shared__ double rs[256], rq[256];
  const int g = blockIdx.x / GNQ_SPLIT, sl = blockIdx.x % GNQ_SPLIT, tid = threadIdx.x, qpg = cg >> 2, nq = C >> 2;
  const int n = nblk * qpg;
  const int e_lo = (int)((long long)n * sl / GNQ_SPLIT), e_hi = (int)((long long)n * (sl + 1) / GNQ_SPLIT);
  double s = 0.0, q = 0.0;
  for (int e0 = e_lo + tid; e0 < e_hi; e0 += 1024) {   // 4 independent 8-B loads in flight per thread
    f32x2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + 256 * u;
      v[u] = e < e_hi ? *reinterpret_cast<const f32x2*>(quads + ((size_t)(e / qpg) * nq + g * qpg + (e % qpg)) * 2) : f32x2{0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { s += v[u][0]; q += v[u][1]; }
  }
  rs[tid] = s; rq[tid] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { rs[tid] += rs[tid + o]; rq[tid] += rq[tid + o]; }
    __syncthreads();
  }
  if (tid == 0) { part[(size_t)blockIdx.x * 2] = rs[0]; part[(size_t)blockIdx.x * 2 + 1] = rq[0]; }
}
// level 2: one thread per group adds its GNQ_SPLIT partial pairs in order -> (mean, rstd)
__global__ void gn_finalize_quads_kernel(const double* __restrict__ part, float* __restrict__ stats, int G, double count, float eps) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  double s = 0.0, q = 0.0;
  for (int i = 0; i < GNQ_SPLIT; ++i) { s += part[((size_t)g * GNQ_SPLIT + i) * 2]; q += part[((size_t)g * GNQ_SPLIT + i) * 2 + 1]; }
  const double mean = s / count;
  double var = q / count - mean * mean;
  if (var < 0) var = 0;
  stats[2 * g] = (float)mean;
  stats[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// thread -> fixed 16-B chunk column (tid % nch; nch a power of two <= 256), rows strided: the eight channels' affine
// y = v * (rstd gamma) + (beta - mean rstd gamma) is folded once per thread, SiLU = y * rcp(1 + exp2(-y log2 e)) on the
// hardware exp2 / rcp (the result is rounded to bf16).  One 16-B load, ~50 VALU, one 16-B store per chunk: HBM-bound.
template <bool SILU, int U = 4>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       bf16_t* __restrict__ out, int M, int nch_sh, int cg_sh, int ldx, int ldo,
                                                       int rows_per_block) {
  const int nch = 1 << nch_sh, ch = threadIdx.x & (nch - 1), rsub = threadIdx.x >> nch_sh, rows_par = 256 >> nch_sh;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 8 * ch + j, g = c >> cg_sh;
    sc[j] = stats[2 * g + 1] * gamma[c];
    sh[j] = beta[c] - stats[2 * g] * sc[j];
  }
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, M);
  for (int rb = r0 + rsub; rb < r1; rb += U * rows_par) {   // U independent 16-B loads in flight per thread
    u32x4 raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = rb + u * rows_par;
      if (r < r1) raw[u] = *reinterpret_cast<const u32x4*>(x + (size_t)r * ldx + 8 * ch);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = rb + u * rows_par;
      if (r >= r1) break;
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = (j & 1) ? __uint_as_float(raw[u][j >> 1] & 0xffff0000u) : __uint_as_float(raw[u][j >> 1] << 16);
        float y = fmaf(v, sc[j], sh[j]);
        if (SILU) y = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * y));
        o[j] = y;
      }
      u32x4 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
      *reinterpret_cast<u32x4*>(out + (size_t)r * ldo + 8 * ch) = pk;
    }
  }
}

// scores fp32 [S][lds] (already scaled) -> P bf16 [S][ldp]; row i attends columns j < ((i / hw) + 1) * hw
__global__ __launch_bounds__(256) void causal_softmax_kernel(const float* __restrict__ sc, bf16_t* __restrict__ P, int S, int hw,
                                                             int lds, int ldp) {
  __shared__ float red[8];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int valid = min(S, (row / hw + 1) * hw);
  const float* s = sc + (size_t)row * lds;
  float m = -3.0e38f;
  for (int j = tid; j < valid; j += 256) m = fmaxf(m, s[j]);
  m = wave_max(m);
  if (lane == 0) red[wv] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < valid; j += 256) sum += expf(s[j] - m);
  sum = wave_sum(sum);
  if (lane == 0) red[wv] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  bf16_t* p = P + (size_t)row * ldp;
  for (int j = tid; j < ldp; j += 256) p[j] = f2bf(j < valid ? expf(s[j] - m) * inv : 0.f);
}

// the same, one read of the row: 1024 threads hold the row's valid scores in registers (NV per thread, NV * 1024 >= valid)
template <int NV>
__global__ __launch_bounds__(1024) void causal_softmax_reg_kernel(const float* __restrict__ sc, bf16_t* __restrict__ P, int S, int hw,
                                                                  int lds, int ldp) {
  __shared__ float red[16];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int valid = min(S, (row / hw + 1) * hw);
  const float* s = sc + (size_t)row * lds;
  float v[NV];
  float m = -3.0e38f;
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int j = tid + 1024 * u;
    v[u] = j < valid ? s[j] : -3.0e38f;
    m = fmaxf(m, v[u]);
  }
  m = wave_max(m);
  if (lane == 0) red[wv] = m;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) m = fmaxf(m, red[w]);
  __syncthreads();
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < NV; ++u) { v[u] = tid + 1024 * u < valid ? expf(v[u] - m) : 0.f; sum += v[u]; }
  sum = wave_sum(sum);
  if (lane == 0) red[wv] = sum;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) tot += red[w];
  const float inv = 1.0f / tot;
  bf16_t* p = P + (size_t)row * ldp;
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int j = tid + 1024 * u;
    if (j < ldp) p[j] = f2bf(v[u] * inv);
  }
  for (int j = tid + 1024 * NV; j < ldp; j += 1024) p[j] = f2bf(0.f);
}

// z (C,T,H,W) fp32 -> [M][Cpad] bf16, channels >= C zero
__global__ __launch_bounds__(256) void nchw_to_mc_kernel(const float* __restrict__ z, bf16_t* __restrict__ out, int C, int64_t M, int Cpad, int64_t cstride) {
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < M * Cpad; g += (int64_t)gridDim.x * 256) {
    const int c = (int)(g % Cpad);
    const int64_t m = g / Cpad;
    out[g] = f2bf(c < C ? z[(int64_t)c * cstride + m] : 0.f);   // cstride >= M: a temporal slice of a longer latent, read in place
  }
}
// [M][ldx] bf16 (first C channels) -> (C, M) bf16
__global__ __launch_bounds__(256) void mc_to_nchw_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int C, int64_t M, int ldx) {
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < M * C; g += (int64_t)gridDim.x * 256) {
    const int64_t m = g % M;
    const int c = (int)(g / M);
    out[g] = x[m * ldx + c];
  }
}

// b[.., y, ..] = bf16( bf16(a[.., La-extent+y, ..] * (1 - y/extent)) + bf16(b[.., y, ..] * (y/extent)) )  along one axis.
// tensors are viewed as [outer][len][inner] (contiguous), a and b may have different lengths along the axis.
__global__ __launch_bounds__(256) void blend_kernel(const bf16_t* __restrict__ a, bf16_t* __restrict__ b, int64_t outer, int la,
                                                    int lb, int64_t inner, int extent) {
  const int64_t total = outer * extent * inner;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
    const int64_t in = g % inner;
    const int y = (int)((g / inner) % extent);
    const int64_t o = g / (inner * extent);
    const float wb = (float)((double)y / (double)extent), wa = (float)(1.0 - (double)y / (double)extent);
    const float av = bf2f(a[(o * la + (la - extent + y)) * inner + in]);
    bf16_t* bp = b + (o * lb + y) * inner + in;
    *bp = f2bf(__fadd_rn(bf_round(__fmul_rn(av, wa)), bf_round(__fmul_rn(bf2f(*bp), wb))));
  }
}

// Round 6 — blend_t + the slice + the concatenation of the temporal tiling loop (vae.py:1144-1204) as ONE pass: views [outer][len][inner] with explicit
// outer strides (a decoded tile minus its first frame, the output video).  dst[o][y] = y < extent ? blend(a[o][la - extent + y], b[o][y]) : b[o][y] for
// y < keep — the arithmetic and rounding points of blend_kernel.  a == nullptr: no cross-fade (the first tile).  16 bytes per thread where inner allows.
__global__ __launch_bounds__(256) void blend_place_kernel(const bf16_t* __restrict__ a, int64_t sa, int la, const bf16_t* __restrict__ b, int64_t sb,
                                                          bf16_t* __restrict__ dst, int64_t sd, int64_t outer, int64_t inner8, int extent, int keep) {
  const int64_t total = outer * keep * inner8;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
    const int64_t in = (g % inner8) * 8;
    const int y = (int)((g / inner8) % keep);
    const int64_t o = g / (inner8 * keep);
    const int64_t inner = inner8 * 8;
    u32x4 bv = *reinterpret_cast<const u32x4*>(b + o * sb + (int64_t)y * inner + in);
    if (a && y < extent) {
      const u32x4 av = *reinterpret_cast<const u32x4*>(a + o * sa + (int64_t)(la - extent + y) * inner + in);
      const float wb = (float)((double)y / (double)extent), wa = (float)(1.0 - (double)y / (double)extent);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a0 = __uint_as_float(av[j] << 16), a1 = __uint_as_float(av[j] & 0xffff0000u);
        const float b0 = __uint_as_float(bv[j] << 16), b1 = __uint_as_float(bv[j] & 0xffff0000u);
        bv[j] = pack_bf16x2(__fadd_rn(bf_round(__fmul_rn(a0, wa)), bf_round(__fmul_rn(b0, wb))), __fadd_rn(bf_round(__fmul_rn(a1, wa)), bf_round(__fmul_rn(b1, wb))));
      }
    }
    *reinterpret_cast<u32x4*>(dst + o * sd + (int64_t)y * inner + in) = bv;
  }
}

// uint8 frames of the pipeline (generation_utils.py:150-151 of the mirror; reference generation_utils.py:222-224): ((x.clamp(-1, 1) + 1) * 127.5).to(uint8) with
// torch's bf16 rounding after every elementwise op and truncation in the conversion — one pass instead of four
__global__ __launch_bounds__(256) void frames_to_uint8_kernel(const bf16_t* __restrict__ x, uint8_t* __restrict__ out, int64_t n8) {
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < n8; g += (int64_t)gridDim.x * 256) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(x + 8 * g);
    uint32_t o[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = (j & 1) ? __uint_as_float(v[j >> 1] & 0xffff0000u) : __uint_as_float(v[j >> 1] << 16);
      f = fminf(fmaxf(f, -1.0f), 1.0f);
      f = bf_round(f + 1.0f);
      f = bf_round(f * 127.5f);
      o[j >> 2] |= (uint32_t)(uint8_t)(int)f << (8 * (j & 3));
    }
    *reinterpret_cast<uint2*>(out + 8 * g) = uint2{o[0], o[1]};
  }
}

inline int grid_for(int64_t n, int cap = 16384) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}
inline int done() { return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP; }

}  // namespace

size_t k5_groupnorm_workspace_bytes(int M, int G) {
  const size_t own = ((size_t)((M + GN_ROWS - 1) / GN_ROWS) * G * 2 + 2 * G) * sizeof(float);   // partials + (mean, rstd) of the two-pass form
  const size_t quads = 2 * 64 * sizeof(float) + (size_t)64 * GNQ_SPLIT * 2 * sizeof(double);      // (mean, rstd) + first-level partials of the fused form
  return own > quads ? own : quads;
}

namespace {
bool gn_shape_ok(int M, int C, int G, int ldx, int ldo) {
  if (M <= 0 || C <= 0 || G <= 0 || G > 64 || (C % G)) return false;
  const int cg = C / G;
  return !((C & 7) || (ldx & 7) || (ldo & 7) || cg < 4 || (cg & (cg - 1)) || (C >> 3) > 256 || ((C >> 3) & ((C >> 3) - 1)));
}
int gn_apply(const void* x, const float* stats, const float* gamma, const float* beta, void* out, int M, int C, int G, int silu, int ldx,
             int ldo, hipStream_t s) {
  const int nch = C >> 3, cg = C / G;   // chunk columns per row: a power of two (C = 128 / 256 / 512 and the tiny test widths)
  int nch_sh = 0, cg_sh = 0;
  while ((1 << nch_sh) < nch) ++nch_sh;
  while ((1 << cg_sh) < cg) ++cg_sh;
  const int rows_par = 256 >> nch_sh, rows_per_block = 32 * rows_par;   // 32 chunks per thread
  const int nb = (M + rows_per_block - 1) / rows_per_block;
  static const int unroll = getenv("K5_GN_UNROLL") ? atoi(getenv("K5_GN_UNROLL")) : 4;   // A/B: 16-B loads in flight per thread
  if (silu && unroll == 8) hipLaunchKernelGGL((gn_apply_kernel<true, 8>), dim3(nb), dim3(256), 0, s, (const bf16_t*)x, stats, gamma, beta, (bf16_t*)out, M,
                                              nch_sh, cg_sh, ldx, ldo, rows_per_block);
  else if (silu) hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(nb), dim3(256), 0, s, (const bf16_t*)x, stats, gamma, beta, (bf16_t*)out, M,
                               nch_sh, cg_sh, ldx, ldo, rows_per_block);
  else hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(nb), dim3(256), 0, s, (const bf16_t*)x, stats, gamma, beta, (bf16_t*)out, M,
                          nch_sh, cg_sh, ldx, ldo, rows_per_block);
  return done();
}
}  // namespace

int k5_launch_groupnorm_bf16(const void* x, const float* gamma, const float* beta, void* out, int M, int C, int G, float eps,
                             int silu, int ldx, int ldo, void* workspace, hipStream_t s) {
  if (M <= 0 || C <= 0 || G <= 0 || G > 64 || (C % G)) return K5_ERR_ARG;
  if (!gn_shape_ok(M, C, G, ldx, ldo)) return K5_ERR_UNSUPPORTED;
  const int cg = C / G;
  const int nblk = (M + GN_ROWS - 1) / GN_ROWS;
  float* partial = (float*)workspace;
  float* stats = partial + (size_t)nblk * G * 2;
  hipLaunchKernelGGL(gn_partial_kernel, dim3(nblk), dim3(256), 0, s, (const bf16_t*)x, partial, M, C, ldx, G);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(G), dim3(256), 0, s, partial, stats, nblk, G, (double)M * cg, eps);
  return gn_apply(x, stats, gamma, beta, out, M, C, G, silu, ldx, ldo, s);
}

int k5_launch_groupnorm_bf16_quads(const void* x, const float* gamma, const float* beta, void* out, int M, int C, int G, float eps,
                                   int silu, int ldx, int ldo, const float* quad_stats, int nblk, float* stats_ws, hipStream_t s) {
  if (M <= 0 || C <= 0 || G <= 0 || G > 64 || (C % G) || !quad_stats || !stats_ws || nblk <= 0) return K5_ERR_ARG;
  if (!gn_shape_ok(M, C, G, ldx, ldo)) return K5_ERR_UNSUPPORTED;
  const int cg = C / G;
  // stats_ws: [2 G floats (mean, rstd)] [G * GNQ_SPLIT * 2 doubles]  (k5_groupnorm_workspace_bytes reserves both)
  double* part = reinterpret_cast<double*>(stats_ws + 2 * 64);
  hipLaunchKernelGGL(gn_fold_quads_kernel, dim3(G * GNQ_SPLIT), dim3(256), 0, s, quad_stats, part, nblk, C, cg);
  hipLaunchKernelGGL(gn_finalize_quads_kernel, dim3(1), dim3(64), 0, s, part, stats_ws, G, (double)M * cg, eps);
  return gn_apply(x, stats_ws, gamma, beta, out, M, C, G, silu, ldx, ldo, s);
}

int k5_launch_causal_softmax(const float* scores, void* P, int S, int hw, int lds, int ldp, hipStream_t s) {
  if (S <= 0 || hw <= 0 || ldp < S) return K5_ERR_ARG;
  // rows of up to 32 768 scores stay in registers (one read of the fp32 scores instead of three)
  if (S > 4096 && S <= 8192) hipLaunchKernelGGL(causal_softmax_reg_kernel<8>, dim3(S), dim3(1024), 0, s, scores, (bf16_t*)P, S, hw, lds, ldp);
  else if (S > 8192 && S <= 16384) hipLaunchKernelGGL(causal_softmax_reg_kernel<16>, dim3(S), dim3(1024), 0, s, scores, (bf16_t*)P, S, hw, lds, ldp);
  else if (S > 16384 && S <= 32768) hipLaunchKernelGGL(causal_softmax_reg_kernel<32>, dim3(S), dim3(1024), 0, s, scores, (bf16_t*)P, S, hw, lds, ldp);
  else hipLaunchKernelGGL(causal_softmax_kernel, dim3(S), dim3(256), 0, s, scores, (bf16_t*)P, S, hw, lds, ldp);
  return done();
}

int k5_launch_nchw_to_mc(const float* z, void* out, int C, int64_t M, int Cpad, hipStream_t s, int64_t cstride) {
  if (C <= 0 || M <= 0 || Cpad < C || (cstride != 0 && cstride < M)) return K5_ERR_ARG;
  hipLaunchKernelGGL(nchw_to_mc_kernel, dim3(grid_for(M * Cpad)), dim3(256), 0, s, z, (bf16_t*)out, C, M, Cpad, cstride ? cstride : M);
  return done();
}

int k5_launch_blend_place_bf16(const void* a, int64_t a_stride, int len_a, const void* b, int64_t b_stride, void* dst, int64_t dst_stride, int64_t outer,
                               int64_t inner, int extent, int keep, hipStream_t s) {
  if (!b || !dst || outer <= 0 || inner <= 0 || keep <= 0 || extent < 0 || (inner & 7)) return K5_ERR_ARG;
  if (a && (extent > len_a || extent > keep)) return K5_ERR_ARG;
  if (((uintptr_t)b | (uintptr_t)dst | (uintptr_t)a) & 15 || ((b_stride | dst_stride | a_stride) & 7)) return K5_ERR_ALIGN;
  hipLaunchKernelGGL(blend_place_kernel, dim3(grid_for(outer * keep * (inner / 8))), dim3(256), 0, s, (const bf16_t*)a, a_stride, len_a, (const bf16_t*)b, b_stride,
                     (bf16_t*)dst, dst_stride, outer, inner / 8, a ? extent : 0, keep);
  return done();
}

int k5_launch_frames_to_uint8(const void* x, void* out, int64_t n, hipStream_t s) {
  if (!x || !out || n <= 0 || (n & 7)) return K5_ERR_ARG;
  hipLaunchKernelGGL(frames_to_uint8_kernel, dim3(grid_for(n / 8)), dim3(256), 0, s, (const bf16_t*)x, (uint8_t*)out, n / 8);
  return done();
}

int k5_launch_mc_to_nchw(const void* x, void* out, int C, int64_t M, int ldx, hipStream_t s) {
  if (C <= 0 || M <= 0) return K5_ERR_ARG;
  hipLaunchKernelGGL(mc_to_nchw_kernel, dim3(grid_for(M * C)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out, C, M, ldx);
  return done();
}

int k5_launch_blend_bf16(const void* a, void* b, int64_t outer, int len_a, int len_b, int64_t inner, int extent, hipStream_t s) {
  if (extent <= 0) return K5_OK;
  if (extent > len_a || extent > len_b) return K5_ERR_ARG;
  hipLaunchKernelGGL(blend_kernel, dim3(grid_for(outer * extent * inner)), dim3(256), 0, s, (const bf16_t*)a, (bf16_t*)b, outer,
                     len_a, len_b, inner, extent);
  return done();
}
