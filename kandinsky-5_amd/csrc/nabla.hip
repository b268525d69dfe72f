// nabla.hip — NABLA adaptive block-sparse attention map for gfx950 (device-side, no host sync).
//
// Replaces nablaT_v2 (kandinsky/models/utils.py:136-163) + fast_sta_nabla (:108-133):
//   1. block_mean_kernel   qa, ka = mean over each 64-token block (tokens are in fractal order, so a block is one
//                          8x8 spatial tile of one frame), rounded to bf16 like the reference's bf16 `.mean(-2)`;
//   2. nabla_select_kernel per (head, query block): logits = bf16(qa . ka_j) / 8, softmax in fp32, keep the
//                          smallest set of blocks whose probability mass is >= P  ==  drop the ascending-sorted
//                          prefix whose cumulative sum stays below 1-P (`cvals >= 1 - thr`), OR the sliding-tile
//                          window |dt|<=wT/2, |dh|<=wH/2, |dw|<=wW/2.  No sort: the cut value is found by bisection on
//                          the fp32 bit pattern (monotone for non-negative floats); entries EQUAL to the cut value are
//                          ranked by index like a stable ascending sort.  Output: one bit per (query block, kv block).
//   3. nabla_union_kernel  the attention workgroup covers 4 query blocks: OR their rows, compact the kv-block ids and
//                          attach a 4-bit membership mask (which of the 4 query blocks wants that kv block).
// The sparse attention kernel itself is attn_fwd.hip (SPARSE variant).
#include "k5_common.h"
#include "k5_kernels.h"

namespace {

// x [N][ld] bf16 holding `heads` heads of 64 -> mean [heads][nb][64] bf16
__global__ __launch_bounds__(256) void block_mean_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int heads, int nb, int ld) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < heads * 8; c += 256) {
    const bf16_t* p = x + (size_t)b * 64 * ld + 8 * c;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < 64; ++t) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(p + (size_t)t * ld);
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[2 * j] += __uint_as_float(raw[j] << 16); acc[2 * j + 1] += __uint_as_float(raw[j] & 0xffff0000u); }
    }
    const int h = c >> 3, dc = c & 7;
    u32x4 pk = {pack_bf16x2(acc[0] * (1.f / 64), acc[1] * (1.f / 64)), pack_bf16x2(acc[2] * (1.f / 64), acc[3] * (1.f / 64)),
                pack_bf16x2(acc[4] * (1.f / 64), acc[5] * (1.f / 64)), pack_bf16x2(acc[6] * (1.f / 64), acc[7] * (1.f / 64))};
    *reinterpret_cast<u32x4*>(out + ((size_t)h * nb + b) * 64 + 8 * dc) = pk;
  }
}

struct SelP {
  const bf16_t* qa; const bf16_t* ka;   // [H][nb][64]
  unsigned long long* bits;             // [H][nb][nw]
  int* kv_nb;                           // [H][nb] kept blocks per row (diagnostics / density)
  int H, nb, nw, T, Hb, Wb, wT, wH, wW;
  int nqb, qb0;                         // query blocks handled here: global blocks [qb0, qb0 + nqb) (sequence parallel: the rank's rows)
  float target;                         // 1 - P
};

constexpr int SEL_WAVES = 4;
constexpr int SEL_MAXNB = 4096;

__global__ __launch_bounds__(256) void nabla_select_kernel(SelP p) {
  extern __shared__ float srow[];   // SEL_WAVES rows of nb floats
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * SEL_WAVES + wave;      // (h, i)
  if (row >= p.H * p.nqb) return;
  const int h = row / p.nqb, il = row % p.nqb, i = p.qb0 + il;   // il: local query block, i: its global block index
  float* pr = srow + wave * p.nb;
  // query block mean (broadcast loads)
  float qa[64];
  {
    const bf16_t* q = p.qa + ((size_t)h * p.nqb + il) * 64;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(q + 8 * c);
#pragma unroll
      for (int j = 0; j < 4; ++j) { qa[8 * c + 2 * j] = __uint_as_float(raw[j] << 16); qa[8 * c + 2 * j + 1] = __uint_as_float(raw[j] & 0xffff0000u); }
    }
  }
  // logits: bf16(qa . ka_j) / sqrt(64)   (bf16 matmul output, then an exact /8)
  float mx = -3.0e38f;
  for (int j = lane; j < p.nb; j += 64) {
    const bf16_t* k = p.ka + ((size_t)h * p.nb + j) * 64;
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(k + 8 * c);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        d = fmaf(qa[8 * c + 2 * jj], __uint_as_float(raw[jj] << 16), d);
        d = fmaf(qa[8 * c + 2 * jj + 1], __uint_as_float(raw[jj] & 0xffff0000u), d);
      }
    }
    const float lg = bf_round(d) * 0.125f;
    pr[j] = lg;
    mx = fmaxf(mx, lg);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < p.nb; j += 64) { const float e = expf(pr[j] - mx); pr[j] = e; sum += e; }
  sum = wave_sum(sum);
  for (int j = lane; j < p.nb; j += 64) pr[j] = pr[j] / sum;
  // smallest value v* with  sum_{p <= v*} p  >= target  (bisection over the bit pattern of non-negative floats)
  unsigned lo = 0u, hi = 0x7f800000u;   // invariant: g(lo-1) < target <= g(hi)
  while (lo < hi) {
    const unsigned mid = lo + ((hi - lo) >> 1);
    float g = 0.f;
    for (int j = lane; j < p.nb; j += 64) { const float v = pr[j]; g += (__float_as_uint(v) <= mid) ? v : 0.f; }
    g = wave_sum(g);
    if (g >= p.target) hi = mid; else lo = mid + 1;
  }
  const unsigned vbits = lo;
  const float vstar = __uint_as_float(vbits);
  float base = 0.f;
  for (int j = lane; j < p.nb; j += 64) { const float v = pr[j]; base += (__float_as_uint(v) < vbits) ? v : 0.f; }
  base = wave_sum(base);
  // ties at v*: a stable ascending sort orders them by index; the m-th tie has cumsum base + m*v*
  int m0 = 1;
  if (vstar > 0.f) { const float need = (p.target - base) / vstar; m0 = (int)ceilf(need); if (m0 < 1) m0 = 1; }
  // emit bits, 64 kv blocks per word
  const int ti = i / (p.Hb * p.Wb), hi_ = (i / p.Wb) % p.Hb, wi = i % p.Wb;
  int tie_seen = 0, kept = 0;
  for (int c = 0; c < p.nw; ++c) {
    const int j = c * 64 + lane;
    bool keep = false, tie = false;
    if (j < p.nb) {
      const unsigned vb = __float_as_uint(pr[j]);
      tie = vb == vbits;
      keep = vb > vbits;
      const int tj = j / (p.Hb * p.Wb), hj = (j / p.Wb) % p.Hb, wj = j % p.Wb;
      keep = keep || (abs(ti - tj) <= p.wT / 2 && abs(hi_ - hj) <= p.wH / 2 && abs(wi - wj) <= p.wW / 2);
    }
    const unsigned long long tmask = __ballot(tie);
    if (tie) {
      const int rank = tie_seen + __popcll(tmask & ((1ull << lane) - 1ull)) + 1;
      keep = keep || rank >= m0;
    }
    tie_seen += __popcll(tmask);
    const unsigned long long w = __ballot(keep);
    kept += __popcll(w);
    if (lane == 0) p.bits[((size_t)h * p.nqb + il) * p.nw + c] = w;
  }
  if (lane == 0 && p.kv_nb) p.kv_nb[row] = kept;
}

// per (h, group of 4 query blocks): list[(h*ng+g)*nb + e] = kv_block | membership << 24 ; cnt[h*ng+g]
__global__ __launch_bounds__(256) void nabla_union_kernel(const unsigned long long* __restrict__ bits, int* __restrict__ list,
                                                          int* __restrict__ cnt, int H, int nqb, int nb, int nw, int ng) {
  const int lane = threadIdx.x & 63;
  const int gi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gi >= H * ng) return;
  const int h = gi / ng, g = gi % ng;
  int pos = 0;
  for (int c = 0; c < nw; ++c) {
    unsigned long long w[4], u = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qb = 4 * g + r;
      w[r] = qb < nqb ? bits[((size_t)h * nqb + qb) * nw + c] : 0ull;
      u |= w[r];
    }
    if ((u >> lane) & 1ull) {
      int mem = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) mem |= (int)((w[r] >> lane) & 1ull) << r;
      list[(size_t)gi * nb + pos + __popcll(u & ((1ull << lane) - 1ull))] = (c * 64 + lane) | (mem << 24);
    }
    pos += __popcll(u);
  }
  if (lane == 0) cnt[gi] = pos;
}

__global__ __launch_bounds__(256) void nabla_expand_kernel(const unsigned long long* __restrict__ bits, unsigned char* __restrict__ out,
                                                           int64_t rows, int nb, int nw) {
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < rows * nb; g += (int64_t)gridDim.x * 256) {
    const int j = (int)(g % nb);
    const int64_t r = g / nb;
    out[g] = (unsigned char)((bits[r * nw + (j >> 6)] >> (j & 63)) & 1ull);
  }
}

// sum of the per-row kept-block counts -> one atomic per workgroup (profiling only: realised map density)
__global__ __launch_bounds__(256) void nabla_count_kernel(const int* __restrict__ kv_nb, int rows, unsigned long long* acc) {
  unsigned long long v = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < rows; i += gridDim.x * 256) v += (unsigned long long)kv_nb[i];
  __shared__ unsigned long long part[4];
  float dummy = 0.f; (void)dummy;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
}

}  // namespace

int k5_launch_nabla_count(const void* workspace, int H, int nqb, int nb, unsigned long long* acc, hipStream_t s) {
  if (!workspace || !acc || H <= 0 || nqb <= 0) return K5_ERR_ARG;
  const int* kv_nb;
  k5_nabla_workspace_views(const_cast<void*>(workspace), H, nb, nullptr, &kv_nb, nullptr, nullptr);
  const int rows = H * nqb;
  hipLaunchKernelGGL(nabla_count_kernel, dim3((rows + 4095) / 4096), dim3(256), 0, s, kv_nb, rows, acc);
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}

int k5_launch_nabla_mask_u8(const void* workspace, int H, int nqb, int nb, void* out, hipStream_t s) {
  if (H <= 0 || nqb <= 0 || nqb > nb) return K5_ERR_ARG;
  const unsigned long long* bits;
  k5_nabla_workspace_views(const_cast<void*>(workspace), H, nb, &bits, nullptr, nullptr, nullptr);
  const int64_t rows = (int64_t)H * nqb;
  int64_t blocks = (rows * nb + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(nabla_expand_kernel, dim3((unsigned)blocks), dim3(256), 0, s, bits, (unsigned char*)out, rows, nb, (nb + 63) / 64);
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}

size_t k5_nabla_workspace_bytes(int H, int nb) {
  const size_t nw = (nb + 63) / 64, ng = (nb + 3) / 4;
  return (size_t)2 * H * nb * 64 * 2      // qa, ka
         + (size_t)H * nb * nw * 8         // bits
         + (size_t)H * nb * 4              // kv_nb
         + (size_t)H * ng * nb * 4         // union lists
         + (size_t)H * ng * 4 + 256;       // counts
}

// q: [Nq][ldq] bf16 = the query rows handled here (global 64-token blocks [q_block0, q_block0 + Nq/64)), k: [N][ldk] = all
// keys; both after norm_qk + RoPE, fractal token order.  Fills the workspace (regions sized for Nq == N, rows indexed by
// the local query block) with: qa|ka means, block bitmap, per-row counts, per-workgroup union lists.
int k5_launch_nabla_select_rect(const void* q, const void* k, int ldq, int ldk, int H, int Nq, int q_block0, int N, int T, int Hb,
                                int Wb, int wT, int wH, int wW, float P, void* workspace, hipStream_t s) {
  if (H <= 0 || N <= 0 || Nq <= 0 || (N % 64) || (Nq % 64) || T * Hb * Wb * 64 != N) return K5_ERR_ARG;
  if (q_block0 < 0 || q_block0 * 64 + Nq > N) return K5_ERR_ARG;
  if ((ldq & 7) || (ldk & 7)) return K5_ERR_ALIGN;
  const int nb = N / 64, nqb = Nq / 64, nw = (nb + 63) / 64, ng = (nqb + 3) / 4;
  if (nb > SEL_MAXNB) return K5_ERR_UNSUPPORTED;
  char* ws = (char*)workspace;
  bf16_t* qa = (bf16_t*)ws; ws += (size_t)H * nb * 64 * 2;
  bf16_t* ka = (bf16_t*)ws; ws += (size_t)H * nb * 64 * 2;
  unsigned long long* bits = (unsigned long long*)ws; ws += (size_t)H * nb * nw * 8;
  int* kv_nb = (int*)ws; ws += (size_t)H * nb * 4;
  int* list = (int*)ws; ws += (size_t)H * ((nb + 3) / 4) * nb * 4;
  int* cnt = (int*)ws;
  hipLaunchKernelGGL(block_mean_kernel, dim3(nqb), dim3(256), 0, s, (const bf16_t*)q, qa, H, nqb, ldq);
  hipLaunchKernelGGL(block_mean_kernel, dim3(nb), dim3(256), 0, s, (const bf16_t*)k, ka, H, nb, ldk);
  SelP p;
  p.qa = qa; p.ka = ka; p.bits = bits; p.kv_nb = kv_nb; p.H = H; p.nb = nb; p.nw = nw; p.T = T; p.Hb = Hb; p.Wb = Wb;
  p.wT = wT; p.wH = wH; p.wW = wW; p.target = (float)(1.0 - (double)P);
  p.nqb = nqb; p.qb0 = q_block0;
  const int rows = H * nqb;
  hipLaunchKernelGGL(nabla_select_kernel, dim3((rows + SEL_WAVES - 1) / SEL_WAVES), dim3(256), (size_t)SEL_WAVES * nb * 4, s, p);
  hipLaunchKernelGGL(nabla_union_kernel, dim3((H * ng + 3) / 4), dim3(256), 0, s, bits, list, cnt, H, nqb, nb, nw, ng);
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}

int k5_launch_nabla_select(const void* q, const void* k, int ldq, int ldk, int H, int N, int T, int Hb, int Wb, int wT, int wH,
                           int wW, float P, void* workspace, hipStream_t s) {
  return k5_launch_nabla_select_rect(q, k, ldq, ldk, H, N, 0, N, T, Hb, Wb, wT, wH, wW, P, workspace, s);
}

// views into the workspace filled above
void k5_nabla_workspace_views(void* workspace, int H, int nb, const unsigned long long** bits, const int** kv_nb, const int** list,
                              const int** cnt) {
  const size_t nw = (nb + 63) / 64, ng = (nb + 3) / 4;
  char* ws = (char*)workspace + (size_t)2 * H * nb * 64 * 2;
  if (bits) *bits = (const unsigned long long*)ws;
  ws += (size_t)H * nb * nw * 8;
  if (kv_nb) *kv_nb = (const int*)ws;
  ws += (size_t)H * nb * 4;
  if (list) *list = (const int*)ws;
  ws += (size_t)H * ng * nb * 4;
  if (cnt) *cnt = (const int*)ws;
}
