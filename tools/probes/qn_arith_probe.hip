// qn_arith_probe.hip — is the attention kernel's fused query norm (attn_fwd.hip, PRE && QN prologue) the SAME arithmetic as rmsnorm_rope_kernel?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I kandinsky-5_amd/csrc tools/probes/qn_arith_probe.hip -o tools/probes/qn_arith_probe.bin
// Two kernels over the same raw rows [rows][H*64] bf16: A = the standalone pass's per-(row, 8-dim chunk) thread mapping, B = the attention prologue's
// (l15, g) lane mapping with the permlane swaps; the normalised + rotated rows are written out and compared word for word on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
#include "k5_common.h"

__global__ void norm_a(const bf16_t* x, bf16_t* out, const float* w, const float* cosT, const float* sinT, int rows, int H, float* dbg) {
  const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = gi & 7; const long long rh = gi >> 3; const int head = rh % H; const long long row = rh / H;
  if (row >= rows) return;
  const u32x4 raw = *reinterpret_cast<const u32x4*>(x + (size_t)row * H * 64 + head * 64 + 8 * c);
  float v[8], sq = 0.f;
  for (int j = 0; j < 4; ++j) {
    v[2 * j] = __uint_as_float(raw[j] << 16); v[2 * j + 1] = __uint_as_float(raw[j] & 0xffff0000u);
    sq = __fadd_rn(sq, fmaf(v[2 * j + 1], v[2 * j + 1], __fmul_rn(v[2 * j], v[2 * j])));
  }
  sq = __fadd_rn(sq, __shfl_xor(sq, 1, 64)); sq = __fadd_rn(sq, __shfl_xor(sq, 2, 64)); sq = __fadd_rn(sq, __shfl_xor(sq, 4, 64));
  const float rs = rsqrtf(fmaf(sq, 1.0f / 64.0f, 1.1920928955078125e-07f));
  if (c == 0) { dbg[2 * (row * H + head)] = sq; dbg[2 * (row * H + head) + 1] = rs; }
  float y[8];
  for (int j = 0; j < 8; ++j) y[j] = bf_round(__fmul_rn(__fmul_rn(v[j], rs), w[8 * c + j]));
  const f32x4 cs = *reinterpret_cast<const f32x4*>(cosT + (size_t)row * 32 + 4 * c), sn = *reinterpret_cast<const f32x4*>(sinT + (size_t)row * 32 + 4 * c);
  for (int j = 0; j < 4; ++j) {
    const float x0 = y[2 * j], x1 = y[2 * j + 1];
    y[2 * j] = fmaf(cs[j], x0, -__fmul_rn(sn[j], x1));
    y[2 * j + 1] = fmaf(sn[j], x0, __fmul_rn(cs[j], x1));
  }
  const u32x4 pk = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7])};
  *reinterpret_cast<u32x4*>(out + (size_t)row * H * 64 + head * 64 + 8 * c) = pk;
}

// one wave per (16 rows, head): lane (l15, g) holds dims 32 ks + 8 g .. + 8 of row 16 blk + l15
__global__ void norm_b(const bf16_t* x, bf16_t* out, const float* w, const float* cosT, const float* sinT, int rows, int H, float* dbg) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
  const int wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int head = wv % H, blk = wv / H;
  const int row = min(16 * blk + l15, rows - 1);
  if (16 * blk >= rows) return;
  bf16x8 qf[2];
  for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(x + (size_t)row * H * 64 + head * 64 + 32 * ks + 8 * g);
  f32x4 wa[2], wb[2], cs[2], sn[2];
  for (int ks = 0; ks < 2; ++ks) {
    wa[ks] = *reinterpret_cast<const f32x4*>(w + 32 * ks + 8 * g); wb[ks] = *reinterpret_cast<const f32x4*>(w + 32 * ks + 8 * g + 4);
    cs[ks] = *reinterpret_cast<const f32x4*>(cosT + (size_t)row * 32 + 16 * ks + 4 * g); sn[ks] = *reinterpret_cast<const f32x4*>(sinT + (size_t)row * 32 + 16 * ks + 4 * g);
  }
  float v[16], sq2[2];
  for (int ks = 0; ks < 2; ++ks) {
    const u32x4 w4 = __builtin_bit_cast(u32x4, qf[ks]);
    float sq = 0.f;
    for (int j = 0; j < 4; ++j) {
      v[8 * ks + 2 * j] = __uint_as_float(w4[j] << 16); v[8 * ks + 2 * j + 1] = __uint_as_float(w4[j] & 0xffff0000u);
      sq = __fadd_rn(sq, fmaf(v[8 * ks + 2 * j + 1], v[8 * ks + 2 * j + 1], __fmul_rn(v[8 * ks + 2 * j], v[8 * ks + 2 * j])));
    }
    const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(sq), __float_as_uint(sq), false, false);
    sq = __fadd_rn(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
    const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(sq), __float_as_uint(sq), false, false);
    sq2[ks] = __fadd_rn(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
  }
  const float ss = __fadd_rn(sq2[0], sq2[1]);
  const float rs = rsqrtf(fmaf(ss, 1.0f / 64.0f, 1.1920928955078125e-07f));
  for (int ks = 0; ks < 2; ++ks) {
    float y[8];
    for (int j = 0; j < 8; ++j) y[j] = bf_round(__fmul_rn(__fmul_rn(v[8 * ks + j], rs), j < 4 ? wa[ks][j] : wb[ks][j - 4]));
    for (int j = 0; j < 4; ++j) {
      const float x0 = y[2 * j], x1 = y[2 * j + 1];
      y[2 * j] = fmaf(cs[ks][j], x0, -__fmul_rn(sn[ks][j], x1));
      y[2 * j + 1] = fmaf(sn[ks][j], x0, __fmul_rn(cs[ks][j], x1));
    }
    const u32x4 pk = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7])};
    if (16 * blk + l15 < rows) *reinterpret_cast<u32x4*>(out + (size_t)row * H * 64 + head * 64 + 32 * ks + 8 * g) = pk;
  }
  if (g == 0 && 16 * blk + l15 < rows) { dbg[2 * ((size_t)row * H + head)] = ss; dbg[2 * ((size_t)row * H + head) + 1] = rs; }
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 33280, H = 4;
  const size_t n = (size_t)rows * H * 64;
  std::vector<uint16_t> hx(n);
  srand(7);
  for (size_t i = 0; i < n; ++i) { float f = ((rand() % 20001) - 10000) * 2.5e-4f * ((rand() & 7) + 1); uint32_t u; memcpy(&u, &f, 4); hx[i] = (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
  std::vector<float> hw(64), hc((size_t)rows * 32), hs((size_t)rows * 32);
  for (int i = 0; i < 64; ++i) hw[i] = 0.85f + 0.3f * (rand() % 1000) * 1e-3f;
  for (size_t i = 0; i < hc.size(); ++i) { const float a = (rand() % 62832) * 1e-4f; hc[i] = cosf(a); hs[i] = sinf(a); }
  bf16_t *x, *oa, *ob; float *w, *c, *s;
  hipMalloc(&x, n * 2); hipMalloc(&oa, n * 2); hipMalloc(&ob, n * 2); hipMalloc(&w, 256); hipMalloc(&c, hc.size() * 4); hipMalloc(&s, hs.size() * 4);
  hipMemcpy(x, hx.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), 256, hipMemcpyHostToDevice);
  hipMemcpy(c, hc.data(), hc.size() * 4, hipMemcpyHostToDevice); hipMemcpy(s, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
  hipMemset(oa, 0, n * 2); hipMemset(ob, 0, n * 2);
  const long long ta = (long long)rows * H * 8;
  float *da, *db; hipMalloc(&da, (size_t)rows * H * 8); hipMalloc(&db, (size_t)rows * H * 8);
  norm_a<<<(unsigned)((ta + 255) / 256), 256>>>(x, oa, w, c, s, rows, H, da);
  const int waves = ((rows + 15) / 16) * H;
  norm_b<<<(waves + 3) / 4, 256>>>(x, ob, w, c, s, rows, H, db);
  hipDeviceSynchronize();
  std::vector<uint16_t> ha(n), hb(n);
  hipMemcpy(ha.data(), oa, n * 2, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), ob, n * 2, hipMemcpyDeviceToHost);
  size_t bad = 0, badrows = 0;
  for (size_t r = 0; r < (size_t)rows * H; ++r) { size_t b = 0; for (int d = 0; d < 64; ++d) b += ha[r * 64 + d] != hb[r * 64 + d]; bad += b; badrows += b != 0; }
  std::vector<float> fa((size_t)rows * H * 2), fb((size_t)rows * H * 2);
  hipMemcpy(fa.data(), da, fa.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(fb.data(), db, fb.size() * 4, hipMemcpyDeviceToHost);
  size_t dss = 0, drs = 0, both = 0;
  for (size_t r = 0; r < (size_t)rows * H; ++r) { const bool a = memcmp(&fa[2 * r], &fb[2 * r], 4) != 0, b = memcmp(&fa[2 * r + 1], &fb[2 * r + 1], 4) != 0; dss += a; drs += b; size_t bb = 0; for (int d = 0; d < 64; ++d) bb += ha[r * 64 + d] != hb[r * 64 + d]; both += (bb != 0) && b; }
  printf("sum of squares differs in %zu (row, head) pairs, 1/rms in %zu; pairs with a differing element AND a differing 1/rms: %zu\n", dss, drs, both);
  printf("rows %d x %d heads: %zu of %zu elements differ, in %zu (row, head) pairs  [%s]\n", rows, H, bad, n, badrows, hipGetErrorString(hipGetLastError()));
  return bad != 0;
}
