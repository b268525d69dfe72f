// Is v_dot2c_f32_bf16 an exact fp32 accumulate of exact bf16 products?  (It is not what nabla_select needs if not.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* a, const unsigned* b, float* o_dot, float* o_fma, int n) {
  float d = 0.f, f = 0.f;
  unsigned xs[32], ys[32];
  for (int i = 0; i < 32; ++i) { xs[i] = a[threadIdx.x * n + i]; ys[i] = b[threadIdx.x * n + i]; }
  // a chain of 32 DEPENDENT dot2c back to back (operands already in registers), as the nabla kernel issues them
#pragma unroll
  for (int i = 0; i < 32; ++i) d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, xs[i]), __builtin_bit_cast(bf16x2, ys[i]), d, false);
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    f = fmaf(__uint_as_float(xs[i] << 16), __uint_as_float(ys[i] << 16), f);
    f = fmaf(__uint_as_float(xs[i] & 0xffff0000u), __uint_as_float(ys[i] & 0xffff0000u), f);
  }
  o_dot[threadIdx.x] = d; o_fma[threadIdx.x] = f;
}
static unsigned short bf(float v) { unsigned u; memcpy(&u, &v, 4); return (unsigned short)(u >> 16); }
int main() {
  const int n = 32, T = 64;
  unsigned ha[T * n], hb[T * n];
  double ref[T];
  srand(1);
  for (int t = 0; t < T; ++t) {
    ref[t] = 0;
    for (int i = 0; i < n; ++i) {
      float v[4];
      for (int j = 0; j < 4; ++j) v[j] = t < 32 ? (float)((rand() % 7) - 3) : (float)((rand() % 2001) - 1000) / 256.0f;   // ints, then 1/256 steps
      unsigned short q[4];
      for (int j = 0; j < 4; ++j) q[j] = bf(v[j]);
      float r[4];
      for (int j = 0; j < 4; ++j) { unsigned u = (unsigned)q[j] << 16; memcpy(&r[j], &u, 4); }
      ha[t * n + i] = q[0] | ((unsigned)q[1] << 16); hb[t * n + i] = q[2] | ((unsigned)q[3] << 16);
      ref[t] += (double)r[0] * r[2] + (double)r[1] * r[3];
    }
  }
  unsigned *da, *db; float *od, *of;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&od, T * 4); hipMalloc(&of, T * 4);
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  k<<<1, T>>>(da, db, od, of, n);
  float hd[T], hf[T];
  hipMemcpy(hd, od, T * 4, hipMemcpyDeviceToHost); hipMemcpy(hf, of, T * 4, hipMemcpyDeviceToHost);
  double ed = 0, ef = 0;
  for (int t = 0; t < T; ++t) { ed = fmax(ed, fabs(hd[t] - ref[t]) / fmax(1.0, fabs(ref[t]))); ef = fmax(ef, fabs(hf[t] - ref[t]) / fmax(1.0, fabs(ref[t]))); }
  printf("max rel err vs fp64: v_dot2c_f32_bf16 %.3e   fmaf chain %.3e\n", ed, ef);
  for (int t = 0; t < 3; ++t) printf("  ints  t=%d ref %.1f dot2 %.6f fma %.6f\n", t, ref[t], hd[t], hf[t]);
  for (int t = 32; t < 35; ++t) printf("  fracs t=%d ref %.6f dot2 %.6f fma %.6f\n", t, ref[t], hd[t], hf[t]);
  return 0;
}
