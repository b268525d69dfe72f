// vae_attn.hip — the VAE mid block's attention (one head of dimension C = 512, frame-causal) without materialising the scores.
//
// Replaces, for C = 512, the three launches of vae_engine.hip's mid_attention (fp32 scores GEMM -> causal softmax -> P V GEMM):
// diffusers `Attention` called from HunyuanVideoMidBlock3D (kandinsky/models/vae.py:341-362) with the mask of
// prepare_causal_attention_mask (vae.py:110-122): query token i (frame i / hw) attends keys j < (i / hw + 1) * hw.
// At S = 30 720 tokens (a (5,64,96) latent tile) the scores were 3.8 GB of fp32 and the probabilities 1.9 GB of bf16 per call.
//
// A workgroup = 4 waves x 16 queries = 64 queries; a step = 32 keys.  MFMA 16x16x32 in the transposed formulation of
// attn_fwd.hip: S^T = K Q^T (A = 16 keys x 32 d from LDS, B = the wave's Q fragments, 16 k-steps over d = 512), the key rows of
// the two 16-key tiles interleaved (row i of tile kt <-> key 8 (i >> 2) + 4 kt + (i & 3)) so that a lane's eight accumulators are
// eight CONSECUTIVE keys of its query = exactly the B operand of O^T += V^T P^T (A = 16 d-rows x 32 keys from LDS, 32 d-tiles).
// 64 MFMAs per wave and step against 64 ds_read_b128, and 64 KB of K / V^T per step and workgroup through L2 -> LDS: the kernel is
// bound by that traffic (measured 2.67 ms per call at S = 30 720 = 6.7 TB/s of L2 -> LDS, 430 TFLOP/s of the causal FLOPs), not by
// the matrix pipe.  128 queries per workgroup would halve it, but one head of d = 512 needs 128 accumulator + 64 Q-fragment + 64
// staging registers per 16 queries: two such tiles per wave do not fit 512 registers, two waves per SIMD do not fit 256.
// What is saved: the 5.6 GB score / probability round trip (3.23 -> 2.67 ms), the masked key frames, two launches.
// Softmax: exp2 domain (scale * log2 e folded into one multiply), running offset per query revised lazily — only when some
// lane's tile maximum leaves a 2^40 window (the wave's first tile sets it) — so the 128 accumulator registers are rescaled rarely.
// LDS: K tile 32 x (1024 + 16) B (rows 4 banks apart, a fragment's 16 rows consecutive: conflict-free ds_read_b128), V^T tile 512 x 64 B
// with a chunk swizzle; two buffers = 130 KB, one
// workgroup per CU.  Workgroups are issued longest first (last frame first): 480 of them on 256 CUs at S = 30 720.
#include <type_traits>

#include "k5_common.h"
#include "k5_kernels.h"

namespace {

constexpr int VA_KROW = 1056;                 // bytes per staged key row (1024 + 32: conflict-free under the real ds_read_b128 lane groups, see gemm_bf16.hip W4_PAD; 1040 until round 5)
// chunk swizzle of the V^T tile (rows 64 B apart): quarter q = (row >> 2) & 3 of a 16-row fragment XORs its chunk with VA_SWZ(q) = 0, 2, 3, 1 — with the
// plain q of rounds 3-5 two (lane group, quarter) pairs of every b128 read group met on the same banks (SQ_LDS_BANK_CONFLICT = 0.50 of the LDS cycles)
#define VA_SWZ(q) ((0x78 >> (2 * (q))) & 3)
constexpr int VA_KT = 32 * VA_KROW;           // K tile
constexpr int VA_VT = 512 * 64;               // V^T tile
constexpr int VA_BUF = VA_KT + VA_VT;
constexpr float VA_THR = 40.f;

typedef __attribute__((address_space(3))) void va_lds_t;
typedef const __attribute__((address_space(1))) void va_gbl_t;

struct VaP {
  const bf16_t* q; const bf16_t* k; const bf16_t* vt; bf16_t* o;
  int S, hw, ldqk, ldvt, ldo;
  float c;   // softmax scale * log2(e)
};

__global__ __launch_bounds__(256) void vae_attn512_kernel(VaP p) {
  extern __shared__ __attribute__((aligned(16))) char vsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, g = lane >> 4;
  const int nwg = gridDim.x;
  const int qb = nwg - 1 - (int)blockIdx.x;               // longest (last) query blocks first
  const int q0 = qb * 64 + wave * 16;                     // this wave's 16 queries
  const int qi = min(q0 + l15, p.S - 1);
  // Q^T fragments: lane (l15 = query, g) holds Q[q][32 ks + 8 g .. +8]
  bf16x8 qf[16];
  {
    const bf16_t* qp = p.q + (size_t)qi * p.ldqk + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 32 * ks);
  }
  const int lim_q = min(p.S, (qi / p.hw + 1) * p.hw);                                  // this lane's query: keys < lim_q
  const int q_last = min(qb * 64 + 63, p.S - 1);
  const int lim_wg = min(p.S, (q_last / p.hw + 1) * p.hw);                             // the workgroup walks keys < lim_wg
  const int lim_min = min(p.S, (min(qb * 64, p.S - 1) / p.hw + 1) * p.hw);             // below this no query of the block masks anything
  const int T = (lim_wg + 31) >> 5;

  // one step's tiles -> LDS buffer `buf`: a wave instruction moves 1 KB = one key row, or 16 d-rows x 64 B of V^T
  auto load_tile = [&](int t, int buf) {
    char* kb = vsm + buf * VA_BUF;
    char* vb = kb + VA_KT;
    const int key0 = t * 32;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kl = 8 * wave + i;                                          // key of the tile; staged at row 16 kt + 4 a + b for
      const int row = 16 * ((kl >> 2) & 1) + 4 * (kl >> 3) + (kl & 3);      // kl = 8 a + 4 kt + b: a fragment read's 16 rows are consecutive
      const bf16_t* src = p.k + (size_t)min(key0 + kl, p.S - 1) * p.ldqk + 8 * lane;
      __builtin_amdgcn_global_load_lds((va_gbl_t*)src, (va_lds_t*)(kb + row * VA_KROW), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int d0 = 16 * (8 * wave + i);
      // LDS position (row = lane >> 2, chunk = lane & 3) receives SOURCE chunk (lane & 3) ^ VA_SWZ((row >> 2) & 3): rows 64 B apart share a
      // bank row in fours, the swizzle spreads the 16 rows of a fragment read over all 64 banks
      const bf16_t* src = p.vt + (size_t)(d0 + (lane >> 2)) * p.ldvt + key0 + 8 * ((lane & 3) ^ VA_SWZ((lane >> 4) & 3));   // ldvt covers whole 32-key tiles (launcher)
      __builtin_amdgcn_global_load_lds((va_gbl_t*)src, (va_lds_t*)(vb + d0 * 64), 16, 0, 0);
    }
  };

  f32x4 ot[32];                         // O^T: [d tile]: lane (l15 = query, g) holds d = 16 dt + 4 g + r
#pragma unroll
  for (int dt = 0; dt < 32; ++dt) ot[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float moff = 0.f, lsum = 0.f;         // the query's softmax offset (exp2 domain) and this lane's share of the row sum
  bool fresh = true;

  // Fragment reads and MFMAs of the step are inline asm in a fixed order (one wave per SIMD: nothing else hides an LDS round trip;
  // left to the compiler every pair of MFMAs waits for the two reads issued just before it, 2.6 ms per call at S = 30 720).
  // Batches of 8 fragments, the next batch in flight while the current one is multiplied; four S^T accumulator chains
  // (key tile x k-step parity) instead of two; all waits explicit.  volatile asm keeps its order; the asm reads are also invisible
  // to the compiler's LDS-DMA guard (it would put a vmcnt wait in front of every one of them).
#define VA_RD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
// The Q fragments are an AGPR operand ("a"): parked there by the register allocator anyway (64 of them next to 64 fragment and 128
// accumulator registers), they would otherwise be copied into a VGPR right before each use — a VALU write to a register the previous
// asm MFMA, which the compiler's hazard recogniser cannot see, is still reading (measured: wrong scores).
#define VA_MF(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "a"(B))
#define VA_MFA(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
  const uint32_t lds0 = (uint32_t)(uintptr_t)(va_lds_t*)vsm;
  uint32_t ka[2][2], va[2];   // fragment base addresses per buffer: K tile kt, V^T
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    ka[b][0] = lds0 + b * VA_BUF + l15 * VA_KROW + 16 * g;
    ka[b][1] = ka[b][0] + 16 * VA_KROW;
    va[b] = lds0 + b * VA_BUF + VA_KT + l15 * 64 + 16 * (g ^ VA_SWZ((l15 >> 2) & 3));
  }
  if (T > 0) load_tile(0, 0);
  __syncthreads();
  auto step = [&](auto BUFC, int t) {
    constexpr int buf = decltype(BUFC)::value;
    if (t + 1 < T) load_tile(t + 1, buf ^ 1);
    bf16x8 fr[2][8];
    f32x4 st[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) st[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- S^T: batch bt = k-steps 4 bt .. 4 bt + 3 of both key tiles ----
#ifdef VA_NO_ASM_ST
    {
      const char* kb = vsm + buf * VA_BUF;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + (16 * kt + l15) * VA_KROW + 64 * ks + 16 * g);
          st[kt][ks & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], st[kt][ks & 1], 0, 0, 0);
        }
    }
#else
#pragma unroll
    for (int i = 0; i < 8; ++i) VA_RD(fr[0][i], ka[buf][i & 1], 64 * (i >> 1));
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) {
      if (bt < 3) {
#pragma unroll
        for (int i = 0; i < 8; ++i) VA_RD(fr[(bt + 1) & 1][i], ka[buf][i & 1], 64 * (4 * (bt + 1) + (i >> 1)));
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) VA_MF(st[i & 1][(i >> 1) & 1], fr[bt & 1][i], qf[4 * bt + (i >> 1)]);
    }
    // the staging registers of the last batch are dead for the compiler from here on: nothing it schedules next may overwrite them
    // while the last MFMAs still read their operands
    asm volatile("s_nop 7" : "+v"(fr[1][0]), "+v"(fr[1][1]), "+v"(fr[1][2]), "+v"(fr[1][3]), "+v"(fr[1][4]), "+v"(fr[1][5]), "+v"(fr[1][6]),
                 "+v"(fr[1][7]) :: "memory");   // (operands: keeps them allocated up to here)
#endif
#ifndef VA_NO_ASM_PV
    // first V^T batch (d tiles 0..7) in flight during the softmax
#pragma unroll
    for (int i = 0; i < 8; ++i) VA_RD(fr[0][i], va[buf], 1024 * i);
#endif
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the asm MFMAs are invisible to the hazard recogniser: let the last ones retire
    // lane (query l15, g): key 32 t + 8 g + 4 kt + r  <-  st[kt][0][r] + st[kt][1][r]
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = (st[j >> 2][0][j & 3] + st[j >> 2][1][j & 3]) * p.c;
    if (32 * t + 32 > lim_min) {        // wave-uniform: some query of the block is masked inside this tile
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (32 * t + 8 * g + j >= lim_q) s[j] = -1.0e30f;
    }
    float mx = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7])));
    if (fresh || __any(mx - moff > VA_THR)) {     // rare: (re)set the offset to the query's current maximum
      float mf = mx;
      {
        const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(mf), __float_as_uint(mf), false, false);
        mf = fmaxf(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
        const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mf), __float_as_uint(mf), false, false);
        mf = fmaxf(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
      }
      const float mnew = fresh ? mf : fmaxf(mf, moff);   // an established offset is never lowered
      const float al = fresh ? 1.f : __builtin_amdgcn_exp2f(moff - mnew);
      moff = mnew;
      lsum *= al;
#pragma unroll
      for (int dt = 0; dt < 32; ++dt) { ot[dt][0] *= al; ot[dt][1] *= al; ot[dt][2] *= al; ot[dt][3] *= al; }
      fresh = false;
    }
    float e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = __builtin_amdgcn_exp2f(s[j] - moff);
    const u32x4 pk = {pack_bf16x2(e[0], e[1]), pack_bf16x2(e[2], e[3]), pack_bf16x2(e[4], e[5]), pack_bf16x2(e[6], e[7])};
    bf16x8 pf = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
    for (int j = 0; j < 4; ++j)   // the row sum of the bf16 probabilities the PV product uses
      lsum += __uint_as_float(pk[j] << 16) + __uint_as_float(pk[j] & 0xffff0000u);
    // ---- O^T += V^T P^T: batch bt = d tiles 8 bt .. 8 bt + 7 ----
#ifdef VA_NO_ASM_PV
    {
      const char* vb = vsm + buf * VA_BUF + VA_KT;
#pragma unroll
      for (int dt = 0; dt < 32; ++dt) {
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vb + (16 * dt + l15) * 64 + 16 * (g ^ VA_SWZ((l15 >> 2) & 3)));
        ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, ot[dt], 0, 0, 0);
      }
    }
#else
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) {
      if (bt < 3) {
#pragma unroll
        for (int i = 0; i < 8; ++i) VA_RD(fr[(bt + 1) & 1][i], va[buf], 1024 * (8 * (bt + 1) + i));
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) VA_MFA(ot[8 * bt + i], fr[bt & 1][i], pf);
    }
    asm volatile("s_nop 7" : "+v"(fr[1][0]), "+v"(fr[1][1]), "+v"(fr[1][2]), "+v"(fr[1][3]), "+v"(fr[1][4]), "+v"(fr[1][5]), "+v"(fr[1][6]),
                 "+v"(fr[1][7]), "+v"(pf) :: "memory");   // as above (the probability fragment and the staging registers die here)
#endif
    __syncthreads();   // vmcnt(0) + barrier: tile t + 1 has landed, tile t's buffer is free
  };
  for (int t = 0; t < T; t += 2) {   // the buffer index is a compile-time constant: every fragment address is a base register + immediate
    step(std::integral_constant<int, 0>{}, t);
    if (t + 1 >= T) break;
    step(std::integral_constant<int, 1>{}, t + 1);
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#undef VA_RD
#undef VA_MF
#undef VA_MFA
  // the query's four lanes (l15 + 16 g) hold 8 of every 32 keys each
  {
    const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(lsum), __float_as_uint(lsum), false, false);
    lsum = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
    const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(lsum), __float_as_uint(lsum), false, false);
    lsum = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
  }
  const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
  if (q0 + l15 < p.S) {
    bf16_t* op = p.o + (size_t)(q0 + l15) * p.ldo + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 32; ++dt)
      *reinterpret_cast<u32x2*>(op + 16 * dt) = u32x2{pack_bf16x2(ot[dt][0] * inv, ot[dt][1] * inv), pack_bf16x2(ot[dt][2] * inv, ot[dt][3] * inv)};
  }
}

}  // namespace

// q, k: [S][ldqk] bf16 (512 columns each), vt: [512][ldvt] bf16 = V transposed with ldvt >= ceil(S / 32) * 32 (the columns S .. ldvt
// must hold finite values: those keys are masked, i.e. multiplied by a probability of exactly 0), o: [S][ldo] bf16.  hw = tokens per frame.
int k5_launch_vae_attention512(const void* q, const void* k, const void* vt, void* o, int S, int hw, int ldqk, int ldvt, int ldo,
                               float scale, hipStream_t stream) {
  if (!q || !k || !vt || !o || S <= 0 || hw <= 0) return K5_ERR_ARG;
  if ((ldqk & 7) || (ldvt & 7) || (ldo & 3) || ldvt < (S + 31) / 32 * 32) return K5_ERR_ALIGN;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)vae_attn512_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * VA_BUF) != hipSuccess) return K5_ERR_HIP;
    attr_set = true;
  }
  VaP p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.vt = (const bf16_t*)vt; p.o = (bf16_t*)o;
  p.S = S; p.hw = hw; p.ldqk = ldqk; p.ldvt = ldvt; p.ldo = ldo;
  p.c = scale * 1.44269504088896340736f;
  hipLaunchKernelGGL(vae_attn512_kernel, dim3((S + 63) / 64), dim3(256), 2 * VA_BUF, stream, p);
  return hipGetLastError() == hipSuccess ? K5_OK : K5_ERR_HIP;
}
