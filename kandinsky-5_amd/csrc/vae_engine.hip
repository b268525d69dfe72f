// vae_engine.hip — HunyuanVideo causal 3D-VAE *decoder* behind the C ABI (include/k5.h: k5_vae_*).
// post_quant_conv + HunyuanVideoDecoder3D.forward (kandinsky/models/vae.py:684-696, 870) on ONE latent tile; the
// tiling / cross-fade policy (vae.py:847-1204) stays on the host mirror, which calls k5_vae_decode_tile per tile
// and k5_blend_bf16 for the cross-fades.
// Activations are channels-last bf16 [T*H*W][C]; 3x3x3 causal convs and the nearest upsample run in conv3d.hip,
// 1x1x1 convs / attention projections in gemm_bf16.hip, GroupNorm(+SiLU) / frame-causal softmax in vae_ops.hip.
// Mid-block attention (1 head, d = C): scores are materialised in fp32 ([S][S], 3.8 GB at S = 30 720 — HBM is
// 288 GB) so that softmax runs in fp32 on unrounded scores exactly like SDPA, then P (bf16) . V on MFMA.
#include <math.h>
#include <string.h>

#include <algorithm>

#include <map>
#include <string>
#include <vector>

#include "k5_common.h"
#include "k5_kernels.h"

void k5_set_error(const char* fmt, ...);

#define HIPCHK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) { k5_set_error("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); return K5_ERR_HIP; } \
  } while (0)
#define K5CHK(x)                                                                                    \
  do {                                                                                              \
    int s_ = (x);                                                                                   \
    if (s_ != K5_OK) { k5_set_error("%s -> status %d (%s:%d)", #x, s_, __FILE__, __LINE__); return s_; } \
  } while (0)

namespace {

struct Buf {   // owning device buffer: freed with the handle (k5_vae_destroy)
  void* p = nullptr; size_t bytes = 0;
  Buf() = default;
  Buf(const Buf&) = delete;
  Buf& operator=(const Buf&) = delete;
  Buf(Buf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  Buf& operator=(Buf&& o) noexcept { if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; } return *this; }
  ~Buf() { release(); }
  int ensure(size_t n) {
    if (n <= bytes) return K5_OK;
    if (p) (void)hipFree(p);
    p = nullptr; bytes = 0;
    HIPCHK(hipMalloc(&p, n));
    bytes = n;
    return K5_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
  template <class T> T* as() const { return (T*)p; }
};

uint16_t bf16_rne(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
float bf16_roundf(float f) { uint32_t u = (uint32_t)bf16_rne(f) << 16; float r; memcpy(&r, &u, 4); return r; }
float half2f(uint16_t h) {
  const uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
  uint32_t bits;
  if (e == 0) {
    if (m == 0) bits = s << 31;
    else { int ee = -1; uint32_t mm = m; do { ++ee; mm <<= 1; } while (!(mm & 1024)); bits = (s << 31) | ((uint32_t)(112 - ee) << 23) | ((mm & 1023) << 13); }
  } else if (e == 31) bits = (s << 31) | 0x7f800000u | (m << 13);
  else bits = (s << 31) | ((e + 112) << 23) | (m << 13);
  float f; memcpy(&f, &bits, 4); return f;
}

struct HostT { std::vector<float> d; std::vector<int64_t> shape; };

struct Conv { Buf w, b; int cin = 0, cin_pad = 0, cout = 0, k = 3; };   // k=3: [Cout][27][CinPad]; k=1: [Cout][CinPad]
struct GN { Buf g, b; int c = 0; };
struct Resnet { GN n1, n2; Conv c1, c2, sc; bool has_sc = false; int cin = 0, cout = 0; };
struct Up { Conv conv; int up_t = 1, up_s = 1; bool present = false; };
struct Down { Conv conv; int st_t = 1, st_s = 1; bool present = false; };
struct MidAttn { GN gn; Buf wqk, bqk, wv, bv, wo, bo, ones; };   // diffusers Attention of the mid block: 1 head of dim C

inline size_t rup(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct k5_vae {
  k5_vae_config cfg{};
  bool finalized = false;
  std::map<std::string, HostT> staged;
  std::vector<std::string> expected;
  Conv pq, conv_in, conv_out;
  Resnet mid0, mid1;
  MidAttn attn;
  std::vector<std::vector<Resnet>> up_res;
  std::vector<Up> ups;
  GN norm_out;
  // encoder half (vae.py:478-586 + quant_conv :747): optional — a T2V checkpoint load may leave it out
  std::vector<std::string> expected_enc;
  bool has_encoder = false;
  Conv e_conv_in, e_conv_out, e_quant;
  std::vector<std::vector<Resnet>> down_res;
  std::vector<Down> downs;
  Resnet e_mid0, e_mid1;
  MidAttn e_attn;
  GN e_norm_out;
  Buf xin, mom;
  int G = 32;
  // workspaces
  Buf zin, x0, bx, balt, bt1, bt2, bres, gnws, gnq, qk, vt, scores, P, o, yout;
  struct PendStats { const void* ptr = nullptr; int M = 0, C = 0, nblk = 0; } pend;   // GroupNorm partials emitted by the last conv (gnq)
  // launches per kernel route since the last reset (k5_vae_path_counts): [0] conv 128x128 tiles, [1] conv 4-wave, [2] conv 4-wave +
  // GroupNorm statistics, [3] conv_out3, [4] GroupNorm from the conv's statistics, [5] GroupNorm with its own statistics pass,
  // [6] mid attention in one kernel (C = 512), [7] mid attention as GEMM - softmax - GEMM
  long long path[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

namespace {

int upload_f32(Buf& b, const float* src, size_t n) { K5CHK(b.ensure(n * 4)); HIPCHK(hipMemcpy(b.p, src, n * 4, hipMemcpyHostToDevice)); return K5_OK; }
int upload_bf16(Buf& b, const std::vector<uint16_t>& v) { K5CHK(b.ensure(v.size() * 2)); HIPCHK(hipMemcpy(b.p, v.data(), v.size() * 2, hipMemcpyHostToDevice)); return K5_OK; }
int upload_bias(Buf& b, const float* src, size_t n) {  // autocast casts the bias to bf16
  std::vector<float> t(n);
  for (size_t i = 0; i < n; ++i) t[i] = bf16_roundf(src[i]);
  return upload_f32(b, t.data(), n);
}

const HostT* find(k5_vae* v, const std::string& k) { auto it = v->staged.find(k); return it == v->staged.end() ? nullptr : &it->second; }

// nn.Conv3d weight [Cout][Cin][kt][kh][kw] -> [Cout][kt*kh*kw][CinPad] bf16 (tap-major, channels contiguous)
int pack_conv(k5_vae* v, const std::string& name, Conv& c) {
  const HostT *w = find(v, name + ".weight"), *b = find(v, name + ".bias");
  if (!w || !b || w->shape.size() != 5) { k5_set_error("missing/odd conv %s", name.c_str()); return K5_ERR_KEY; }
  c.cout = (int)w->shape[0]; c.cin = (int)w->shape[1]; c.k = (int)w->shape[2];
  const int taps = c.k * c.k * c.k;
  c.cin_pad = c.k == 3 ? (int)rup(c.cin, 64) : (int)rup(c.cin, 8);
  std::vector<uint16_t> p((size_t)c.cout * taps * c.cin_pad, 0);
  for (int o = 0; o < c.cout; ++o)
    for (int i = 0; i < c.cin; ++i)
      for (int t = 0; t < taps; ++t)
        p[((size_t)o * taps + t) * c.cin_pad + i] = bf16_rne(w->d[((size_t)o * c.cin + i) * taps + t]);
  K5CHK(upload_bf16(c.w, p));
  K5CHK(upload_bias(c.b, b->d.data(), c.cout));
  return K5_OK;
}
int pack_gn(k5_vae* v, const std::string& name, GN& g) {
  const HostT *w = find(v, name + ".weight"), *b = find(v, name + ".bias");
  if (!w || !b) { k5_set_error("missing norm %s", name.c_str()); return K5_ERR_KEY; }
  g.c = (int)w->d.size();
  K5CHK(upload_f32(g.g, w->d.data(), g.c)); K5CHK(upload_f32(g.b, b->d.data(), g.c));
  return K5_OK;
}
int pack_resnet(k5_vae* v, const std::string& p, Resnet& r) {
  K5CHK(pack_gn(v, p + ".norm1", r.n1)); K5CHK(pack_conv(v, p + ".conv1.conv", r.c1));
  K5CHK(pack_gn(v, p + ".norm2", r.n2)); K5CHK(pack_conv(v, p + ".conv2.conv", r.c2));
  r.cin = r.c1.cin; r.cout = r.c1.cout;
  r.has_sc = find(v, p + ".conv_shortcut.conv.weight") != nullptr;
  if (r.has_sc) K5CHK(pack_conv(v, p + ".conv_shortcut.conv", r.sc));
  return K5_OK;
}
int pack_linear(k5_vae* v, const std::string& n, std::vector<uint16_t>& w, std::vector<float>& b) {
  const HostT *W = find(v, n + ".weight"), *B = find(v, n + ".bias");
  if (!W || !B) { k5_set_error("missing linear %s", n.c_str()); return K5_ERR_KEY; }
  for (float f : W->d) w.push_back(bf16_rne(f));
  for (float f : B->d) b.push_back(bf16_roundf(f));
  return K5_OK;
}

int pack_mid_attn(k5_vae* v, const std::string& ap, MidAttn& a) {
  K5CHK(pack_gn(v, ap + "group_norm", a.gn));
  std::vector<uint16_t> w; std::vector<float> b;
  K5CHK(pack_linear(v, ap + "to_q", w, b)); K5CHK(pack_linear(v, ap + "to_k", w, b));
  K5CHK(upload_bf16(a.wqk, w)); K5CHK(upload_f32(a.bqk, b.data(), b.size()));
  w.clear(); b.clear(); K5CHK(pack_linear(v, ap + "to_v", w, b));
  K5CHK(upload_bf16(a.wv, w)); K5CHK(upload_f32(a.bv, b.data(), b.size()));
  w.clear(); b.clear(); K5CHK(pack_linear(v, ap + "to_out.0", w, b));
  K5CHK(upload_bf16(a.wo, w)); K5CHK(upload_f32(a.bo, b.data(), b.size()));
  std::vector<float> one(a.gn.c, 1.0f);
  return upload_f32(a.ones, one.data(), one.size());
}

// encoder.* / quant_conv.* keys (HunyuanVideoEncoder3D.__init__ vae.py:504-572)
void expected_enc_keys(const k5_vae_config& c, std::vector<std::string>& out) {
  auto wb = [&](const std::string& n) { out.push_back(n + ".weight"); out.push_back(n + ".bias"); };
  auto resnet = [&](const std::string& p, bool sc) {
    wb(p + ".norm1"); wb(p + ".conv1.conv"); wb(p + ".norm2"); wb(p + ".conv2.conv");
    if (sc) wb(p + ".conv_shortcut.conv");
  };
  wb("quant_conv"); wb("encoder.conv_in.conv");
  int prev = c.block_out_channels[0];
  for (int i = 0; i < 4; ++i) {
    const int outc = c.block_out_channels[i];
    for (int j = 0; j < c.layers_per_block; ++j)
      resnet("encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), (j == 0 ? prev : outc) != outc);
    if (i < 3) wb("encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv.conv");
    prev = outc;
  }
  resnet("encoder.mid_block.resnets.0", false); resnet("encoder.mid_block.resnets.1", false);
  for (const char* n : {"group_norm", "to_q", "to_k", "to_v", "to_out.0"}) wb(std::string("encoder.mid_block.attentions.0.") + n);
  wb("encoder.conv_norm_out"); wb("encoder.conv_out.conv");
}

void expected_keys(const k5_vae_config& c, std::vector<std::string>& out) {
  auto wb = [&](const std::string& n) { out.push_back(n + ".weight"); out.push_back(n + ".bias"); };
  auto resnet = [&](const std::string& p, bool sc) {
    wb(p + ".norm1"); wb(p + ".conv1.conv"); wb(p + ".norm2"); wb(p + ".conv2.conv");
    if (sc) wb(p + ".conv_shortcut.conv");
  };
  wb("post_quant_conv"); wb("decoder.conv_in.conv");
  resnet("decoder.mid_block.resnets.0", false); resnet("decoder.mid_block.resnets.1", false);
  for (const char* n : {"group_norm", "to_q", "to_k", "to_v", "to_out.0"}) wb(std::string("decoder.mid_block.attentions.0.") + n);
  int prev = c.block_out_channels[3];
  for (int i = 0; i < 4; ++i) {
    const int outc = c.block_out_channels[3 - i];
    for (int j = 0; j < c.layers_per_block + 1; ++j)
      resnet("decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), (j == 0 ? prev : outc) != outc);
    if (i < 3) wb("decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv.conv");
    prev = outc;
  }
  wb("decoder.conv_norm_out"); wb("decoder.conv_out.conv");
}

// GroupNorm statistics ride in the producing conv's epilogue when that conv ran on the 4-wave kernel: conv() records what it
// emitted (tensor, rows, channels, blocks), the NEXT gn() uses it if it normalises exactly that tensor, and any conv or gn in
// between clears the record (nothing else in this file writes an activation buffer between a conv and the norm that reads it).
int gn(k5_vae* v, hipStream_t s, const GN& g, const void* x, void* out, int M, bool silu) {
  const auto pend = v->pend;
  v->pend = {};
  K5CHK(v->gnws.ensure(k5_groupnorm_workspace_bytes(M, v->G)));
  ++v->path[(pend.ptr == x && pend.M == M && pend.C == g.c && pend.nblk > 0) ? 4 : 5];
  if (pend.ptr == x && pend.M == M && pend.C == g.c && pend.nblk > 0)
    return k5_launch_groupnorm_bf16_quads(x, g.g.as<float>(), g.b.as<float>(), out, M, g.c, v->G, 1e-6f, silu ? 1 : 0, g.c, g.c,
                                          v->gnq.as<float>(), pend.nblk, v->gnws.as<float>(), s);
  return k5_launch_groupnorm_bf16(x, g.g.as<float>(), g.b.as<float>(), out, M, g.c, v->G, 1e-6f, silu ? 1 : 0, g.c, g.c, v->gnws.p, s);
}
int conv(k5_vae* v, hipStream_t s, const Conv& c, const void* x, void* out, int T, int H, int W, int up_t, int up_s, const void* resid) {
  v->pend = {};
  static const bool fused_stats = !(getenv("K5_VAE_GN_STATS") && atoi(getenv("K5_VAE_GN_STATS")) == 0);   // A/B switch for benchmarking
  const long long M = (long long)(up_t == 2 ? 2 * T - 1 : T) * (up_s * H) * (up_s * W);
  if (fused_stats && M < 0x7fffffffLL && (c.cout % 4) == 0) {
    const int nblk = 2 * (int)((M + 255) / 256);
    K5CHK(v->gnq.ensure((size_t)nblk * (c.cout / 4) * 2 * sizeof(float)));
    const int r = k5_launch_conv3d_w4(x, c.w.p, c.b.as<float>(), out, T, H, W, c.cin_pad, c.cout, up_t, up_s, c.cout, resid, c.cout,
                                      v->gnq.as<float>(), s);
    if (r == K5_OK) { v->pend = {out, (int)M, c.cout, nblk}; ++v->path[K5_CONV_KIND_W4_STATS]; return K5_OK; }
    if (r != K5_ERR_UNSUPPORTED) return r;
  }
  const int r = k5_launch_conv3d_bf16(x, c.w.p, c.b.as<float>(), out, T, H, W, c.cin_pad, c.cout, up_t, up_s, c.cout, resid, c.cout, s);
  if (r == K5_OK && k5_conv3d_last_kind() >= 0 && k5_conv3d_last_kind() < 4) ++v->path[k5_conv3d_last_kind()];
  return r;
}

// x [M][cin] -> out [M][cout] ; uses t1,t2,res as scratch.  vae.py:257-275
int resnet(k5_vae* v, hipStream_t s, const Resnet& r, const void* x, void* out, int T, int H, int W) {
  const int M = T * H * W;
  K5CHK(gn(v, s, r.n1, x, v->bt1.p, M, true));
  K5CHK(conv(v, s, r.c1, v->bt1.p, v->bt2.p, T, H, W, 1, 1, nullptr));
  K5CHK(gn(v, s, r.n2, v->bt2.p, v->bt1.p, M, true));
  const void* res = x;
  if (r.has_sc) {
    K5CHK(k5_launch_gemm_bf16(x, r.sc.w.p, r.sc.b.as<float>(), v->bres.p, M, r.cout, r.sc.cin_pad, r.cin, r.sc.cin_pad, r.cout,
                              K5_EPI_BIAS, nullptr, 0, nullptr, s));
    res = v->bres.p;
  }
  return conv(v, s, r.c2, v->bt1.p, out, T, H, W, 1, 1, res);
}

int mid_attention(k5_vae* v, hipStream_t s, const MidAttn& a, void* h, int T, int H, int W) {
  static const bool flash_ok = !(getenv("K5_VAE_FLASH") && atoi(getenv("K5_VAE_FLASH")) == 0);   // A/B switch for benchmarking
  const bool flash = flash_ok && a.gn.c == 512;   // one kernel, no score matrix (vae_attn.hip); other widths: GEMM - softmax - GEMM
  const int C = a.gn.c, S = T * H * W, Sp = (int)rup(S, flash ? 32 : 8);
  K5CHK(v->qk.ensure((size_t)S * 2 * C * 2)); K5CHK(v->vt.ensure((size_t)C * Sp * 2)); K5CHK(v->o.ensure((size_t)S * C * 2));
  if (!flash) { K5CHK(v->scores.ensure((size_t)S * Sp * 4)); K5CHK(v->P.ensure((size_t)S * Sp * 2)); }
  ++v->path[flash ? 6 : 7];
  K5CHK(gn(v, s, a.gn, h, v->bt1.p, S, false));
  K5CHK(k5_launch_gemm_bf16(v->bt1.p, a.wqk.p, a.bqk.as<float>(), v->qk.p, S, 2 * C, C, C, C, 2 * C, K5_EPI_BIAS, nullptr, 0, nullptr, s));
  HIPCHK(hipMemsetAsync(v->vt.p, 0, (size_t)C * Sp * 2, s));
  K5CHK(k5_launch_gemm_bf16(a.wv.p, v->bt1.p, a.bv.as<float>(), v->vt.p, C, S, C, C, C, Sp, K5_EPI_BIAS_M, nullptr, 0, nullptr, s));
  if (flash) {
    K5CHK(k5_launch_vae_attention512(v->qk.p, v->qk.as<bf16_t>() + C, v->vt.p, v->o.p, S, H * W, 2 * C, Sp, C, 1.0f / sqrtf((float)C), s));
    return k5_launch_gemm_bf16(v->o.p, a.wo.p, a.bo.as<float>(), h, S, C, C, C, C, C, K5_EPI_GATE, h, C, a.ones.as<float>(), s);
  }
  K5CHK(k5_launch_gemm_bf16_f32out(v->qk.p, v->qk.as<bf16_t>() + C, v->scores.as<float>(), S, S, C, 2 * C, 2 * C, Sp,
                                    1.0f / sqrtf((float)C), H * W, s));   // frame-causal: key frames after the query's are never read
  K5CHK(k5_launch_causal_softmax(v->scores.as<float>(), v->P.p, S, H * W, Sp, Sp, s));
  K5CHK(k5_launch_gemm_bf16(v->P.p, v->vt.p, nullptr, v->o.p, S, C, Sp, Sp, Sp, C, K5_EPI_BIAS, nullptr, 0, nullptr, s));
  // to_out[0] + residual (diffusers Attention residual_connection=True): bf16(h + 1 * bf16(o Wo^T + bo)), in place
  return k5_launch_gemm_bf16(v->o.p, a.wo.p, a.bo.as<float>(), h, S, C, C, C, C, C, K5_EPI_GATE, h, C, a.ones.as<float>(), s);
}

}  // namespace

extern "C" int k5_vae_create(const k5_vae_config* cfg, k5_vae** out) {
  if (!cfg || !out) return K5_ERR_ARG;
  for (int i = 0; i < 4; ++i)
    if (cfg->block_out_channels[i] % 64) { k5_set_error("block_out_channels must be multiples of 64"); return K5_ERR_UNSUPPORTED; }
  if (cfg->latent_channels % 8 || cfg->latent_channels > 64) { k5_set_error("latent_channels must be a multiple of 8, <= 64"); return K5_ERR_UNSUPPORTED; }
  const int G = cfg->norm_num_groups;
  for (int i = 0; i < 4; ++i) {
    const int cg = cfg->block_out_channels[i] / G;
    if (G > 64 || cfg->block_out_channels[i] % G || cg < 4 || (cg & (cg - 1))) { k5_set_error("unsupported norm_num_groups"); return K5_ERR_UNSUPPORTED; }
  }
  k5_vae* v = new k5_vae();
  v->cfg = *cfg; v->G = G;
  expected_keys(*cfg, v->expected);
  expected_enc_keys(*cfg, v->expected_enc);
  *out = v;
  return K5_OK;
}

extern "C" void k5_vae_destroy(k5_vae* v) { delete v; /* every Buf (weights and workspaces) frees its device memory */ }

extern "C" int k5_vae_load_tensor(k5_vae* v, const char* key, const void* ptr, int dtype, const int64_t* shape, int rank) {
  if (!v || !key || !ptr || rank < 1 || rank > 5) return K5_ERR_ARG;
  if (v->finalized) return K5_ERR_STATE;
  bool known = false;
  for (auto& e : v->expected) if (e == key) { known = true; break; }
  if (!known) for (auto& e : v->expected_enc) if (e == key) { known = true; break; }
  if (!known) return K5_OK;  // anything else in the checkpoint is ignored
  HostT t; size_t n = 1;
  for (int i = 0; i < rank; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
  t.d.resize(n);
  if (dtype == K5_F32) HIPCHK(hipMemcpy(t.d.data(), ptr, n * 4, hipMemcpyDefault));
  else if (dtype == K5_BF16 || dtype == K5_F16) {
    std::vector<uint16_t> raw(n);
    HIPCHK(hipMemcpy(raw.data(), ptr, n * 2, hipMemcpyDefault));
    for (size_t i = 0; i < n; ++i) {
      if (dtype == K5_BF16) { uint32_t u = (uint32_t)raw[i] << 16; memcpy(&t.d[i], &u, 4); } else t.d[i] = half2f(raw[i]);
    }
  } else return K5_ERR_ARG;
  v->staged[key] = std::move(t);
  return K5_OK;
}

extern "C" int k5_vae_finalize(k5_vae* v) {
  if (!v) return K5_ERR_ARG;
  if (v->finalized) return K5_OK;
  int miss = 0; std::string names;
  for (auto& e : v->expected) if (!v->staged.count(e)) { if (miss++ < 6) names += e + " "; }
  if (miss) { k5_set_error("VAE decoder: missing %d key(s): %s", miss, names.c_str()); return K5_ERR_KEY; }
  const k5_vae_config& c = v->cfg;
  K5CHK(pack_conv(v, "post_quant_conv", v->pq));
  K5CHK(pack_conv(v, "decoder.conv_in.conv", v->conv_in));
  K5CHK(pack_resnet(v, "decoder.mid_block.resnets.0", v->mid0));
  K5CHK(pack_resnet(v, "decoder.mid_block.resnets.1", v->mid1));
  K5CHK(pack_mid_attn(v, "decoder.mid_block.attentions.0.", v->attn));
  v->up_res.resize(4); v->ups.resize(4);
  for (int i = 0; i < 4; ++i) {  // up-block schedule vae.py:644-659 (time_compression 4, spatial 8)
    v->up_res[i].resize(c.layers_per_block + 1);
    for (int j = 0; j < c.layers_per_block + 1; ++j)
      K5CHK(pack_resnet(v, "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), v->up_res[i][j]));
    const bool sp = i < 3, tm = (i >= 1) && (i != 3);
    if (sp || tm) {
      K5CHK(pack_conv(v, "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv.conv", v->ups[i].conv));
      v->ups[i].present = true; v->ups[i].up_t = tm ? 2 : 1; v->ups[i].up_s = sp ? 2 : 1;
    }
  }
  K5CHK(pack_gn(v, "decoder.conv_norm_out", v->norm_out));
  K5CHK(pack_conv(v, "decoder.conv_out.conv", v->conv_out));
  // encoder: all of its keys or none (decode-only checkpoints)
  {
    int have = 0;
    for (auto& e : v->expected_enc) have += v->staged.count(e) ? 1 : 0;
    if (have && have != (int)v->expected_enc.size()) {
      std::string names; int miss = 0;
      for (auto& e : v->expected_enc) if (!v->staged.count(e)) { if (miss++ < 6) names += e + " "; }
      k5_set_error("VAE encoder: %d of its keys are missing: %s", miss, names.c_str());
      return K5_ERR_KEY;
    }
    if (have) {
      K5CHK(pack_conv(v, "quant_conv", v->e_quant));
      K5CHK(pack_conv(v, "encoder.conv_in.conv", v->e_conv_in));
      v->down_res.resize(4); v->downs.resize(4);
      for (int i = 0; i < 4; ++i) {   // down-block schedule vae.py:522-566 (temporal_compression_ratio 4, spatial 8)
        v->down_res[i].resize(c.layers_per_block);
        for (int j = 0; j < c.layers_per_block; ++j)
          K5CHK(pack_resnet(v, "encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), v->down_res[i][j]));
        const bool sp = i < 3, tm = (i >= 1) && (i != 3);
        if (sp || tm) {
          K5CHK(pack_conv(v, "encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv.conv", v->downs[i].conv));
          v->downs[i].present = true; v->downs[i].st_t = tm ? 2 : 1; v->downs[i].st_s = sp ? 2 : 1;
        }
      }
      K5CHK(pack_resnet(v, "encoder.mid_block.resnets.0", v->e_mid0));
      K5CHK(pack_resnet(v, "encoder.mid_block.resnets.1", v->e_mid1));
      K5CHK(pack_mid_attn(v, "encoder.mid_block.attentions.0.", v->e_attn));
      K5CHK(pack_gn(v, "encoder.conv_norm_out", v->e_norm_out));
      K5CHK(pack_conv(v, "encoder.conv_out.conv", v->e_conv_out));
      v->has_encoder = true;
    }
  }
  v->staged.clear();
  v->finalized = true;
  return K5_OK;
}

// z: device fp32 (Cz, T, H, W) ; out: device bf16 (Cout, To, 8H, 8W), To = 4(T-1)+1
extern "C" int k5_vae_decode_tile_strided(k5_vae* v, const float* z, int64_t z_channel_stride, int T, int H, int W, void* out, void* stream);
extern "C" int k5_vae_decode_tile(k5_vae* v, const float* z, int T, int H, int W, void* out, void* stream) {
  return k5_vae_decode_tile_strided(v, z, 0, T, H, W, out, stream);
}
// z_channel_stride: elements between the latent channels of z (0 = T H W: contiguous) — a temporal slice z[:, t0 : t0 + T] of a longer latent is read in
// place (round 6: the host mirror's tiling loop no longer copies it)
extern "C" int k5_vae_decode_tile_strided(k5_vae* v, const float* z, int64_t z_channel_stride, int T, int H, int W, void* out, void* stream) {
  if (!v || !z || !out || T <= 0 || H <= 0 || W <= 0 || (z_channel_stride != 0 && z_channel_stride < (int64_t)T * H * W)) return K5_ERR_ARG;
  if (!v->finalized) { k5_set_error("k5_vae_decode_tile before k5_vae_finalize"); return K5_ERR_STATE; }
  hipStream_t s = (hipStream_t)stream;
  const k5_vae_config& c = v->cfg;
  const int Cz = c.latent_channels;
  // activation high-water mark over the schedule
  size_t maxel = 0;
  {
    int t = T, h = H, w = W, prev = c.block_out_channels[3];
    auto upd = [&](int ch) { maxel = std::max(maxel, (size_t)t * h * w * (size_t)ch); };
    upd(prev);
    for (int i = 0; i < 4; ++i) {
      const int outc = c.block_out_channels[3 - i];
      upd(std::max(prev, outc));
      if (v->ups[i].present) { if (v->ups[i].up_t == 2) t = 2 * t - 1; h *= v->ups[i].up_s; w *= v->ups[i].up_s; upd(outc); }
      prev = outc;
    }
  }
  const int M0 = T * H * W;
  K5CHK(v->zin.ensure((size_t)M0 * Cz * 2)); K5CHK(v->x0.ensure((size_t)M0 * 64 * 2));
  for (Buf* b : {&v->bx, &v->balt, &v->bt1, &v->bt2, &v->bres}) K5CHK(b->ensure(maxel * 2));
  // post_quant_conv (1x1x1) into a 64-channel zero-padded buffer, then conv_in
  K5CHK(k5_launch_nchw_to_mc(z, v->zin.p, Cz, M0, Cz, s, z_channel_stride));
  HIPCHK(hipMemsetAsync(v->x0.p, 0, (size_t)M0 * 64 * 2, s));
  K5CHK(k5_launch_gemm_bf16(v->zin.p, v->pq.w.p, v->pq.b.as<float>(), v->x0.p, M0, Cz, v->pq.cin_pad, Cz, v->pq.cin_pad, 64, K5_EPI_BIAS,
                            nullptr, 0, nullptr, s));
  K5CHK(conv(v, s, v->conv_in, v->x0.p, v->bx.p, T, H, W, 1, 1, nullptr));
  // mid block (vae.py:341-362): resnet -> attention -> resnet ; results ping-pong through bres-free buffers
  void* cur = v->bx.p;
  void* nxt = v->balt.p;
  auto swap = [&]() { void* t_ = cur; cur = nxt; nxt = t_; };
  K5CHK(resnet(v, s, v->mid0, cur, nxt, T, H, W)); swap();
  K5CHK(mid_attention(v, s, v->attn, cur, T, H, W));
  K5CHK(resnet(v, s, v->mid1, cur, nxt, T, H, W)); swap();
  int t = T, h = H, w = W;
  for (int i = 0; i < 4; ++i) {
    for (auto& r : v->up_res[i]) { K5CHK(resnet(v, s, r, cur, nxt, t, h, w)); swap(); }
    if (v->ups[i].present) {
      K5CHK(conv(v, s, v->ups[i].conv, cur, nxt, t, h, w, v->ups[i].up_t, v->ups[i].up_s, nullptr)); swap();
      if (v->ups[i].up_t == 2) t = 2 * t - 1;
      h *= v->ups[i].up_s; w *= v->ups[i].up_s;
    }
  }
  const int M = t * h * w;
  K5CHK(gn(v, s, v->norm_out, cur, v->bt1.p, M, true));
  K5CHK(v->yout.ensure((size_t)M * c.out_channels * 2));
  K5CHK(conv(v, s, v->conv_out, v->bt1.p, v->yout.p, t, h, w, 1, 1, nullptr));
  K5CHK(k5_launch_mc_to_nchw(v->yout.p, out, c.out_channels, M, c.out_channels, s));
  return K5_OK;
}

// x: device fp32 (in_channels = 3, T, H, W), T = 4k + 1 frames (or 1), H, W multiples of 8 ->
// out: device bf16 (2 * latent_channels, (T-1)/4+1, H/8, W/8) = quant_conv(encoder(x)) = [mean | logvar]   (vae.py:808-809)
extern "C" int k5_vae_encode_tile(k5_vae* v, const float* x, int T, int H, int W, void* out, void* stream) {
  if (!v || !x || !out || T <= 0 || H <= 0 || W <= 0) return K5_ERR_ARG;
  if (!v->finalized) { k5_set_error("k5_vae_encode_tile before k5_vae_finalize"); return K5_ERR_STATE; }
  if (!v->has_encoder) { k5_set_error("this VAE handle was loaded without encoder.* / quant_conv.* tensors"); return K5_ERR_STATE; }
  if ((H & 7) || (W & 7) || ((T - 1) & 3)) { k5_set_error("encode: need H, W multiples of 8 and T = 4k + 1 frames (got %d x %d x %d)", T, H, W); return K5_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const k5_vae_config& c = v->cfg;
  const int Cin = 3;
  // activation high-water mark: the full-resolution stage
  size_t maxel = (size_t)T * H * W * (size_t)std::max(64, c.block_out_channels[0]);
  {
    int t = T, h = H, w = W;
    for (int i = 0; i < 4; ++i) {
      maxel = std::max(maxel, (size_t)t * h * w * (size_t)std::max(c.block_out_channels[i], i ? c.block_out_channels[i - 1] : 0));
      if (v->downs[i].present) { t = (t - 1) / v->downs[i].st_t + 1; h = (h - 1) / v->downs[i].st_s + 1; w = (w - 1) / v->downs[i].st_s + 1; }
    }
  }
  const int M0 = T * H * W;
  K5CHK(v->xin.ensure((size_t)M0 * 64 * 2));
  for (Buf* b : {&v->bx, &v->balt, &v->bt1, &v->bt2, &v->bres}) K5CHK(b->ensure(maxel * 2));
  K5CHK(k5_launch_nchw_to_mc(x, v->xin.p, Cin, M0, 64, s));     // channels 3..63 zero: conv_in's weight is packed to 64 input channels
  K5CHK(conv(v, s, v->e_conv_in, v->xin.p, v->bx.p, T, H, W, 1, 1, nullptr));
  void* cur = v->bx.p;
  void* nxt = v->balt.p;
  auto swap = [&]() { void* t_ = cur; cur = nxt; nxt = t_; };
  int t = T, h = H, w = W;
  for (int i = 0; i < 4; ++i) {
    for (auto& r : v->down_res[i]) { K5CHK(resnet(v, s, r, cur, nxt, t, h, w)); swap(); }
    if (v->downs[i].present) {
      const Conv& dc = v->downs[i].conv;
      K5CHK(k5_launch_conv3d_bf16_strided(cur, dc.w.p, dc.b.as<float>(), nxt, t, h, w, dc.cin_pad, dc.cout, 1, 1, v->downs[i].st_t,
                                          v->downs[i].st_s, dc.cout, nullptr, dc.cout, s));
      swap();
      t = (t - 1) / v->downs[i].st_t + 1; h = (h - 1) / v->downs[i].st_s + 1; w = (w - 1) / v->downs[i].st_s + 1;
    }
  }
  K5CHK(resnet(v, s, v->e_mid0, cur, nxt, t, h, w)); swap();
  K5CHK(mid_attention(v, s, v->e_attn, cur, t, h, w));
  K5CHK(resnet(v, s, v->e_mid1, cur, nxt, t, h, w)); swap();
  const int M = t * h * w, C2 = 2 * c.latent_channels;
  K5CHK(gn(v, s, v->e_norm_out, cur, v->bt1.p, M, true));
  K5CHK(conv(v, s, v->e_conv_out, v->bt1.p, v->bt2.p, t, h, w, 1, 1, nullptr));      // [M][2 Cz]
  K5CHK(v->mom.ensure((size_t)M * C2 * 2));
  K5CHK(k5_launch_gemm_bf16(v->bt2.p, v->e_quant.w.p, v->e_quant.b.as<float>(), v->mom.p, M, C2, v->e_quant.cin_pad, C2, v->e_quant.cin_pad,
                            C2, K5_EPI_BIAS, nullptr, 0, nullptr, s));             // quant_conv 1x1x1
  K5CHK(k5_launch_mc_to_nchw(v->mom.p, out, C2, M, C2, s));
  return K5_OK;
}

extern "C" int k5_vae_has_encoder(k5_vae* v) { return v && v->has_encoder ? 1 : 0; }

extern "C" int k5_vae_path_counts(k5_vae* v, long long* out8, int reset) {
  if (!v || !out8) return K5_ERR_ARG;
  for (int i = 0; i < 8; ++i) { out8[i] = v->path[i]; if (reset) v->path[i] = 0; }
  return K5_OK;
}

extern "C" int k5_blend_bf16(const void* a, void* b, int64_t outer, int len_a, int len_b, int64_t inner, int extent, void* stream) {
  return k5_launch_blend_bf16(a, b, outer, len_a, len_b, inner, extent, (hipStream_t)stream);
}
extern "C" int k5_blend_place_bf16(const void* a, int64_t a_stride, int len_a, const void* b, int64_t b_stride, void* dst, int64_t dst_stride, int64_t outer,
                                   int64_t inner, int extent, int keep, void* stream) {
  return k5_launch_blend_place_bf16(a, a_stride, len_a, b, b_stride, dst, dst_stride, outer, inner, extent, keep, (hipStream_t)stream);
}
extern "C" int k5_frames_to_uint8(const void* x_bf16, void* out_u8, int64_t n, void* stream) {
  return k5_launch_frames_to_uint8(x_bf16, out_u8, n, (hipStream_t)stream);
}
