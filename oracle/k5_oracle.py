"""k5_oracle — CPU restatement of the Kandinsky-5 T2V Lite denoising path.

*** TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
*** leg may import this module.  The product path (kandinsky-5_amd/) never does; it fails
*** loudly when libk5.so is missing.

Own code, plain torch-CPU ops, no autocast tricks.  Every function cites the reference
file:line it restates (paths relative to the reference repo root).  Parity pinning: the
reference ships no tests / golden vectors, so this restatement is pinned
against *outputs of the reference itself* generated in the build container by
oracle/gen_golden.py (imports /root/reference under the patches in oracle/_ref_import.py)
and committed as data under tests/golden/.  tests/test_oracle_vs_golden.py checks every one.
Third-party arithmetic the reference calls but does not contain (flash-attn `FA`, diffusers
`Attention`) is restated from its mathematical definition: softmax(QK^T/sqrt(d))V.

Two arithmetic modes:
  mode="fp32"  everything fp32 — equals the reference run under the fp32 oracle patches.
  mode="bf16"  the reference's CUDA-autocast rounding points: bf16
               rounding after every autocast linear / K1 / K2 / RoPE / RMSNorm / attention /
               GELU, fp32 islands (time MLP, modulation, LayerNorm math, RoPE tables), bf16
               CFG combine, fp32 latent.  This is the parity target of the HIP engine.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------
# config
# ------------------------------------------------------------------------------------------
@dataclass
class DitConfig:
    """ctor kwargs of DiffusionTransformer3D (kandinsky/models/dit.py:83-97)."""

    in_visual_dim: int = 4
    in_text_dim: int = 3584
    in_text_dim2: int = 768
    time_dim: int = 512
    out_visual_dim: int = 4
    patch_size: Tuple[int, int, int] = (1, 2, 2)
    model_dim: int = 2048
    ff_dim: int = 5120
    num_text_blocks: int = 2
    num_visual_blocks: int = 32
    axes_dims: Tuple[int, int, int] = (16, 24, 24)
    visual_cond: bool = False

    @property
    def head_dim(self) -> int:  # dit.py:99
        return sum(self.axes_dims)

    @property
    def num_heads(self) -> int:  # nn.py:222
        return self.model_dim // self.head_dim

    @property
    def visual_embed_dim(self) -> int:  # dit.py:105
        return 2 * self.in_visual_dim + 1 if self.visual_cond else self.in_visual_dim


LITE_2B = dict(  # configs/config_5s_sft.yaml:12-30 == kandinsky/utils.py:143-156
    in_visual_dim=16, out_visual_dim=16, time_dim=512, patch_size=(1, 2, 2), model_dim=1792,
    ff_dim=7168, num_text_blocks=2, num_visual_blocks=32, axes_dims=(16, 24, 24),
    visual_cond=True, in_text_dim=3584, in_text_dim2=768,
)


def state_dict_manifest(cfg: DitConfig) -> Dict[str, Tuple[int, ...]]:
    """Names and shapes of the DiT checkpoint (dit.py:100-127, nn.py)."""
    D, FF, TD = cfg.model_dim, cfg.ff_dim, cfg.time_dim
    hd = cfg.head_dim
    m: Dict[str, Tuple[int, ...]] = {}

    def lin(name, o, i, bias=True):
        m[name + ".weight"] = (o, i)
        if bias:
            m[name + ".bias"] = (o,)

    lin("time_embeddings.in_layer", TD, D)
    lin("time_embeddings.out_layer", TD, TD)
    lin("text_embeddings.in_layer", D, cfg.in_text_dim)
    m["text_embeddings.norm.weight"] = (D,)
    m["text_embeddings.norm.bias"] = (D,)
    lin("pooled_text_embeddings.in_layer", TD, cfg.in_text_dim2)
    m["pooled_text_embeddings.norm.weight"] = (TD,)
    m["pooled_text_embeddings.norm.bias"] = (TD,)
    lin("visual_embeddings.in_layer", D, math.prod(cfg.patch_size) * cfg.visual_embed_dim)

    def attn(prefix):
        for n in ("to_query", "to_key", "to_value"):
            lin(f"{prefix}.{n}", D, D)
        m[f"{prefix}.query_norm.weight"] = (hd,)
        m[f"{prefix}.key_norm.weight"] = (hd,)
        lin(f"{prefix}.out_layer", D, D)

    for i in range(cfg.num_text_blocks):
        p = f"text_transformer_blocks.{i}"
        lin(f"{p}.text_modulation.out_layer", 6 * D, TD)
        attn(f"{p}.self_attention")
        lin(f"{p}.feed_forward.in_layer", FF, D, bias=False)
        lin(f"{p}.feed_forward.out_layer", D, FF, bias=False)
    for i in range(cfg.num_visual_blocks):
        p = f"visual_transformer_blocks.{i}"
        lin(f"{p}.visual_modulation.out_layer", 9 * D, TD)
        attn(f"{p}.self_attention")
        attn(f"{p}.cross_attention")
        lin(f"{p}.feed_forward.in_layer", FF, D, bias=False)
        lin(f"{p}.feed_forward.out_layer", D, FF, bias=False)
    lin("out_layer.modulation.out_layer", 2 * D, TD)
    lin("out_layer.out_layer", math.prod(cfg.patch_size) * cfg.out_visual_dim, D)
    return m


def synthetic_state_dict(cfg: DitConfig, seed: int = 0, std: float = 0.02,
                         dtype=torch.float32) -> Dict[str, Tensor]:
    """Synthetic weights laid out like the checkpoint: Linear ~ N(0, std²)
    incl. the (reference: zero-init, nn.py:158-159) Modulation layers, norm weights 1,
    biases N(0, std²).  Per-tensor generator seeded by (seed, index) so that any subset can be
    regenerated independently."""
    sd = {}
    for idx, (name, shape) in enumerate(state_dict_manifest(cfg).items()):
        if name.endswith("norm.weight") and len(shape) == 1:
            sd[name] = torch.ones(shape, dtype=dtype)
            continue
        g = torch.Generator().manual_seed(seed * 1000003 + idx)
        s = std
        if "modulation" in name:
            s = 2.5 * std  # visible AdaLN modulation so every block is exercised
        sd[name] = (torch.randn(shape, generator=g) * s).to(dtype)
    return sd


# ------------------------------------------------------------------------------------------
# rounding helper
# ------------------------------------------------------------------------------------------
def _r(x: Tensor, mode: str) -> Tensor:
    """bf16 rounding point (identity in fp32 mode). Values stay stored as fp32."""
    if mode == "bf16":
        return x.to(torch.bfloat16).to(torch.float32)
    return x


LN_FOLD = False   # experiment switch (round 6, oracle/gen_lnfold_yardstick.py): the LayerNorm + modulation of the visual blocks folded into the consumer linear layer


class _LnIn:
    """What scale_shift_norm returns under LN_FOLD: the un-normalised (bf16) residual rows with the modulation vectors — the consumer `_linear` computes
    y = rstd * (x . W'^T - mean * colsum(W')) + W . shift + b with W' = bf16(W * (1 + scale)): the engine-side form in which the three LayerNorm passes of
    a block (nn.py:25-28) disappear into the GEMMs.  The rounding point moves from the normalised activations to the scaled weights."""
    def __init__(self, x, scale, shift):
        self.x, self.scale, self.shift = x, scale.reshape(-1), shift.reshape(-1)
        self.shape = x.shape


def _linear(x, w, b, mode):
    """autocast nn.Linear: bf16 operands, fp32 accumulate, single rounding of acc+bias
    (nn.py:180-191,204-206,359-361 under generation_utils.py:185)."""
    if isinstance(x, _LnIn):
        xr = _r(x.x, mode)                                            # the residual stream as it sits in memory
        mu = xr.mean(-1, keepdim=True)
        rs = torch.rsqrt(((xr - mu) ** 2).mean(-1, keepdim=True) + 1e-5)
        wr = _r(w, mode)
        w1 = _r(wr * (1.0 + x.scale)[None, :], mode)                  # W' rounded to bf16 once per forward
        y = rs * (xr @ w1.t() - mu * w1.sum(-1)[None, :]) + (wr @ x.shift)[None, :]
        if b is not None:
            y = y + _r(b, mode)
        return _r(y, mode)
    y = _r(x, mode) @ _r(w, mode).t()
    if b is not None:
        y = y + _r(b, mode)
    return _r(y, mode)


# ------------------------------------------------------------------------------------------
# elementary pieces
# ------------------------------------------------------------------------------------------
def get_freqs(dim: int, max_period: float = 10000.0) -> Tensor:
    """models/utils.py:21-28."""
    return torch.exp(-math.log(max_period) * torch.arange(0, dim, dtype=torch.float32) / dim)


def time_embeddings(sd, time: Tensor, cfg: DitConfig) -> Tensor:
    """nn.py:43-61 (fp32 island in both modes)."""
    freqs = get_freqs(cfg.model_dim // 2)
    args = torch.outer(time.float(), freqs)
    e = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    h = F.silu(e @ sd["time_embeddings.in_layer.weight"].float().t()
               + sd["time_embeddings.in_layer.bias"].float())
    return h @ sd["time_embeddings.out_layer.weight"].float().t() \
        + sd["time_embeddings.out_layer.bias"].float()


def text_embeddings(sd, prefix: str, x: Tensor, mode: str) -> Tensor:
    """nn.py:64-72: Linear (autocast bf16) -> LayerNorm affine eps 1e-5 (fp32) -> type_as."""
    h = _linear(x, sd[f"{prefix}.in_layer.weight"].float(), sd[f"{prefix}.in_layer.bias"].float(), mode)
    h = F.layer_norm(h, (h.shape[-1],), sd[f"{prefix}.norm.weight"].float(),
                     sd[f"{prefix}.norm.bias"].float(), eps=1e-5)
    return _r(h, mode)


def patchify(x: Tensor, patch: Sequence[int]) -> Tensor:
    """nn.py:81-95: (T,H,W,C) -> (T/pt, H/ph, W/pw, pt*ph*pw*C), feature order (pt,ph,pw,c)."""
    T, H, W, C = x.shape
    pt, ph, pw = patch
    x = x.reshape(T // pt, pt, H // ph, ph, W // pw, pw, C)
    return x.permute(0, 2, 4, 1, 3, 5, 6).reshape(T // pt, H // ph, W // pw, pt * ph * pw * C)


def unpatchify(x: Tensor, patch: Sequence[int]) -> Tensor:
    """nn.py:384-399: (T,H,W,C*pt*ph*pw) with feature order (c,pt,ph,pw) -> (T*pt,H*ph,W*pw,C)."""
    T, H, W, F_ = x.shape
    pt, ph, pw = patch
    C = F_ // (pt * ph * pw)
    x = x.reshape(T, H, W, C, pt, ph, pw)
    return x.permute(0, 4, 1, 5, 2, 6, 3).reshape(T * pt, H * ph, W * pw, C)


def visual_embeddings(sd, x: Tensor, cfg: DitConfig, mode: str) -> Tensor:
    """nn.py:75-96."""
    return _linear(patchify(x.float(), cfg.patch_size), sd["visual_embeddings.in_layer.weight"].float(),
                   sd["visual_embeddings.in_layer.bias"].float(), mode)


def rope_1d_args(pos: Tensor, dim: int) -> Tensor:
    """nn.py:99-116: angles (L, dim/2) = pos ⊗ freqs."""
    return torch.outer(pos.float(), get_freqs(dim // 2))


def rope_3d_args(shape, pos, axes_dims, scale_factor=(1.0, 1.0, 1.0)) -> Tensor:
    """nn.py:119-150: angles (T,H,W, sum(axes)/2), order t|h|w, each axis angle / scale_factor.
    The reference gathers from a precomputed outer(pos, freq) table; outer of the gathered
    positions is the same fp32 product."""
    T, H, W = shape
    at = torch.outer(pos[0].float(), get_freqs(axes_dims[0] // 2)) / scale_factor[0]
    ah = torch.outer(pos[1].float(), get_freqs(axes_dims[1] // 2)) / scale_factor[1]
    aw = torch.outer(pos[2].float(), get_freqs(axes_dims[2] // 2)) / scale_factor[2]
    return torch.cat([
        at.view(T, 1, 1, -1).expand(T, H, W, -1),
        ah.view(1, H, 1, -1).expand(T, H, W, -1),
        aw.view(1, 1, W, -1).expand(T, H, W, -1),
    ], dim=-1)


PRESCALE_K = False   # test switch: the engine's dense visual self-attention hands the softmax keys pre-multiplied by
#                      log2(e)/8 (one bf16 rounding, csrc/small_ops.hip rmsnorm_rope_kernel) and exponentiates in base 2
SOFTMAX_C = 0.125 * 1.44269504088896340736


def apply_rotary(x: Tensor, cos: Tensor, sin: Tensor, mode: str, out_scale: float = 1.0) -> Tensor:
    """nn.py:35-40 with rope = [[cos,-sin],[sin,cos]] (nn.py:112-116): adjacent pairs
    (x0,x1) -> (c*x0 - s*x1, s*x0 + c*x1), fp32 math, bf16 rounding.
    x (S,H,hd); cos/sin (S, hd/2)."""
    S, H, hd = x.shape
    xp = x.float().reshape(S, H, hd // 2, 2)
    c = cos[:, None, :]
    s = sin[:, None, :]
    o0 = c * xp[..., 0] + (-s) * xp[..., 1]
    o1 = s * xp[..., 0] + c * xp[..., 1]
    rot = torch.stack([o0, o1], dim=-1).reshape(S, H, hd)
    if out_scale != 1.0:
        rot = rot * torch.tensor(out_scale, dtype=torch.float32)
    return _r(rot, mode)


def modulation(sd, prefix: str, temb: Tensor) -> Tensor:
    """nn.py:153-164 (fp32 island)."""
    return F.silu(temb.float()) @ sd[f"{prefix}.out_layer.weight"].float().t() \
        + sd[f"{prefix}.out_layer.bias"].float()


def scale_shift_norm(x: Tensor, scale: Tensor, shift: Tensor, mode: str) -> Tensor:
    """apply_scale_shift_norm nn.py:25-28: LayerNorm(no affine, eps 1e-5) fp32 · (scale+1) + shift -> bf16."""
    if LN_FOLD and mode == "bf16" and x.shape[0] > 256:               # the visual stream (the text stream has <= 256 rows in every workload)
        return _LnIn(x, scale, shift)
    n = F.layer_norm(x.float(), (x.shape[-1],), None, None, eps=1e-5)
    return _r(n * (scale + 1.0) + shift, mode)


def gate_sum(x: Tensor, out: Tensor, gate: Tensor, mode: str) -> Tensor:
    """apply_gate_sum nn.py:30-33."""
    return _r(x + gate * out, mode)


def rms_norm_heads(x: Tensor, w: Tensor, mode: str) -> Tensor:
    """norm_qk nn.py:193-197: nn.RMSNorm(head_dim) on x.float(), eps=finfo(fp32).eps, weight; type_as."""
    xf = x.float()
    eps = torch.finfo(torch.float32).eps
    y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * w.float()
    return _r(y, mode)


def sdpa(q: Tensor, k: Tensor, v: Tensor, mode: str, block_mask: Optional[Tensor] = None, base2: bool = False) -> Tensor:
    """FA(q,k,v) nn.py:201,254,336 — third-party flash-attn (absent from the reference tree):
    softmax(q k^T / sqrt(d)) v, non-causal, fp32 softmax, bf16 output.
    q (Sq,H,d), k,v (Sk,H,d) -> (Sq, H*d).  block_mask (H, Sq/64, Sk/64) bool for NABLA."""
    Sq, H, d = q.shape
    qh, kh, vh = q.float().transpose(0, 1), k.float().transpose(0, 1), v.float().transpose(0, 1)
    out = torch.empty(H, Sq, d)
    scale = math.log(2.0) if base2 else 1.0 / math.sqrt(d)   # base2: k already carries log2(e)/sqrt(d); 2^x = e^(x ln 2)
    chunk = 2048
    for h in range(H):
        for s0 in range(0, Sq, chunk):
            s = (qh[h, s0:s0 + chunk] @ kh[h].t()) * scale
            if block_mask is not None:
                bm = block_mask[h].repeat_interleave(64, 0).repeat_interleave(64, 1)
                s = s.masked_fill(~bm[s0:s0 + chunk], float("-inf"))
            out[h, s0:s0 + chunk] = torch.softmax(s, dim=-1) @ vh[h]
    return _r(out.transpose(0, 1).reshape(Sq, H * d), mode)


FP8_FF = False   # test switch: restate the engine's opt-in W8A8 e4m3 feed-forward (csrc/gemm_fp8.hip) in bf16 mode


def _q8(x: Tensor) -> Tensor:
    """saturating e4m3 round trip"""
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


def feed_forward(sd, prefix: str, x: Tensor, mode: str) -> Tensor:
    """nn.py:352-361: Linear(no bias) -> exact-erf GELU (on the bf16 GEMM output) -> Linear."""
    if FP8_FF and mode == "bf16" and x.shape[0] >= 256 and prefix.startswith("visual_transformer_blocks"):
        w1, w2 = _r(sd[f"{prefix}.in_layer.weight"].float(), mode), _r(sd[f"{prefix}.out_layer.weight"].float(), mode)
        s1, s2 = w1.abs().amax(1) / 448.0, w2.abs().amax(1) / 448.0          # per output channel
        h = _r((_q8(x) @ _q8(w1 / s1[:, None]).t()) * s1, mode)
        h = _q8(F.gelu(h))
        return _r((h @ _q8(w2 / s2[:, None]).t()) * s2, mode)
    h = _linear(x, sd[f"{prefix}.in_layer.weight"].float(), None, mode)
    h = _r(F.gelu(h), mode)
    return _linear(h, sd[f"{prefix}.out_layer.weight"].float(), None, mode)


# ------------------------------------------------------------------------------------------
# NABLA pieces
# ------------------------------------------------------------------------------------------
def fractal_perm(shape: Sequence[int]) -> Tensor:
    """models/utils.py:31-41,54-78: index vector `perm` with flat_fractal[i] = flat_raster[perm[i]];
    token i = ((t*Hb+hb)*Wb+wb)*64 + hi*8+wi  <->  raster (t, hb*8+hi, wb*8+wi)."""
    T, H, W = shape
    idx = torch.arange(T * H * W).reshape(T, H // 8, 8, W // 8, 8)
    return idx.permute(0, 1, 3, 2, 4).reshape(-1)


def fast_sta(T: int, H: int, W: int, wT: int = 3, wH: int = 3, wW: int = 3) -> Tensor:
    """fast_sta_nabla models/utils.py:108-133: (T*H*W, T*H*W) bool; block (t,h,w) sees (t',h',w')
    iff |t-t'|<=wT//2 and |h-h'|<=wH//2 and |w-w'|<=wW//2."""
    t = torch.arange(T)
    h = torch.arange(H)
    w = torch.arange(W)
    mt = (t[:, None] - t[None, :]).abs() <= wT // 2
    mh = (h[:, None] - h[None, :]).abs() <= wH // 2
    mw = (w[:, None] - w[None, :]).abs() <= wW // 2
    m = mt[:, None, None, :, None, None] & mh[None, :, None, None, :, None] & mw[None, None, :, None, None, :]
    return m.reshape(T * H * W, T * H * W)


def nabla_block_mask(q: Tensor, k: Tensor, sta: Tensor, thr: float, mode: str) -> Tensor:
    """nablaT_v2 models/utils.py:136-163, returning the dense (H, nb, nb) bool block mask
    (the reference builds a flex BlockMask from the same boolean matrix).
    q,k (S,H,d) after RMSNorm+RoPE, fractal order."""
    S, H, d = q.shape
    nb = S // 64
    qa = _r(q.float().transpose(0, 1).reshape(H, nb, 64, d).mean(-2), mode)
    ka = _r(k.float().transpose(0, 1).reshape(H, nb, 64, d).mean(-2), mode)
    m = _r(qa @ ka.transpose(-2, -1), mode)
    m = torch.softmax(_r(m / math.sqrt(d), mode), dim=-1)
    vals, inds = m.sort(-1)
    cvals = vals.cumsum(-1)
    keep_sorted = (cvals >= 1 - thr)
    keep = torch.zeros_like(keep_sorted)
    keep.scatter_(-1, inds, keep_sorted)
    return keep | sta.bool()[None]


# ------------------------------------------------------------------------------------------
# blocks / forward
# ------------------------------------------------------------------------------------------
FP8_QKV = False  # test switches: the engine's opt-in e4m3 q | k | V^T / out projections of the visual self-attention (k5_dit_set_fp8 bits 1 / 2)
FP8_OUT = False


def _linear_fp8(x, w, b, mode):
    """W8A8 e4m3 linear as csrc/gemm_fp8.hip computes it: activations with the static scale 1, weights per output channel (max|w| / 448),
    fp32 accumulate, one rounding of acc * scale + bias"""
    w = _r(w, mode)
    s = w.abs().amax(1) / 448.0
    y = (_q8(_r(x, mode)) @ _q8(w / s[:, None]).t()) * s
    if b is not None:
        y = y + _r(b, mode)
    return _r(y, mode)


def _attn_qkv(sd, prefix, xq, xkv, mode, H):
    lin = _linear_fp8 if (FP8_QKV and mode == "bf16" and xq.shape[0] >= 256 and prefix.startswith("visual_transformer_blocks")
                          and prefix.endswith("self_attention")) else _linear
    q = lin(xq, sd[f"{prefix}.to_query.weight"].float(), sd[f"{prefix}.to_query.bias"].float(), mode)
    k = lin(xkv, sd[f"{prefix}.to_key.weight"].float(), sd[f"{prefix}.to_key.bias"].float(), mode)
    v = lin(xkv, sd[f"{prefix}.to_value.weight"].float(), sd[f"{prefix}.to_value.bias"].float(), mode)
    q = q.reshape(q.shape[0], H, -1)
    k = k.reshape(k.shape[0], H, -1)
    v = v.reshape(v.shape[0], H, -1)
    q = rms_norm_heads(q, sd[f"{prefix}.query_norm.weight"], mode)
    k = rms_norm_heads(k, sd[f"{prefix}.key_norm.weight"], mode)
    return q, k, v


def self_attention(sd, prefix, x, cos, sin, cfg, mode, sparse=None, taps=None):
    """MultiheadSelfAttentionEnc/Dec.forward nn.py:208-217,286-298."""
    q, k, v = _attn_qkv(sd, prefix, x, x, mode, cfg.num_heads)
    pre = PRESCALE_K and mode == "bf16" and sparse is None and prefix.startswith("visual_transformer_blocks")
    q = apply_rotary(q, cos, sin, mode)
    k = apply_rotary(k, cos, sin, mode, SOFTMAX_C if pre else 1.0)
    bm = None
    lin_o = _linear_fp8 if (FP8_OUT and mode == "bf16" and x.shape[0] >= 256 and prefix.startswith("visual_transformer_blocks")) else _linear
    if pre:
        o = sdpa(q, k, v, mode, None, base2=True)
        return lin_o(o, sd[f"{prefix}.out_layer.weight"].float(), sd[f"{prefix}.out_layer.bias"].float(), mode)
    if sparse is not None:
        bm = nabla_block_mask(q, k, sparse["sta_mask"], sparse["P"], mode)
        if taps is not None:
            taps.setdefault("nabla_masks", []).append(bm)
    o = sdpa(q, k, v, mode, bm)
    return lin_o(o, sd[f"{prefix}.out_layer.weight"].float(), sd[f"{prefix}.out_layer.bias"].float(), mode)


def cross_attention(sd, prefix, x, cond, cfg, mode):
    """MultiheadCrossAttention.forward nn.py:343-349 (no RoPE)."""
    q, k, v = _attn_qkv(sd, prefix, x, cond, mode, cfg.num_heads)
    o = sdpa(q, k, v, mode)
    return _linear(o, sd[f"{prefix}.out_layer.weight"].float(), sd[f"{prefix}.out_layer.bias"].float(), mode)


def encoder_block(sd, p, x, temb, cos, sin, cfg, mode):
    """TransformerEncoderBlock.forward dit.py:33-44."""
    mod = modulation(sd, f"{p}.text_modulation", temb)
    sa, ff = torch.chunk(mod, 2, dim=-1)
    shift, scale, gate = torch.chunk(sa, 3, dim=-1)
    out = scale_shift_norm(x, scale, shift, mode)
    out = self_attention(sd, f"{p}.self_attention", out, cos, sin, cfg, mode)
    x = gate_sum(x, out, gate, mode)
    shift, scale, gate = torch.chunk(ff, 3, dim=-1)
    out = scale_shift_norm(x, scale, shift, mode)
    out = feed_forward(sd, f"{p}.feed_forward", out, mode)
    return gate_sum(x, out, gate, mode)


def decoder_block(sd, p, x, text, temb, cos, sin, cfg, mode, sparse=None, taps=None):
    """TransformerDecoderBlock.forward dit.py:61-79."""
    mod = modulation(sd, f"{p}.visual_modulation", temb)
    sa, ca, ff = torch.chunk(mod, 3, dim=-1)
    shift, scale, gate = torch.chunk(sa, 3, dim=-1)
    out = scale_shift_norm(x, scale, shift, mode)
    out = self_attention(sd, f"{p}.self_attention", out, cos, sin, cfg, mode, sparse, taps)
    x = gate_sum(x, out, gate, mode)
    shift, scale, gate = torch.chunk(ca, 3, dim=-1)
    out = scale_shift_norm(x, scale, shift, mode)
    out = cross_attention(sd, f"{p}.cross_attention", out, text, cfg, mode)
    x = gate_sum(x, out, gate, mode)
    shift, scale, gate = torch.chunk(ff, 3, dim=-1)
    out = scale_shift_norm(x, scale, shift, mode)
    out = feed_forward(sd, f"{p}.feed_forward", out, mode)
    return gate_sum(x, out, gate, mode)


def out_layer(sd, x, temb, cfg, mode):
    """OutLayer.forward nn.py:374-400 on (Ntok, D) raster-ordered tokens -> (Ntok, prod(patch)*C)."""
    shift, scale = torch.chunk(modulation(sd, "out_layer.modulation", temb), 2, dim=-1)
    h = scale_shift_norm(x, scale, shift, mode)
    return _linear(h, sd["out_layer.out_layer.weight"].float(), sd["out_layer.out_layer.bias"].float(), mode)


class MagCache:
    """MagCache state machine (magcache_utils.py:16-39 `set_magcache_params`, :59-76,91-100 the decision inside
    `magcache_forward`).  Decisions depend on the ratio table and the call counter only, never on the data: after the
    first `retention_ratio` of the 2*num_steps calls, the 32 visual blocks are skipped (the cached residual of the same
    cond/uncond slot is re-applied) while the accumulated |1 - prod(ratios)| stays below `thresh` for at most K
    consecutive calls of that slot.  All scalar arithmetic is float64 (numpy in the reference)."""

    def __init__(self, mag_ratios, num_steps: int, no_cfg: bool, thresh: float = 0.12, K: int = 2,
                 retention_ratio: float = 0.2):
        self.cnt = 0
        self.num_steps = num_steps * 2
        self.thresh, self.K, self.retention_ratio, self.no_cfg = thresh, K, retention_ratio, no_cfg
        self.accumulated_err, self.accumulated_steps, self.accumulated_ratio = [0.0, 0.0], [0, 0], [1.0, 1.0]
        self.residual_cache = [None, None]
        r = np.array([1.0] * 2 + list(mag_ratios), dtype=np.float64)
        if len(r) != num_steps * 2:  # nearest_interp magcache_utils.py:6-13 on the cond / uncond halves
            r = np.stack([self._nearest(r[0::2], num_steps), self._nearest(r[1::2], num_steps)], axis=1).reshape(-1)
        self.mag_ratios = r
        self.ran_blocks = []  # test hook: 1 if the visual blocks ran in that call

    @staticmethod
    def _nearest(src, n):
        if n == 1:
            return np.array([src[-1]])
        idx = np.round(np.arange(n) * ((len(src) - 1) / (n - 1))).astype(int)  # np.round: half to even
        return src[idx]

    def decide(self) -> bool:
        """True = skip the visual blocks in this call (magcache_utils.py:62-76)."""
        skip, s = False, self.cnt % 2
        if self.cnt >= int(self.num_steps * self.retention_ratio):
            self.accumulated_ratio[s] = self.accumulated_ratio[s] * self.mag_ratios[self.cnt]
            self.accumulated_steps[s] += 1
            self.accumulated_err[s] += np.abs(1 - self.accumulated_ratio[s])
            if self.accumulated_err[s] < self.thresh and self.accumulated_steps[s] <= self.K:
                skip = True
            else:
                self.accumulated_err[s], self.accumulated_steps[s], self.accumulated_ratio[s] = 0, 0, 1.0
        return skip

    def advance(self):
        """magcache_utils.py:91-100."""
        self.cnt += 2 if self.no_cfg else 1
        if self.cnt >= self.num_steps:
            self.cnt = 0
            self.accumulated_ratio, self.accumulated_err, self.accumulated_steps = [1.0, 1.0], [0.0, 0.0], [0, 0]


def dit_forward(sd, cfg: DitConfig, x, text_embed, pooled_text_embed, time, visual_rope_pos,
                text_rope_pos, scale_factor=(1.0, 1.0, 1.0), sparse_params=None, mode="fp32",
                taps: Optional[dict] = None, num_visual_blocks: Optional[int] = None,
                magcache: Optional["MagCache"] = None) -> Tensor:
    """DiffusionTransformer3D.forward dit.py:155-181 (with `magcache`: magcache_utils.py:42-101).
    x (T,H,W,Cin) fp32; text_embed (L,in_text_dim); pooled (1,in_text_dim2); time (1,) = 1000*sigma.
    Returns velocity (T,H,W,out_visual_dim) (bf16-valued in bf16 mode)."""
    with torch.no_grad():
        # before_text_transformer_blocks dit.py:129-137
        text = text_embeddings(sd, "text_embeddings", text_embed.float(), mode)
        temb = time_embeddings(sd, time, cfg)
        temb = temb + text_embeddings(sd, "pooled_text_embeddings", pooled_text_embed.float(), mode)
        vis = visual_embeddings(sd, x, cfg, mode)
        ta = rope_1d_args(text_rope_pos, cfg.head_dim)
        tcos, tsin = torch.cos(ta), torch.sin(ta)
        if taps is not None:
            taps["time_embed"] = temb
            taps["text_in"] = text
            taps["visual_in"] = vis
        for i in range(cfg.num_text_blocks):
            text = encoder_block(sd, f"text_transformer_blocks.{i}", text, temb, tcos, tsin, cfg, mode)
        if taps is not None:
            taps["text_out"] = text
        # before_visual_transformer_blocks dit.py:139-147
        Tp, Hp, Wp, D = vis.shape
        va = rope_3d_args((Tp, Hp, Wp), visual_rope_pos, cfg.axes_dims, scale_factor)
        vcos, vsin = torch.cos(va).reshape(-1, va.shape[-1]), torch.sin(va).reshape(-1, va.shape[-1])
        vis = vis.reshape(-1, D)
        to_fractal = sparse_params is not None and sparse_params.get("to_fractal", False)
        if to_fractal:
            perm = fractal_perm((Tp, Hp, Wp))
            vis, vcos, vsin = vis[perm], vcos[perm], vsin[perm]
        nvb = cfg.num_visual_blocks if num_visual_blocks is None else num_visual_blocks
        skip = magcache.decide() if magcache is not None else False
        if skip:  # magcache_utils.py:78-79: bf16 add of the cached residual of this slot
            residual = magcache.residual_cache[magcache.cnt % 2]
            vis = _r(vis + residual, mode)
        else:
            ori = vis
            for i in range(nvb):
                vis = decoder_block(sd, f"visual_transformer_blocks.{i}", vis, text, temb, vcos, vsin,
                                    cfg, mode, sparse_params, taps)
                if taps is not None:
                    taps.setdefault("visual_blocks", []).append(vis)
            residual = _r(vis - ori, mode) if magcache is not None else None
        if magcache is not None:  # :86-100
            magcache.residual_cache[magcache.cnt % 2] = residual
            magcache.ran_blocks.append(0 if skip else 1)
            magcache.advance()
        # after_blocks dit.py:149-153
        if to_fractal:
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(perm.numel())
            vis = vis[inv]
        o = out_layer(sd, vis, temb, cfg, mode)
        return unpatchify(o.reshape(Tp, Hp, Wp, -1), cfg.patch_size)


# ------------------------------------------------------------------------------------------
# sampler
# ------------------------------------------------------------------------------------------
def sigma_schedule(num_steps: int, scheduler_scale: float) -> Tensor:
    """generation_utils.py:102-103."""
    t = torch.linspace(1, 0, num_steps + 1)
    return scheduler_scale * t / (1 + (scheduler_scale - 1) * t)


def get_sparse_params(attention: dict, latent_shape, patch_size):
    """generation_utils.py:10-36."""
    if attention.get("type") != "nabla":
        return None
    T, H, W = latent_shape[0] // patch_size[0], latent_shape[1] // patch_size[1], latent_shape[2] // patch_size[2]
    sta = fast_sta(T, H // 8, W // 8, attention["wT"], attention["wH"], attention["wW"])
    return {"sta_mask": sta, "to_fractal": True, "P": attention["P"], "visual_shape": (T, H, W)}


def get_velocity(sd, cfg, x, t, text_embeds, null_text_embeds, vpos, tpos, ntpos, guidance_weight,
                 scale_factor, sparse_params, mode, magcache=None):
    """generation_utils.py:39-77.  CFG combine on bf16 tensors in eager torch: each of
    (c-u), w*(.), u+(.) rounds to bf16."""
    v = dit_forward(sd, cfg, x, text_embeds["text_embeds"], text_embeds["pooled_embed"], t * 1000,
                    vpos, tpos, scale_factor, sparse_params, mode, magcache=magcache)
    if abs(guidance_weight - 1.0) > 1e-6:
        u = dit_forward(sd, cfg, x, null_text_embeds["text_embeds"], null_text_embeds["pooled_embed"],
                        t * 1000, vpos, ntpos, scale_factor, sparse_params, mode, magcache=magcache)
        v = _r(u + _r(guidance_weight * _r(v - u, mode), mode), mode)
    return v


def generate(sd, cfg, noise, num_steps, text_embeds, null_text_embeds, vpos, tpos, ntpos,
             guidance_weight, scheduler_scale, scale_factor=(1.0, 2.0, 2.0), attention=None,
             mode="fp32", return_trajectory=False, magcache=None):
    """generate generation_utils.py:80-129 with the initial noise passed in explicitly
    (the reference draws it from torch.Generator("cuda"), :97-99 — not reproducible off-CUDA).
    In bf16 mode `timestep_diff * pred_velocity` is (0-dim fp32 tensor) x (bf16 tensor) = bf16
    by torch type promotion, then promoted to fp32 by the add (:128)."""
    img = noise.clone().float()
    sparse = get_sparse_params(attention or {"type": "flash"}, img.shape, cfg.patch_size)
    sig = sigma_schedule(num_steps, scheduler_scale)
    traj = []
    for i in range(num_steps):
        t = sig[i].unsqueeze(0)
        dt = sig[i + 1] - sig[i]
        if cfg.visual_cond:
            x = torch.cat([img, torch.zeros_like(img), torch.zeros(*img.shape[:-1], 1)], dim=-1)
        else:
            x = img
        v = get_velocity(sd, cfg, x, t, text_embeds, null_text_embeds, vpos, tpos, ntpos,
                         guidance_weight, scale_factor, sparse, mode, magcache=magcache)
        img = img + _r(dt * v, mode)
        if return_trajectory:
            traj.append(img.clone())
    return (img, traj) if return_trajectory else img
