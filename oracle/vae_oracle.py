"""vae_oracle — CPU restatement of the HunyuanVideo 3D-VAE *decode* path of Kandinsky-5.

*** TEST INFRASTRUCTURE (same rules as oracle/k5_oracle.py): only tests/, __graft_entry__.smoke() and bench.py's
*** cpu_baseline leg may import it.

Own code on plain torch-CPU ops, operating on a state_dict (checkpoint layout).  Pinned against
vectors produced by the reference's own kandinsky/models/vae.py (oracle/gen_golden_vae.py ->
tests/golden/vae_tiny.safetensors).  The mid-block `Attention` is diffusers code that is not in the reference tree:
restated from its definition for the configuration instantiated at vae.py:312-322 — **parity unpinned** for that one
module beyond the reference's call site (mask, layout, residual).

mode="fp32": plain fp32.  mode="bf16": the CUDA-autocast behaviour of generation_utils.py:211-222 — conv / linear
operands and outputs bf16 (fp16 checkpoint weights are re-rounded to bf16 by autocast), GroupNorm and softmax in fp32,
eager bf16 adds / blends.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def _r(x: Tensor, mode: str) -> Tensor:
    return x.to(torch.bfloat16).float() if mode == "bf16" else x


def synthetic_decoder_state_dict(manifest: Dict[str, Sequence[int]], seed: int = 0) -> Dict[str, Tensor]:
    """Synthetic decoder weights laid out like the checkpoint (`manifest`: key -> shape, tests/golden/vae_meta.json
    "full_manifest").  One torch CPU generator per tensor seeded by (seed, index in the sorted key order), so the golden
    generator in the build container and the test on the GPU box make the same tensors.  Conv / linear weights ~ N(0, 1.5 /
    sqrt(fan_in)) rounded to fp16 (the checkpoint's dtype, vae.py:1279), norm weights N(1, 0.2), biases N(0, 0.1)."""
    sd = {}
    for idx, k in enumerate(sorted(manifest)):
        shape = tuple(manifest[k])
        g = torch.Generator().manual_seed(seed * 1000003 + idx)
        if "norm" in k and k.endswith("weight"):
            sd[k] = 1.0 + 0.2 * torch.randn(shape, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.1 * torch.randn(shape, generator=g)
        else:
            fan = 1
            for d in shape[1:]:
                fan *= d
            sd[k] = (torch.randn(shape, generator=g) * (1.5 / fan ** 0.5)).half().float()
    return sd


# ------------------------------------------------------------------------------------------ layers
def causal_conv3d(sd, name: str, x: Tensor, mode: str) -> Tensor:
    """HunyuanVideoCausalConv3d vae.py:125-163: replicate pad (W 1,1 ; H 1,1 ; T k-1,0) then Conv3d (stride 1)."""
    w, b = sd[name + ".conv.weight"].float(), sd[name + ".conv.bias"].float()
    k = w.shape[-1]
    if k > 1:
        x = F.pad(x, (k // 2, k // 2, k // 2, k // 2, k - 1, 0), mode="replicate")
    return _r(F.conv3d(_r(x, mode), _r(w, mode), _r(b, mode)), mode)


def group_norm(sd, name: str, x: Tensor, groups: int, eps: float = 1e-6) -> Tensor:
    """nn.GroupNorm (fp32 under autocast), vae.py:246,249,672."""
    return F.group_norm(x.float(), groups, sd[name + ".weight"].float(), sd[name + ".bias"].float(), eps)


def resnet_block(sd, p: str, x: Tensor, groups: int, mode: str) -> Tensor:
    """HunyuanVideoResnetBlockCausal3D.forward vae.py:257-275."""
    h = causal_conv3d(sd, p + ".conv1", F.silu(group_norm(sd, p + ".norm1", x, groups)), mode)
    h = causal_conv3d(sd, p + ".conv2", F.silu(group_norm(sd, p + ".norm2", h, groups)), mode)
    res = causal_conv3d(sd, p + ".conv_shortcut", x, mode) if (p + ".conv_shortcut.conv.weight") in sd else x
    return _r(h + res, mode)


def upsample(sd, p: str, x: Tensor, factor: Sequence[int], mode: str) -> Tensor:
    """HunyuanVideoUpsampleCausal3D.forward vae.py:187-205: first frame x(fh,fw), other frames x(ft,fh,fw), nearest."""
    ft, fh, fw = factor
    first, rest = x[:, :, :1], x[:, :, 1:]
    first = F.interpolate(first.squeeze(2), scale_factor=(fh, fw), mode="nearest").unsqueeze(2)
    if rest.shape[2] > 0:
        rest = F.interpolate(rest.contiguous(), scale_factor=(ft, fh, fw), mode="nearest")
        x = torch.cat([first, rest], dim=2)
    else:
        x = first
    return causal_conv3d(sd, p + ".conv", x, mode)


def causal_attention_mask(f: int, s: int) -> Tensor:
    """prepare_causal_attention_mask vae.py:110-122: (f*s, f*s) additive mask, 0 where key frame <= query frame."""
    fr = torch.arange(f).repeat_interleave(s)
    return torch.where(fr[None, :] <= fr[:, None], 0.0, float("-inf"))


def mid_attention(sd, p: str, x: Tensor, groups: int, mode: str) -> Tensor:
    """vae.py:343-359 call site + diffusers Attention (1 head of dim C, residual_connection=True)."""
    B, C, T, H, W = x.shape
    tok = x.permute(0, 2, 3, 4, 1).reshape(B, T * H * W, C)
    h = F.group_norm(tok.float().transpose(1, 2), groups, sd[p + ".group_norm.weight"].float(),
                     sd[p + ".group_norm.bias"].float(), 1e-6).transpose(1, 2)

    def lin(n, t):
        return _r(_r(t, mode) @ _r(sd[f"{p}.{n}.weight"].float(), mode).t() + _r(sd[f"{p}.{n}.bias"].float(), mode), mode)

    q, k, v = lin("to_q", h), lin("to_k", h), lin("to_v", h)
    mask = causal_attention_mask(T, H * W)
    s = (q @ k.transpose(1, 2)) / math.sqrt(C) + mask
    o = _r(torch.softmax(s, dim=-1) @ v, mode)
    o = lin("to_out.0", o)
    out = _r(o + tok, mode)
    return out.reshape(B, T, H, W, C).permute(0, 4, 1, 2, 3)


def decoder_forward(sd, z: Tensor, cfg: dict, mode: str = "fp32") -> Tensor:
    """post_quant_conv (vae.py:870) + HunyuanVideoDecoder3D.forward vae.py:684-696; z (B,C,T,H,W)."""
    G = cfg["norm_num_groups"]
    boc = list(cfg["block_out_channels"])
    h = _r(F.conv3d(_r(z.float(), mode), _r(sd["post_quant_conv.weight"].float(), mode),
                    _r(sd["post_quant_conv.bias"].float(), mode)), mode)
    h = causal_conv3d(sd, "decoder.conv_in", h, mode)
    h = resnet_block(sd, "decoder.mid_block.resnets.0", h, G, mode)
    h = mid_attention(sd, "decoder.mid_block.attentions.0", h, G, mode)
    h = resnet_block(sd, "decoder.mid_block.resnets.1", h, G, mode)
    n = len(boc)
    n_sp = int(math.log2(cfg.get("spatial_compression_ratio", 8)))
    n_t = int(math.log2(cfg.get("temporal_compression_ratio", 4)))
    for i in range(n):  # up-block schedule vae.py:644-659
        for j in range(cfg["layers_per_block"] + 1):
            h = resnet_block(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, G, mode)
        sp = i < n_sp
        tm = (i >= n - 1 - n_t) and (i != n - 1)
        if sp or tm:
            h = upsample(sd, f"decoder.up_blocks.{i}.upsamplers.0", h, (2 if tm else 1, 2 if sp else 1, 2 if sp else 1), mode)
    h = F.silu(group_norm(sd, "decoder.conv_norm_out", h, G))
    return causal_conv3d(sd, "decoder.conv_out", h, mode)


# ------------------------------------------------------------------------------------------ tiling (vae.py:847-1273)
def blend(a: Tensor, b: Tensor, extent: int, dim: int, mode: str) -> Tensor:
    """blend_v / blend_h / blend_t vae.py:908-936 (in place on b). Eager torch: each op rounds in bf16 mode."""
    extent = min(a.shape[dim], b.shape[dim], extent)
    for y in range(extent):
        ia = [slice(None)] * 5
        ib = [slice(None)] * 5
        ia[dim] = a.shape[dim] - extent + y
        ib[dim] = y
        b[tuple(ib)] = _r(_r(a[tuple(ia)] * (1 - y / extent), mode) + _r(b[tuple(ib)] * (y / extent), mode), mode)
    return b


def get_dec_optimal_tiling(shape, temporal_table: Dict[int, Tuple[int, int]], spatial_table: Dict[int, Tuple[int, int]]):
    """get_dec_optimal_tiling / get_enc_optimal_tiling vae.py:1245-1273."""
    b, _, f, h, w = shape
    nf, H, W = 4 * (f - 1) + 1, 8 * h, 8 * w
    if math.sqrt(H * W) < 450 and nf <= 97:
        ft, fs = nf, nf
    else:
        ft, fs = temporal_table[nf]
    if math.sqrt(H * W) > 900:
        (ht, hs), (wt, ws) = spatial_table[H], spatial_table[W]
    else:
        ht, hs, wt, ws = H, H, W, W
    return (1, ft, ht, wt), (fs, hs, ws)


def tiled_decode(sd, z, cfg, tile, stride, mode="fp32", decode_tile=None):
    """AutoencoderKLHunyuanVideo._decode with apply_tiling(tile, stride) (vae.py:847-877, 1012-1094, 1144-1204,
    1230-1243), including the quirk that _decode derives tile_latent_min_width from tile_sample_STRIDE_width."""
    _, ft, ht, wt = tile
    fs, hs, ws = stride
    min_f, str_f = ft - 1, fs
    dec = decode_tile or (lambda t: decoder_forward(sd, t, cfg, mode))
    lat_min_h, lat_min_w, lat_str_h, lat_str_w = ht // 8, wt // 8, hs // 8, ws // 8

    def spatial(zz):
        _, _, _, H, W = zz.shape
        rows = []
        for i in range(0, H - lat_min_h + 1, lat_str_h):
            rows.append([dec(zz[:, :, :, i:i + lat_min_h, j:j + lat_min_w]).clone()
                         for j in range(0, W - lat_min_w + 1, lat_str_w)])
        out_rows = []
        for i, row in enumerate(rows):
            out = []
            for j, t in enumerate(row):
                if i > 0:
                    t = blend(rows[i - 1][j], t, ht - hs, 3, mode)
                if j > 0:
                    t = blend(row[j - 1], t, wt - ws, 4, mode)
                out.append(t[:, :, :, :(ht if i == len(rows) - 1 else hs), :(wt if j == len(row) - 1 else ws)])
            out_rows.append(torch.cat(out, dim=-1))
        return torch.cat(out_rows, dim=3)[:, :, :, :H * 8, :W * 8]

    def maybe_spatial(zz):
        if zz.shape[-1] > lat_min_w or zz.shape[-2] > lat_min_h:
            return spatial(zz)
        return dec(zz).clone()

    _, _, nf, H, W = z.shape
    lat_min_f, lat_str_f = min_f // 4, str_f // 4
    if nf > lat_min_f + 1:  # _temporal_tiled_decode
        row = []
        for i in range(0, nf - lat_min_f + 1, lat_str_f):
            d = maybe_spatial(z[:, :, i:i + lat_min_f + 1])
            row.append(d[:, :, 1:] if i > 0 else d)
        out = []
        for i, t in enumerate(row):
            if i > 0:
                t = blend(row[i - 1], t, min_f - str_f, 2, mode)
                out.append(t[:, :, :(min_f if i == len(row) - 1 else str_f)])
            else:
                out.append(t[:, :, :str_f + 1])
        return torch.cat(out, dim=2)[:, :, :(nf - 1) * 4 + 1]
    if W > (ws // 8) or H > lat_min_h:  # the reference compares width against the STRIDE-derived value
        return spatial(z)
    return dec(z)


def postprocess_uint8(images: Tensor) -> Tensor:
    """generation_utils.py:222."""
    return ((images.clamp(-1.0, 1.0) + 1.0) * 127.5).to(torch.uint8)


# ------------------------------------------------------------------------------------------ encode path
def causal_conv3d_strided(sd, name: str, x: Tensor, stride: Sequence[int], mode: str) -> Tensor:
    """HunyuanVideoDownsampleCausal3D vae.py:208-227 = HunyuanVideoCausalConv3d with a stride and padding 0 (the encoder
    passes downsample_padding=0, vae.py:560): the same replicate pad (W 1,1 ; H 1,1 ; T 2,0), then Conv3d(stride)."""
    w, b = sd[name + ".conv.weight"].float(), sd[name + ".conv.bias"].float()
    x = F.pad(x, (1, 1, 1, 1, 2, 0), mode="replicate")
    return _r(F.conv3d(_r(x, mode), _r(w, mode), _r(b, mode), stride=tuple(stride)), mode)


def down_schedule(n_blocks: int, n_sp: int = 3, n_t: int = 2):
    """per down block: stride (t, h, w) of its downsampler or None — HunyuanVideoEncoder3D.__init__ vae.py:522-566
    (temporal_compression_ratio 4: time is halved by the blocks i >= n-1-n_t that are not the last one)."""
    out = []
    for i in range(n_blocks):
        sp = i < n_sp
        tm = (i >= n_blocks - 1 - n_t) and (i != n_blocks - 1)
        out.append((2 if tm else 1, 2 if sp else 1, 2 if sp else 1) if (sp or tm) else None)
    return out


def encoder_forward(sd, x: Tensor, cfg: dict, mode: str = "fp32") -> Tensor:
    """HunyuanVideoEncoder3D.forward vae.py:574-586 + quant_conv (vae.py:808-809): x (B,3,T,H,W) -> moments
    (B, 2*latent_channels, (T-1)/4+1, H/8, W/8) = [mean | logvar]."""
    G = cfg["norm_num_groups"]
    boc = list(cfg["block_out_channels"])
    h = causal_conv3d(sd, "encoder.conv_in", x.float(), mode)
    for i, st in enumerate(down_schedule(len(boc))):
        for j in range(cfg["layers_per_block"]):
            h = resnet_block(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, G, mode)
        if st is not None:
            h = causal_conv3d_strided(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", h, st, mode)
    h = resnet_block(sd, "encoder.mid_block.resnets.0", h, G, mode)
    h = mid_attention(sd, "encoder.mid_block.attentions.0", h, G, mode)
    h = resnet_block(sd, "encoder.mid_block.resnets.1", h, G, mode)
    h = F.silu(group_norm(sd, "encoder.conv_norm_out", h, G))
    h = causal_conv3d(sd, "encoder.conv_out", h, mode)
    return _r(F.conv3d(_r(h, mode), _r(sd["quant_conv.weight"].float(), mode), _r(sd["quant_conv.bias"].float(), mode)), mode)


def gaussian_moments(h: Tensor):
    """diffusers DiagonalGaussianDistribution (NOT in the reference tree; restated from its definition — parity unpinned):
    mean, logvar = chunk(h, 2, dim=1); logvar clamped to [-30, 20]; std = exp(logvar / 2); mode() = mean;
    sample() = mean + std * N(0, 1)."""
    mean, logvar = torch.chunk(h, 2, dim=1)
    logvar = logvar.clamp(-30.0, 20.0)
    return mean, logvar, torch.exp(0.5 * logvar)


def tiled_encode(sd, x, cfg, tile, stride, mode="fp32", encode_tile=None):
    """AutoencoderKLHunyuanVideo._encode with apply_tiling(tile, stride): vae.py:795-810 (dispatch), 938-1010 (spatial
    tiles, blends on the LATENT grid), 1096-1142 (temporal tiles of min_frames + 1 frames, first latent frame of every
    later tile dropped, blend_t)."""
    _, ft, ht, wt = tile
    fs, hs, ws = stride
    min_f, str_f = ft - 1, fs
    enc = encode_tile or (lambda t: encoder_forward(sd, t, cfg, mode))
    lat_min_h, lat_min_w, lat_str_h, lat_str_w = ht // 8, wt // 8, hs // 8, ws // 8

    def spatial(xx):
        _, _, _, H, W = xx.shape
        rows = []
        for i in range(0, H - ht + 1, hs):
            rows.append([enc(xx[:, :, :, i:i + ht, j:j + wt]).clone() for j in range(0, W - wt + 1, ws)])
        out_rows = []
        for i, row in enumerate(rows):
            out = []
            for j, t in enumerate(row):
                if i > 0:
                    t = blend(rows[i - 1][j], t, lat_min_h - lat_str_h, 3, mode)
                if j > 0:
                    t = blend(row[j - 1], t, lat_min_w - lat_str_w, 4, mode)
                out.append(t[:, :, :, :(lat_min_h if i == len(rows) - 1 else lat_str_h), :(lat_min_w if j == len(row) - 1 else lat_str_w)])
            out_rows.append(torch.cat(out, dim=4))
        return torch.cat(out_rows, dim=3)[:, :, :, :H // 8, :W // 8]

    _, _, nf, H, W = x.shape
    if nf > min_f + 1:   # _temporal_tiled_encode
        lat_nf = (nf - 1) // 4 + 1
        lat_min_f, lat_str_f = min_f // 4, str_f // 4
        row = []
        for i in range(0, nf - min_f + 1, str_f):
            t = x[:, :, i:i + min_f + 1]
            t = spatial(t) if (H > ht or W > wt) else enc(t).clone()
            row.append(t[:, :, 1:] if i > 0 else t)
        out = []
        for i, t in enumerate(row):
            if i > 0:
                t = blend(row[i - 1], t, lat_min_f - lat_str_f, 2, mode)
                out.append(t[:, :, :(lat_min_f if i == len(row) - 1 else lat_str_f)])
            else:
                out.append(t[:, :, :lat_str_f + 1])
        return torch.cat(out, dim=2)[:, :, :lat_nf]
    if W > wt or H > ht:
        return spatial(x)
    return enc(x)
