"""AutoencoderKLHunyuanVideo — host mirror of kandinsky/models/vae.py (reference): decode (the T2V hot path) and encode
(image / video conditioning).

Keeps what callers touch: `build_vae(conf)` (vae.py:1276-1282), `.decode(z).sample`, `.config.scaling_factor`, `.eval()`,
`.to()`, the checkpoint key names of `decoder.*` / `post_quant_conv.*`, and the reference's tiling
policy — `get_dec_optimal_tiling` tables (data in vae_tiling.json), temporal tiles with a dropped first frame, optional
spatial tiles, linear cross-fades, including the quirk that `_decode` compares the width with the STRIDE-derived tile
width (vae.py:854-856).  The arithmetic runs in libk5.so: `k5_vae_decode_tile` per tile, `k5_blend_bf16` for the
cross-fades; torch only slices / concatenates.  The encoder half (`encode`, `tiled_encode`, `_temporal_tiled_encode`,
vae.py:795-845, 938-1010, 1096-1142) works the same way through `k5_vae_encode_tile`; its tensors (`encoder.*`,
`quant_conv.*`) are optional in a checkpoint load — T2V itself never encodes.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
from types import SimpleNamespace

import torch
from torch import nn

from .. import _engine as E

_HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(_HERE, "vae_tiling.json")) as _f:
    _T = json.load(_f)
OPT_TEMPORAL_TILING = {int(k): tuple(v) for k, v in _T["temporal"].items()}
OPT_SPATIAL_TILING = {int(k): tuple(v) for k, v in _T["spatial"].items()}


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class DiagonalGaussianDistribution:
    """diffusers' class of that name (third-party, not in the reference tree; restated from its definition): the encoder's
    moments tensor = [mean | logvar] along the channel axis, logvar clamped to [-30, 20]."""

    def __init__(self, parameters):
        self.parameters = parameters                         # moments keep the encoder's dtype (bf16 from the engine), as in diffusers
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator=None):
        eps = torch.randn(self.mean.shape, generator=generator, device=self.mean.device if generator is None else generator.device,
                          dtype=self.mean.dtype).to(self.mean.device)
        return self.mean + self.std * eps

    def mode(self):
        return self.mean


class AutoencoderKLOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


def encoder_manifest(in_channels, latent_channels, block_out_channels, layers_per_block):
    """state_dict names / shapes of encoder + quant_conv (vae.py:478-572, 747)."""
    m = {}
    boc = list(block_out_channels)

    def conv(n, o, i, k):
        m[n + ".weight"], m[n + ".bias"] = (o, i, k, k, k), (o,)

    def norm(n, c):
        m[n + ".weight"], m[n + ".bias"] = (c,), (c,)

    def resnet(p, i, o):
        norm(p + ".norm1", i); conv(p + ".conv1.conv", o, i, 3); norm(p + ".norm2", o); conv(p + ".conv2.conv", o, o, 3)
        if i != o:
            conv(p + ".conv_shortcut.conv", o, i, 1)

    conv("encoder.conv_in.conv", boc[0], in_channels, 3)
    prev = boc[0]
    for i, outc in enumerate(boc):
        for j in range(layers_per_block):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else outc, outc)
        if i < len(boc) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv.conv", outc, outc, 3)
        prev = outc
    top = boc[-1]
    resnet("encoder.mid_block.resnets.0", top, top)
    a = "encoder.mid_block.attentions.0."
    norm(a + "group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        m[a + n + ".weight"], m[a + n + ".bias"] = (top, top), (top,)
    resnet("encoder.mid_block.resnets.1", top, top)
    norm("encoder.conv_norm_out", top)
    conv("encoder.conv_out.conv", 2 * latent_channels, top, 3)
    conv("quant_conv", 2 * latent_channels, 2 * latent_channels, 1)
    return m


def decoder_manifest(latent_channels, out_channels, block_out_channels, layers_per_block):
    """state_dict names / shapes of post_quant_conv + decoder (vae.py:589-696, 748)."""
    m = {}
    boc = list(block_out_channels)

    def conv(n, o, i, k):
        m[n + ".weight"], m[n + ".bias"] = (o, i, k, k, k), (o,)

    def norm(n, c):
        m[n + ".weight"], m[n + ".bias"] = (c,), (c,)

    def resnet(p, i, o):
        norm(p + ".norm1", i); conv(p + ".conv1.conv", o, i, 3); norm(p + ".norm2", o); conv(p + ".conv2.conv", o, o, 3)
        if i != o:
            conv(p + ".conv_shortcut.conv", o, i, 1)

    conv("post_quant_conv", latent_channels, latent_channels, 1)
    top = boc[-1]
    conv("decoder.conv_in.conv", top, latent_channels, 3)
    resnet("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0."
    norm(a + "group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        m[a + n + ".weight"], m[a + n + ".bias"] = (top, top), (top,)
    resnet("decoder.mid_block.resnets.1", top, top)
    prev = top
    for i, outc in enumerate(reversed(boc)):
        for j in range(layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else outc, outc)
        if i < len(boc) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv.conv", outc, outc, 3)
        prev = outc
    norm("decoder.conv_norm_out", boc[0])
    conv("decoder.conv_out.conv", out_channels, boc[0], 3)
    return m


class _Node(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("VAE parameter container: the decoder runs in the HIP engine (AutoencoderKLHunyuanVideo.decode)")


class AutoencoderKLHunyuanVideo(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, latent_channels=16, down_block_types=None, up_block_types=None,
                 block_out_channels=(128, 256, 512, 512), layers_per_block=2, act_fn="silu", norm_num_groups=32,
                 scaling_factor=0.476986, spatial_compression_ratio=8, temporal_compression_ratio=4,
                 mid_block_add_attention=True, **_ignored):
        super().__init__()
        if spatial_compression_ratio != 8 or temporal_compression_ratio != 4 or not mid_block_add_attention:
            raise NotImplementedError("only the HunyuanVideo 8x/4x decoder with mid-block attention is built")
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, latent_channels=latent_channels,
                                      block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                      norm_num_groups=norm_num_groups, scaling_factor=scaling_factor,
                                      spatial_compression_ratio=8, temporal_compression_ratio=4)
        names = dict(decoder_manifest(latent_channels, out_channels, block_out_channels, layers_per_block))
        self._encoder_keys = list(encoder_manifest(in_channels, latent_channels, block_out_channels, layers_per_block))
        names.update(encoder_manifest(in_channels, latent_channels, block_out_channels, layers_per_block))
        self._encoder_loaded = None     # None: never went through load_state_dict (parameters filled by the caller) — everything is real;
        # False: a decode-only checkpoint left the encoder.* / quant_conv.* parameters as placeholders; True: loaded
        for name, shape in names.items():
            node = self
            parts = name.split(".")
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
            node.register_parameter(parts[-1], nn.Parameter(torch.empty(shape), requires_grad=False))
        self.spatial_compression_ratio, self.temporal_compression_ratio = 8, 4
        self.use_tiling = self.use_framewise_decoding = True
        # defaults of vae.py:765-771, overwritten by apply_tiling
        self.tile_sample_min_height = self.tile_sample_min_width = 256
        self.tile_sample_min_num_frames = 16
        self.tile_sample_stride_height = self.tile_sample_stride_width = 192
        self.tile_sample_stride_num_frames = 12
        self.tile_size = None
        self._handle, self._handle_device = None, None

    # ------------------------------------------------------------------ loading
    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, **kw):
        """diffusers-style folder: <path>/<subfolder>/config.json + diffusion_pytorch_model.safetensors (vae.py:1278)."""
        from safetensors.torch import load_file
        root = os.path.join(path, subfolder) if subfolder and os.path.isdir(os.path.join(path, subfolder)) else path
        with open(os.path.join(root, "config.json")) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        model = cls(**cfg)
        files = [f for f in sorted(os.listdir(root)) if f.endswith(".safetensors")]
        if not files:
            raise FileNotFoundError(f"no .safetensors checkpoint under {root}")
        sd = {}
        for f in files:
            sd.update(load_file(os.path.join(root, f)))
        want = model.state_dict().keys()
        sd = {k: (v.to(torch_dtype) if torch_dtype is not None else v) for k, v in sd.items() if k in want}
        model.load_state_dict(sd, assign=True)
        return model

    def load_state_dict(self, state_dict, strict=True, assign=False):
        """decoder.* / post_quant_conv.* are required; encoder.* / quant_conv.* are optional as a group (a decode-only load
        leaves `encode` unavailable, loudly)."""
        state_dict = {k: v for k, v in state_dict.items() if k in self.state_dict()}
        enc = set(self._encoder_keys)
        have = [k for k in state_dict if k in enc]
        if have and len(have) != len(enc):
            raise RuntimeError(f"VAE checkpoint has only {len(have)} of the {len(enc)} encoder tensors")
        self._encoder_loaded = bool(have)
        if not have:   # strict load of the decoder half only
            missing = [k for k in self.state_dict() if k not in enc and k not in state_dict]
            if strict and missing:
                raise RuntimeError(f"missing VAE decoder keys: {missing[:6]}")
            out = super().load_state_dict(state_dict, strict=False, assign=assign)
        else:
            out = super().load_state_dict(state_dict, strict=strict, assign=assign)
        self._drop_engine()
        return out

    def _drop_engine(self):
        if self._handle is not None:
            E.lib().k5_vae_destroy(self._handle)
        self._handle = None

    def _engine(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("the VAE decoder runs on the MI355X engine only (no CPU fallback)")
        if self._handle is not None and self._handle_device == device:
            return self._handle
        self._drop_engine()
        c = self.config
        cc = E.VaeConfig(c.latent_channels, c.out_channels, (C.c_int * 4)(*c.block_out_channels), c.layers_per_block,
                         c.norm_num_groups)
        h = C.c_void_p()
        with torch.cuda.device(device):
            E.check(E.lib().k5_vae_create(C.byref(cc), C.byref(h)), "k5_vae_create")
            for name, t in self.state_dict().items():
                if self._encoder_loaded is False and name in self._encoder_keys:
                    continue      # placeholders of a decode-only load
                t = t.detach()
                if t.dtype not in (torch.float32, torch.bfloat16, torch.float16):
                    t = t.float()
                t = t.contiguous()
                shape = (C.c_int64 * t.dim())(*t.shape)
                E.check(E.lib().k5_vae_load_tensor(h, name.encode(), t.data_ptr(), E.k5_dtype(t), shape, t.dim()), name)
            E.check(E.lib().k5_vae_finalize(h), "k5_vae_finalize")
        self._handle, self._handle_device = h, device
        return h

    # ------------------------------------------------------------------ engine calls
    def _decode_tile(self, z):
        """z (1,C,t,h,w) -> (1,3,4(t-1)+1,8h,8w) bf16 = decoder(post_quant_conv(z)).  A temporal slice of a longer fp32 latent (the tiling loop's
        z[:, :, i : i + t]) is read in place through its channel stride (k5_vae_decode_tile_strided): no copy, no torch kernel."""
        h = self._engine(z.device)
        zz = z[0]
        _, t, hh, ww = zz.shape
        in_place = zz.dtype == torch.float32 and zz.stride()[1:] == (hh * ww, ww, 1) and zz.stride(0) >= t * hh * ww
        if not in_place:
            zz = zz.float().contiguous()
        out = torch.empty(1, self.config.out_channels, 4 * (t - 1) + 1, 8 * hh, 8 * ww, dtype=torch.bfloat16, device=z.device)
        with torch.cuda.device(z.device):
            E.check(E.lib().k5_vae_decode_tile_strided(h, zz.data_ptr(), int(zz.stride(0)), t, hh, ww, out.data_ptr(), E.stream_ptr(z.device)),
                    "k5_vae_decode_tile_strided")
        return out

    def _encode_tile(self, x):
        """x (1,3,T,H,W) -> moments (1, 2*latent, (T-1)/4+1, H/8, W/8) bf16 = quant_conv(encoder(x))."""
        h = self._engine(x.device)
        if self._encoder_loaded is False or not E.lib().k5_vae_has_encoder(h):
            raise RuntimeError("this VAE was loaded without its encoder.* / quant_conv.* tensors: encode() is unavailable")
        xx = x[0].float().contiguous()
        _, t, hh, ww = xx.shape
        out = torch.empty(1, 2 * self.config.latent_channels, (t - 1) // 4 + 1, hh // 8, ww // 8, dtype=torch.bfloat16, device=x.device)
        with torch.cuda.device(x.device):
            E.check(E.lib().k5_vae_encode_tile(h, xx.data_ptr(), t, hh, ww, out.data_ptr(), E.stream_ptr(x.device)), "k5_vae_encode_tile")
        return out

    @staticmethod
    def _blend(a, b, extent, dim):
        """blend_t / blend_v / blend_h (vae.py:908-936): cross-fade the first `extent` slices of b with the last of a."""
        extent = min(a.shape[dim], b.shape[dim], extent)
        if extent <= 0:
            return b
        a, bc = a.contiguous(), b.contiguous()
        outer = math.prod(b.shape[:dim])
        inner = math.prod(b.shape[dim + 1:])
        if math.prod(a.shape[:dim]) != outer or math.prod(a.shape[dim + 1:]) != inner:
            raise ValueError("blend: tiles differ outside the blended axis")
        E.check(E.lib().k5_blend_bf16(a.data_ptr(), bc.data_ptr(), outer, a.shape[dim], b.shape[dim], inner, extent,
                                      E.stream_ptr(b.device)), "k5_blend_bf16")
        return bc

    # ------------------------------------------------------------------ tiling policy (reference semantics)
    def get_enc_optimal_tiling(self, shape):
        _, _, num_frames, height, width = shape
        if math.sqrt(height * width) < 450 and num_frames <= 97:
            ft = fs = num_frames
        else:
            ft, fs = OPT_TEMPORAL_TILING[num_frames]
        if math.sqrt(height * width) > 900:
            (ht, hs), (wt, ws) = OPT_SPATIAL_TILING[height], OPT_SPATIAL_TILING[width]
        else:
            ht, hs, wt, ws = height, height, width, width
        return (1, ft, ht, wt), (fs, hs, ws)

    def get_dec_optimal_tiling(self, shape):
        b, _, f, h, w = shape
        return self.get_enc_optimal_tiling([b, 3, 4 * (f - 1) + 1, 8 * h, 8 * w])

    def apply_tiling(self, tile, stride):
        _, ft, ht, wt = tile
        fs, hs, ws = stride
        self.use_tiling = True
        self.tile_sample_min_num_frames, self.tile_sample_stride_num_frames = ft - 1, fs
        self.tile_sample_min_height, self.tile_sample_min_width = ht, wt
        self.tile_sample_stride_height, self.tile_sample_stride_width = hs, ws

    def tiled_decode(self, z, return_dict=True):
        _, _, _, H, W = z.shape
        mh, mw = self.tile_sample_min_height // 8, self.tile_sample_min_width // 8
        sh, sw = self.tile_sample_stride_height // 8, self.tile_sample_stride_width // 8
        bh = self.tile_sample_min_height - self.tile_sample_stride_height
        bw = self.tile_sample_min_width - self.tile_sample_stride_width
        rows = [[self._decode_tile(z[:, :, :, i:i + mh, j:j + mw]) for j in range(0, W - mw + 1, sw)]
                for i in range(0, H - mh + 1, sh)]
        out_rows = []
        for i, row in enumerate(rows):
            out = []
            for j, tile in enumerate(row):
                if i > 0:
                    tile = self._blend(rows[i - 1][j], tile, bh, 3)
                if j > 0:
                    tile = self._blend(row[j - 1], tile, bw, 4)
                rows[i][j] = tile
                hl = self.tile_sample_min_height if i == len(rows) - 1 else self.tile_sample_stride_height
                wl = self.tile_sample_min_width if j == len(row) - 1 else self.tile_sample_stride_width
                out.append(tile[:, :, :, :hl, :wl])
            out_rows.append(torch.cat(out, dim=-1))
        dec = torch.cat(out_rows, dim=3)[:, :, :, :H * 8, :W * 8]
        return DecoderOutput(dec) if return_dict else (dec,)

    def _temporal_tiled_decode(self, z, return_dict=True):
        nf = z.shape[2]
        mh, mw = self.tile_sample_min_height // 8, self.tile_sample_min_width // 8
        mf, sf = self.tile_sample_min_num_frames // 4, self.tile_sample_stride_num_frames // 4
        bf = self.tile_sample_min_num_frames - self.tile_sample_stride_num_frames
        starts = list(range(0, nf - mf + 1, sf))

        def decode_one(i):
            tile = z[:, :, i:i + mf + 1]
            if self.use_tiling and (tile.shape[-1] > mw or tile.shape[-2] > mh):
                return self.tiled_decode(tile).sample
            return self._decode_tile(tile)

        tp = getattr(self, "_tile_parallel", None)
        # Round 6 — one GPU, contiguous tiles at least two cross-fades long: tile k is decoded, cross-faded with the tail of tile k - 1 and written
        # into its frames of the output in ONE pass (k5_blend_place_bf16); the slices, .contiguous() copies and torch.cat of the loop below are gone
        # (the same numbers: blend_t only ever reads the previous tile's LAST bf frames, which its own cross-fade — the first bf — did not touch).
        sn, mn = self.tile_sample_stride_num_frames, self.tile_sample_min_num_frames
        if (tp is None or tp[1] <= 1) and len(starts) > 1 and mn >= 2 * bf and os.environ.get("K5_VAE_LEGACY_GLUE", "0") != "1":
            total = (nf - 1) * 4 + 1
            first = decode_one(starts[0])
            _, C3, F0, Ho, Wo = first.shape
            if first.is_cuda and first.dtype == torch.bfloat16 and first.is_contiguous() and F0 == mn + 1 and (Ho * Wo) % 8 == 0:
                dec = torch.empty(1, C3, sn + 1 + sn * (len(starts) - 2) + mn, Ho, Wo, dtype=first.dtype, device=first.device)
                inner, stream = Ho * Wo, E.stream_ptr(first.device)

                def place(prev, cur, drop, f0, keep):   # frames [drop, drop + keep) of cur -> dec[:, :, f0 : f0 + keep], the first bf cross-faded with prev's tail
                    a_ptr = 0 if prev is None else prev[0].data_ptr() + prev[1] * inner * 2
                    E.check(E.lib().k5_blend_place_bf16(a_ptr, 0 if prev is None else prev[0].stride(1), 0 if prev is None else prev[0].shape[2] - prev[1],
                                                        cur.data_ptr() + drop * inner * 2, cur.stride(1), dec.data_ptr() + f0 * inner * 2, dec.stride(1),
                                                        C3, inner, 0 if prev is None else bf, keep, stream), "k5_blend_place_bf16")
                with torch.cuda.device(first.device):
                    place(None, first, 0, 0, sn + 1)
                    prev, f0 = (first, 0), sn + 1
                    for k in range(1, len(starts)):
                        cur = decode_one(starts[k])
                        keep = mn if k == len(starts) - 1 else sn
                        place(prev, cur, 1, f0, keep)
                        prev, f0 = (cur, 1), f0 + keep
                dec = dec[:, :, :total]
                return DecoderOutput(dec) if return_dict else (dec,)
            decoded = [first] + [decode_one(i) for i in starts[1:]]
        elif tp is not None and tp[1] > 1 and len(starts) > 1:
            # a rank without a tile in a round takes part in that round's gather with zeros of a tile's shape (round 6: also in the FIRST round — a
            # 1 s clip has 2 temporal tiles, a node has 4 or 8 ranks; the launch-contract test found ranks 2 and 3 refusing to decode)
            tile_shape = (z.shape[0], self.config.out_channels, 4 * mf + 1, 8 * z.shape[-2], 8 * z.shape[-1])
            decoded = self._decode_tiles_distributed(starts, decode_one, *tp, tile_shape=tile_shape, device=z.device)
        else:
            decoded = [decode_one(i) for i in starts]
        row = [d[:, :, 1:] if k > 0 else d for k, d in enumerate(decoded)]
        out = []
        for i, tile in enumerate(row):
            if i > 0:
                tile = self._blend(row[i - 1], tile, bf, 2)
                row[i] = tile
                out.append(tile[:, :, :(self.tile_sample_min_num_frames if i == len(row) - 1 else self.tile_sample_stride_num_frames)])
            else:
                out.append(tile[:, :, :self.tile_sample_stride_num_frames + 1])
        dec = torch.cat(out, dim=2)[:, :, :(nf - 1) * 4 + 1]
        return DecoderOutput(dec) if return_dict else (dec,)

    def enable_tile_parallel(self, rank, world, group=None):
        """Temporal tiles are independent until blend_t: tile k is decoded by rank k % world, one
        all-gather per round of `world` tiles hands every rank all decoded tiles (40 MB each at 512x768), the cheap
        cross-fades are then replicated.  Collective: every rank must call decode() with the same latent."""
        self._tile_parallel = (int(rank), int(world), group)
        return self

    @staticmethod
    def _decode_tiles_distributed(starts, decode_one, rank, world, group, tile_shape=None, device=None):
        import torch.distributed as dist
        decoded = [None] * len(starts)
        for r0 in range(0, len(starts), world):
            k = r0 + rank
            mine = decode_one(starts[k]) if k < len(starts) else None
            if mine is None:   # ragged round: take part in the collective with a dummy of the right shape
                if decoded[0] is not None:
                    mine = torch.zeros_like(decoded[0])
                elif tile_shape is not None:
                    mine = torch.zeros(tile_shape, dtype=torch.bfloat16, device=device)
            if mine is None:
                raise RuntimeError("tile-parallel decode: a rank without a tile needs the tile shape")
            if dist.get_backend(group) == "gloo":   # IPC transport (K5_SP_TRANSPORT=ipc): the process group is a host-side one
                parts = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(world)]
                dist.all_gather(parts, mine.contiguous().cpu(), group=group)
                buf = torch.stack(parts).to(mine.device)
            else:
                buf = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
                dist.all_gather_into_tensor(buf.view((world * mine.shape[0],) + tuple(mine.shape[1:])), mine.contiguous(), group=group)
            for j in range(min(world, len(starts) - r0)):
                decoded[r0 + j] = buf[j]
        return decoded

    def _decode(self, z, return_dict=True):
        _, _, nf, H, W = z.shape
        mh = self.tile_sample_min_height // 8
        mw = self.tile_sample_stride_width // 8   # sic: the reference derives this one from the stride (vae.py:854-856)
        mf = self.tile_sample_min_num_frames // 4
        if self.use_framewise_decoding and nf > mf + 1:
            return self._temporal_tiled_decode(z, return_dict)
        if self.use_tiling and (W > mw or H > mh):
            return self.tiled_decode(z, return_dict)
        dec = self._decode_tile(z)
        return DecoderOutput(dec) if return_dict else (dec,)

    @torch.no_grad()
    def decode(self, z, return_dict=True):
        """vae.py:879-906."""
        if z.shape[0] != 1:
            outs = [self.decode(z[i:i + 1]).sample for i in range(z.shape[0])]
            dec = torch.cat(outs, 0)
            return DecoderOutput(dec) if return_dict else (dec,)
        tile_size, tile_stride = self.get_dec_optimal_tiling(z.shape)
        if tile_size != self.tile_size:
            self.tile_size = tile_size
            self.apply_tiling(tile_size, tile_stride)
        dec = self._decode(z).sample
        return DecoderOutput(dec) if return_dict else (dec,)

    # ------------------------------------------------------------------ encode (vae.py:795-845, 938-1010, 1096-1142)
    def tiled_encode(self, x):
        _, _, _, H, W = x.shape
        mh, mw = self.tile_sample_min_height, self.tile_sample_min_width
        sh, sw = self.tile_sample_stride_height, self.tile_sample_stride_width
        lmh, lmw, lsh, lsw = mh // 8, mw // 8, sh // 8, sw // 8
        rows = [[self._encode_tile(x[:, :, :, i:i + mh, j:j + mw]) for j in range(0, W - mw + 1, sw)]
                for i in range(0, H - mh + 1, sh)]
        out_rows = []
        for i, row in enumerate(rows):
            out = []
            for j, tile in enumerate(row):
                if i > 0:
                    tile = self._blend(rows[i - 1][j], tile, lmh - lsh, 3)
                if j > 0:
                    tile = self._blend(row[j - 1], tile, lmw - lsw, 4)
                rows[i][j] = tile
                out.append(tile[:, :, :, :(lmh if i == len(rows) - 1 else lsh), :(lmw if j == len(row) - 1 else lsw)])
            out_rows.append(torch.cat(out, dim=4))
        return torch.cat(out_rows, dim=3)[:, :, :, :H // 8, :W // 8]

    def _temporal_tiled_encode(self, x):
        _, _, nf, H, W = x.shape
        lat_nf = (nf - 1) // 4 + 1
        mf, sf = self.tile_sample_min_num_frames, self.tile_sample_stride_num_frames
        lmf, lsf = mf // 4, sf // 4
        row = []
        for i in range(0, nf - mf + 1, sf):
            tile = x[:, :, i:i + mf + 1]
            if self.use_tiling and (H > self.tile_sample_min_height or W > self.tile_sample_min_width):
                tile = self.tiled_encode(tile)
            else:
                tile = self._encode_tile(tile)
            row.append(tile[:, :, 1:] if i > 0 else tile)
        out = []
        for i, tile in enumerate(row):
            if i > 0:
                tile = self._blend(row[i - 1], tile, lmf - lsf, 2)
                row[i] = tile
                out.append(tile[:, :, :(lmf if i == len(row) - 1 else lsf)])
            else:
                out.append(tile[:, :, :lsf + 1])
        return torch.cat(out, dim=2)[:, :, :lat_nf]

    def _encode(self, x):
        _, _, nf, H, W = x.shape
        if self.use_framewise_decoding and nf > self.tile_sample_min_num_frames + 1:
            return self._temporal_tiled_encode(x)
        if self.use_tiling and (W > self.tile_sample_min_width or H > self.tile_sample_min_height):
            return self.tiled_encode(x)
        return self._encode_tile(x)

    @torch.no_grad()
    def encode(self, x, opt_tiling=True, return_dict=True):
        """vae.py:813-845: x (B,3,F,H,W) in [-1,1] -> AutoencoderKLOutput(latent_dist=DiagonalGaussianDistribution)."""
        if x.shape[0] != 1:     # the engine encodes one sample at a time; the caller's tiling choice goes with every sample
            h = torch.cat([self.encode(x[i:i + 1], opt_tiling=opt_tiling).latent_dist.parameters for i in range(x.shape[0])], 0)
        else:
            if opt_tiling:
                tile_size, tile_stride = self.get_enc_optimal_tiling(x.shape)
            else:
                b, _, f, hh, ww = x.shape
                tile_size, tile_stride = (b, f, hh, ww), (f, hh, ww)
            if tile_size != self.tile_size:
                self.tile_size = tile_size
                self.apply_tiling(tile_size, tile_stride)
            h = self._encode(x)
        posterior = DiagonalGaussianDistribution(h)
        return AutoencoderKLOutput(posterior) if return_dict else (posterior,)

    def forward(self, sample, sample_posterior=False, return_dict=True, generator=None):
        """vae.py:1206-1228: encode -> mode / sample -> decode."""
        posterior = self.encode(sample).latent_dist
        z = posterior.sample(generator=generator) if sample_posterior else posterior.mode()
        return self.decode(z, return_dict=return_dict)


def build_vae(conf):
    """reference vae.py:1276-1282"""
    name = conf["name"] if isinstance(conf, dict) else conf.name
    path = conf["checkpoint_path"] if isinstance(conf, dict) else conf.checkpoint_path
    if name == "hunyuan":
        return AutoencoderKLHunyuanVideo.from_pretrained(path, subfolder="vae", torch_dtype=torch.float16)
    assert False, f"unknown vae name {name}"
