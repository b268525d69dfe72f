"""Golden vectors for the HunyuanVideo VAE *encode* path (VAE encoder / I2V conditioning), generated FROM
THE REFERENCE's own kandinsky/models/vae.py (HunyuanVideoEncoder3D, DownBlock3D, DownsampleCausal3D, _encode, tiled_encode,
_temporal_tiled_encode, get_enc_optimal_tiling) on CPU in fp32, under the diffusers shims of oracle/gen_golden_vae.py.

TEST INFRASTRUCTURE — run once in the build container:   PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_vae_enc.py

`DiagonalGaussianDistribution` is diffusers code (not in the reference tree): the goldens stop at the moments tensor
h = quant_conv(encoder(x)) that the reference hands to it (vae.py:839-841); its mean / logvar split is restated in the
oracle and the host mirror from its definition (parity unpinned for that one class, as for the mid-block Attention).
Only data is written (tests/golden/vae_enc_tiny.safetensors, vae_enc_meta.json)."""
import json
import os
import sys
import types

os.environ["TORCH_COMPILE_DISABLE"] = "1"
sys.dont_write_bytecode = True

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden_vae as G  # noqa: E402  (shims + reference path)

OUT = G.OUT


def main():
    G._install_shims()
    for name, sub in (("kandinsky", "/kandinsky"), ("kandinsky.models", "/kandinsky/models")):
        m = types.ModuleType(name)
        m.__path__ = [G.REF + sub]
        sys.modules[name] = m
    import kandinsky.models.vae as kvae

    torch.manual_seed(78)
    T = {}
    cfg = dict(latent_channels=16, block_out_channels=(16, 32, 32, 32), layers_per_block=2, norm_num_groups=8)
    vae = kvae.AutoencoderKLHunyuanVideo(**cfg).eval()
    with torch.no_grad():
        for n, p in vae.named_parameters():
            if not (n.startswith("encoder") or n.startswith("quant_conv")):
                continue
            if "norm" in n and n.endswith("weight"):
                p.normal_(1.0, 0.2)
            elif n.endswith("bias"):
                p.normal_(0, 0.1)
            else:
                p.normal_(0, 1.5 / (p[0].numel() ** 0.5))
        for k, v in vae.state_dict().items():
            if k.startswith("encoder.") or k.startswith("quant_conv"):
                T["w." + k] = v.clone()
        meta = {"config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}}
        enc = vae.encoder
        # ---- per-module vectors: the strided causal conv of the down blocks ----
        x = torch.randn(1, 16, 5, 9, 8)
        T["m.down_hw.x"] = x
        T["m.down_hw.out"] = enc.down_blocks[0].downsamplers[0](x)      # stride (1,2,2), odd height
        x2 = torch.randn(1, 32, 5, 8, 6)
        T["m.down_thw.x"] = x2
        T["m.down_thw.out"] = enc.down_blocks[1].downsamplers[0](x2)    # stride (2,2,2)
        x3 = torch.randn(1, 32, 1, 8, 6)
        T["m.down_thw1.x"] = x3
        T["m.down_thw1.out"] = enc.down_blocks[1].downsamplers[0](x3)   # a single frame
        T["m.block0.out"] = enc.down_blocks[0](x)                      # 2 resnets + downsample
        # ---- whole encoder ----
        v = torch.randn(1, 3, 9, 32, 24)
        T["e.x"] = v
        T["e.encoder"] = enc(v)
        T["e.moments"] = vae.quant_conv(T["e.encoder"])
        vae.apply_tiling((1, 9, 32, 24), (9, 32, 24))                  # one tile = the whole clip
        assert torch.equal(vae._encode(v), T["e.moments"])
        img = torch.randn(1, 3, 1, 16, 24)                             # a single image (I2V conditioning frame)
        T["e.img.x"] = img
        T["e.img.moments"] = vae.quant_conv(enc(img))
        # temporal tiling: 17 frames, tiles of 8(+1) frames, stride 4
        v17 = torch.randn(1, 3, 17, 16, 16)
        vae.apply_tiling((1, 9, 16, 16), (4, 16, 16))
        T["e.tt.x"] = v17
        T["e.tt.out"] = vae._encode(v17)
        meta["temporal_tiling_case"] = {"tile": [1, 9, 16, 16], "stride": [4, 16, 16]}
        # spatial tiling: 80 x 112 pixels, 48 x 48 tiles, stride 32
        vs = torch.randn(1, 3, 5, 80, 112)
        vae.apply_tiling((1, 9, 48, 48), (8, 32, 32))
        T["e.st.x"] = vs
        T["e.st.out"] = vae._encode(vs)
        meta["spatial_tiling_case"] = {"tile": [1, 9, 48, 48], "stride": [8, 32, 32]}
        # both
        vb = torch.randn(1, 3, 17, 80, 80)
        vae.apply_tiling((1, 9, 48, 48), (4, 32, 32))
        T["e.bt.x"] = vb
        T["e.bt.out"] = vae._encode(vb)
        meta["both_tiling_case"] = {"tile": [1, 9, 48, 48], "stride": [4, 32, 32]}
        meta["enc_tiling"] = {}
        for shape in ((1, 3, 121, 512, 768), (1, 3, 241, 512, 768), (1, 3, 241, 768, 1280), (1, 3, 1, 512, 768), (1, 3, 49, 256, 256)):
            ts, st = vae.get_enc_optimal_tiling(list(shape))
            meta["enc_tiling"]["x".join(map(str, shape))] = [list(ts), list(st)]
    save_file({k: t.contiguous() for k, t in T.items()}, os.path.join(OUT, "vae_enc_tiny.safetensors"))
    with torch.device("meta"):
        full = kvae.AutoencoderKLHunyuanVideo()
    meta["full_manifest"] = {k: list(t.shape) for k, t in full.state_dict().items()
                             if k.startswith("encoder.") or k.startswith("quant_conv")}
    with open(os.path.join(OUT, "vae_enc_meta.json"), "w") as f:
        json.dump(meta, f, indent=0)
    print("vae encoder golden written", os.path.getsize(os.path.join(OUT, "vae_enc_tiny.safetensors")) / 1e6, "MB;",
          len(meta["full_manifest"]), "encoder tensors in the full manifest;", meta["enc_tiling"])


if __name__ == "__main__":
    main()
