"""Golden vectors for the HunyuanVideo VAE *decode* path, generated FROM THE REFERENCE's own
kandinsky/models/vae.py (conv / resnet / upsample / decoder / tiling / blend code) on CPU in fp32.

TEST INFRASTRUCTURE — run once in the build container:   PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_vae.py

diffusers is not installed, so the seven names vae.py imports from it are shimmed.  Six are
plumbing (config mixins, output holders).  `Attention` (the mid-block attention, diffusers code that is NOT part of
the reference tree, un-pinned in requirements.txt:14) is restated from its definition for the configuration the
reference instantiates at vae.py:312-322: GroupNorm -> to_q/to_k/to_v Linear(+bias) -> 1 head of dim C ->
softmax(q k^T / sqrt(C) + mask) v -> to_out[0] Linear -> + residual (rescale_output_factor 1).  Parity of that one
module is therefore pinned only through the reference's call site (mask construction, layout), as DESIGN.md states.
Only data is written (tests/golden/vae_tiny.safetensors, vae_tiling.json); no reference source text is stored.
"""
import json
import os
import sys
import types

os.environ["TORCH_COMPILE_DISABLE"] = "1"
sys.dont_write_bytecode = True

import torch
import torch.nn as nn
import torch.nn.functional as F
from safetensors.torch import save_file

REF = os.environ.get("K5_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


class _Attention(nn.Module):
    def __init__(self, query_dim, heads=1, dim_head=64, eps=1e-6, norm_num_groups=32, residual_connection=True, bias=True,
                 upcast_softmax=True, _from_deprecated_attn_block=True, **kw):
        super().__init__()
        assert heads == 1 and dim_head == query_dim
        self.group_norm = nn.GroupNorm(norm_num_groups, query_dim, eps=eps, affine=True)
        self.to_q = nn.Linear(query_dim, query_dim, bias=bias)
        self.to_k = nn.Linear(query_dim, query_dim, bias=bias)
        self.to_v = nn.Linear(query_dim, query_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(query_dim, query_dim, bias=True), nn.Dropout(0.0)])
        self.residual_connection = residual_connection

    def forward(self, hidden_states, attention_mask=None):
        residual = hidden_states
        h = self.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        o = F.scaled_dot_product_attention(q.unsqueeze(1), k.unsqueeze(1), v.unsqueeze(1),
                                           attn_mask=attention_mask.unsqueeze(1)).squeeze(1)
        o = self.to_out[1](self.to_out[0](o))
        return o + residual if self.residual_connection else o


def _install_shims():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class ConfigMixin:
        @property
        def config(self):
            return self._k5_config

    def register_to_config(init):
        import inspect

        def wrapped(self, *a, **k):
            sig = inspect.signature(init)
            bound = sig.bind(self, *a, **k)
            bound.apply_defaults()
            cfg = {n: v for n, v in bound.arguments.items() if n != "self"}
            self._k5_config = types.SimpleNamespace(**cfg)
            init(self, *a, **k)
        return wrapped

    class DecoderOutput:
        def __init__(self, sample):
            self.sample = sample

    for pkg in ("diffusers", "diffusers.utils", "diffusers.models", "diffusers.models.autoencoders"):
        mod(pkg).__path__ = []
    mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    mod("diffusers.utils.accelerate_utils", apply_forward_hook=lambda f: f)
    mod("diffusers.models.activations", get_activation=lambda name: nn.SiLU())
    mod("diffusers.models.attention_processor", Attention=_Attention)
    mod("diffusers.models.modeling_outputs", AutoencoderKLOutput=DecoderOutput)
    mod("diffusers.models.modeling_utils", ModelMixin=nn.Module)
    mod("diffusers.models.autoencoders.vae", DecoderOutput=DecoderOutput, DiagonalGaussianDistribution=object)


def main():
    _install_shims()
    for name, sub in (("kandinsky", "/kandinsky"), ("kandinsky.models", "/kandinsky/models")):
        m = types.ModuleType(name)
        m.__path__ = [REF + sub]
        sys.modules[name] = m
    import kandinsky.models.vae as kvae

    torch.manual_seed(77)
    T = {}
    cfg = dict(latent_channels=16, block_out_channels=(16, 32, 32, 32), layers_per_block=2, norm_num_groups=8)
    vae = kvae.AutoencoderKLHunyuanVideo(**cfg).eval()
    with torch.no_grad():
        for n, p in vae.named_parameters():
            if n.startswith("encoder") or n.startswith("quant_conv"):
                continue
            if "norm" in n and n.endswith("weight"):
                p.normal_(1.0, 0.2)
            elif n.endswith("bias"):
                p.normal_(0, 0.1)
            else:
                p.normal_(0, 1.5 / (p[0].numel() ** 0.5))
        for k, v in vae.state_dict().items():
            if k.startswith("decoder.") or k.startswith("post_quant_conv"):
                T["w." + k] = v.clone()
        meta = {"config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()},
                "scaling_factor": vae.config.scaling_factor}
        dec = vae.decoder
        # ---- per-module vectors ----
        x = torch.randn(1, 32, 3, 6, 5)
        T["m.conv.x"] = x
        T["m.conv.out"] = dec.mid_block.resnets[0].conv1(x)
        T["m.resnet.out"] = dec.mid_block.resnets[0](x)
        x16 = torch.randn(1, 32, 3, 6, 5)
        T["m.resnet_sc.x"] = x16
        T["m.resnet_sc.out"] = dec.up_blocks[3].resnets[0](x16)   # 32 -> 16 with 1x1x1 shortcut
        T["m.up_hw.out"] = dec.up_blocks[0].upsamplers[0](x)       # (1,2,2)
        T["m.up_thw.out"] = dec.up_blocks[1].upsamplers[0](x)      # (2,2,2)
        x1 = torch.randn(1, 32, 1, 6, 5)
        T["m.up_thw1.x"] = x1
        T["m.up_thw1.out"] = dec.up_blocks[1].upsamplers[0](x1)    # single frame
        T["m.mid.out"] = dec.mid_block(x)
        T["m.mask.3x4"] = kvae.prepare_causal_attention_mask(3, 4, torch.float32, "cpu", 1)[0]
        # ---- whole decoder / decode paths ----
        z = torch.randn(1, 16, 3, 6, 5)
        T["d.z"] = z
        T["d.decoder"] = dec(vae.post_quant_conv(z))
        T["d.decode"] = vae.decode(z).sample                      # picks "no tiling" for this small shape
        assert torch.equal(T["d.decode"], T["d.decoder"])
        # temporal tiling: 7 latent frames, tiles of 2(+1) latent frames, stride 1
        z7 = torch.randn(1, 16, 7, 4, 4)
        vae.apply_tiling((1, 9, 32, 32), (4, 32, 32))              # tile_sample_min_num_frames=8, stride 4
        T["d.tt.z"] = z7
        T["d.tt.out"] = vae._decode(z7).sample
        meta["temporal_tiling_case"] = {"tile": [1, 9, 32, 32], "stride": [4, 32, 32]}
        # spatial tiling (single temporal tile): 10 x 14 latent, 6x6 tiles, stride 4
        zs = torch.randn(1, 16, 2, 10, 14)
        vae.apply_tiling((1, 9, 48, 48), (8, 32, 32))
        T["d.st.z"] = zs
        T["d.st.out"] = vae._decode(zs).sample
        meta["spatial_tiling_case"] = {"tile": [1, 9, 48, 48], "stride": [8, 32, 32]}
        # both: temporal tiles that are themselves spatially tiled
        zb = torch.randn(1, 16, 5, 10, 10)
        vae.apply_tiling((1, 9, 48, 48), (4, 32, 32))
        T["d.bt.z"] = zb
        T["d.bt.out"] = vae._decode(zb).sample
        meta["both_tiling_case"] = {"tile": [1, 9, 48, 48], "stride": [4, 32, 32]}
        # blends
        a, b = torch.randn(1, 3, 5, 6, 7), torch.randn(1, 3, 5, 6, 7)
        T["b.a"], T["b.b"] = a, b
        T["b.t"] = vae.blend_t(a, b.clone(), 3)
        T["b.v"] = vae.blend_v(a, b.clone(), 4)
        T["b.h"] = vae.blend_h(a, b.clone(), 5)
        # tiling policy for the BASELINE shapes (and a few more)
        meta["dec_tiling"] = {}
        for shape in ((1, 16, 13, 32, 32), (1, 16, 31, 64, 96), (1, 16, 61, 64, 96), (1, 16, 61, 96, 160),
                      (1, 16, 31, 96, 64), (1, 16, 31, 64, 64), (1, 16, 1, 64, 96)):
            ts, st = vae.get_dec_optimal_tiling(list(shape))
            meta["dec_tiling"]["x".join(map(str, shape))] = [list(ts), list(st)]
    save_file({k: v.contiguous() for k, v in T.items()}, os.path.join(OUT, "vae_tiny.safetensors"))
    # full-size decoder state_dict manifest (names + shapes)
    with torch.device("meta"):
        full = kvae.AutoencoderKLHunyuanVideo()
    meta["full_manifest"] = {k: list(v.shape) for k, v in full.state_dict().items()
                             if k.startswith("decoder.") or k.startswith("post_quant_conv")}
    meta["opt_temporal_tiling"] = {str(k): list(v) for k, v in kvae.OPT_TEMPORAL_TILING.items()}
    meta["opt_spatial_tiling"] = {str(k): list(v) for k, v in kvae.OPT_SPATIAL_TILING.items()}
    with open(os.path.join(OUT, "vae_meta.json"), "w") as f:
        json.dump(meta, f, indent=0)
    print("vae golden written", os.path.getsize(os.path.join(OUT, "vae_tiny.safetensors")) / 1e6, "MB;",
          len(meta["full_manifest"]), "decoder tensors in the full manifest;", meta["dec_tiling"])


if __name__ == "__main__":
    main()
