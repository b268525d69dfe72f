"""Golden vector for the HunyuanVideo VAE decoder at PRODUCTION WIDTH (128 / 256 / 512 / 512 channels, 32 groups) on the
production latent tile (5, 64, 96) — FROM THE REFERENCE's own kandinsky/models/vae.py (decoder :589-696, resnet :230-275,
upsample :166-205, causal conv :125-163), fp32 on CPU.

TEST INFRASTRUCTURE — run once in the build container (needs /root/reference; ~118.8 TFLOP of fp32 conv on the host
cores: 20-40 minutes, ~30 GB):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_vae_fullwidth.py [ref|oracle32|oraclebf16 ...]

The (5,64,96) tile is what the tiling policy cuts the 5 s 768x512 clip into (tests/golden/vae_meta.json dec_tiling), i.e.
what bench.py's decode times: every layer takes the kernel it takes in production (conv3d_w4 with fused GroupNorm
statistics, the quad-statistics GroupNorm, the C = 512 flash mid-attention, conv_out3).  Weights:
oracle.vae_oracle.synthetic_decoder_state_dict(full_manifest, seed = 21) (the consumer regenerates them); latent N(0,1)
seed 22.  Stored (data only): 32768 sampled output elements with their flat indices, sum / sum of squares of the whole
output, and the same samples from oracle/vae_oracle.py in fp32 mode (pins the oracle at production width) and bf16 mode
(the tighter target for the engine).
"""
import json
import os
import sys
import time
import types

os.environ["TORCH_COMPILE_DISABLE"] = "1"
sys.dont_write_bytecode = True

import torch
from safetensors.torch import load_file, save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = os.environ.get("K5_REFERENCE", "/root/reference")
TILE = (5, 64, 96)
WSEED, ZSEED, NSAMP = 21, 22, 32768
PATH = os.path.join(OUT, "vae_fullwidth.safetensors")


def inputs():
    import vae_oracle as V
    meta = json.load(open(os.path.join(OUT, "vae_meta.json")))
    sd = V.synthetic_decoder_state_dict(meta["full_manifest"], WSEED)
    z = torch.randn(1, 16, *TILE, generator=torch.Generator().manual_seed(ZSEED))
    numel = 3 * (4 * (TILE[0] - 1) + 1) * 8 * TILE[1] * 8 * TILE[2]
    idx = torch.randperm(numel, generator=torch.Generator().manual_seed(23))[:NSAMP].sort().values
    return V, sd, z, idx


def run_reference(sd, z):
    import gen_golden_vae as G
    G._install_shims()
    for name, sub in (("kandinsky", "/kandinsky"), ("kandinsky.models", "/kandinsky/models")):
        m = types.ModuleType(name)
        m.__path__ = [REF + sub]
        sys.modules[name] = m
    import kandinsky.models.vae as kvae
    vae = kvae.AutoencoderKLHunyuanVideo().eval()
    own = vae.state_dict()
    own.update({k: v for k, v in sd.items()})
    missing = vae.load_state_dict(own, strict=True)
    with torch.no_grad():
        return vae.decoder(vae.post_quant_conv(z)), str(missing)


def main():
    what = sys.argv[1:] or ["ref", "oracle32", "oraclebf16"]
    V, sd, z, idx = inputs()
    T = load_file(PATH) if os.path.exists(PATH) else {}
    mpath = os.path.join(OUT, "vae_fullwidth_meta.json")
    meta = json.load(open(mpath)) if os.path.exists(mpath) else {}
    meta.update({"tile": list(TILE), "weights_seed": WSEED, "latent_seed": ZSEED, "index_seed": 23})
    T["sample_idx"] = idx
    cfg = dict(latent_channels=16, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2, norm_num_groups=32)
    for w in what:
        t0 = time.time()
        with torch.no_grad():
            if w == "ref":
                out, missing = run_reference(sd, z)
                meta["load_state_dict"] = missing
            else:
                out = V.decoder_forward(sd, z, cfg, "fp32" if w == "oracle32" else "bf16")
        out = out.float()
        T[w + ".sample_val"] = out.reshape(-1)[idx].contiguous()
        meta[w] = {"out_shape": list(out.shape), "out_sum": float(out.double().sum()), "out_sumsq": float(out.double().pow(2).sum()),
                   "abs_max": float(out.abs().max()), "seconds": round(time.time() - t0, 1)}
        print(w, meta[w], flush=True)
        del out
        save_file({k: v.contiguous() for k, v in T.items()}, PATH)
        json.dump(meta, open(mpath, "w"), indent=1)


if __name__ == "__main__":
    main()
