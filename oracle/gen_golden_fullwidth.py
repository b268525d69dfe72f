"""Golden vector at the FULL 2B-Lite width (D = 1792, 28 heads, FF = 7168, text 3584 / 768) FROM THE REFERENCE ITSELF.

TEST INFRASTRUCTURE — run once in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_fullwidth.py

SURVEY.md §8c: "one full-width single decoder block ... checksum + a few hundred sampled elements, weights regenerated
from a documented seed".  The reference DiffusionTransformer3D (fp32, patches of oracle/_ref_import.py) is built with
1 text block + 2 visual blocks at full width, loaded with oracle.synthetic_state_dict(cfg, seed = 3) (torch CPU
generator, one seed per tensor — the consumer regenerates the same weights), and run on a (5,16,16) latent with a
37-token prompt.  Stored (data only): the inputs, 4096 sampled output elements with their flat indices, and the sum /
sum of squares of the whole output.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from _ref_import import import_reference  # noqa: E402

import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    import k5_oracle as O          # before import_reference(): it aliases torch.bfloat16 for the reference's fp32 mode
    cfgd = dict(O.LITE_2B, num_visual_blocks=2, num_text_blocks=1)
    cfg = O.DitConfig(**cfgd)
    sd = O.synthetic_state_dict(cfg, seed=3)
    r = import_reference()
    dit = r.dit.DiffusionTransformer3D(**cfgd).eval()
    missing = dit.load_state_dict(sd, strict=True)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(5, 16, 16, 16, generator=g)
    text, pooled = torch.randn(37, 3584, generator=g), torch.randn(1, 768, generator=g)
    t = torch.tensor([875.0])
    pos = [torch.arange(5), torch.arange(8), torch.arange(8)]
    xin = torch.cat([x, torch.zeros(5, 16, 16, 17)], dim=-1)
    with torch.no_grad():
        out = dit(xin, text, pooled, t, pos, torch.arange(37), scale_factor=(1.0, 2.0, 2.0)).float()
    idx = torch.randperm(out.numel(), generator=torch.Generator().manual_seed(5))[:4096].sort().values
    save_file({"x": x, "text": text, "pooled": pooled, "time": t, "sample_idx": idx, "sample_val": out.reshape(-1)[idx].contiguous()},
              os.path.join(OUT, "dit_fullwidth.safetensors"))
    meta = {"config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfgd.items()}, "weights_seed": 3, "input_seed": 11,
            "out_shape": list(out.shape), "out_sum": float(out.double().sum()), "out_sumsq": float(out.double().pow(2).sum()),
            "load_state_dict": str(missing)}
    json.dump(meta, open(os.path.join(OUT, "dit_fullwidth_meta.json"), "w"), indent=1)
    print(meta["out_shape"], meta["out_sum"], meta["out_sumsq"])


if __name__ == "__main__":
    main()
