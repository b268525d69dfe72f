"""What would folding the LayerNorm + modulation of the visual blocks into the consumer GEMMs cost in parity?  (VERDICT r5 "Next round" #5: "pinned first".)

    python oracle/gen_lnfold_yardstick.py        # ~ 10-30 min of host cores; writes profiles/r06_lnfold_yardstick.json

BASELINE config 1 IN FULL (32 visual blocks x 16 steps, the weights / noise / prompt streams of oracle/gen_golden_fulldepth.py c1) through the bf16-island
oracle with oracle.k5_oracle.LN_FOLD = True: every `apply_scale_shift_norm` of the visual stream (reference nn.py:25-28, dit.py:61-79, nn.py:374-400)
becomes y = rstd (x W'^T - mean colsum(W')) + W shift + b with W' = bf16(W (1 + scale)) inside the consumer linear layer.  Compared on the 16 384 samples
of the committed golden (tests/golden/dit_fulldepth_c1.safetensors) with the REFERENCE's generate() (fp32) and with the unfolded bf16-island oracle.
The reference is not imported: its numbers are the golden's.  Test infrastructure, like everything under oracle/."""
import json
import os
import sys
import time

import torch
from safetensors.torch import load_file

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import k5_oracle as O  # noqa: E402


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "dit_fulldepth_meta.json")))
    c = meta["c1"]
    G = load_file(os.path.join(ROOT, "tests", "golden", "dit_fulldepth_c1.safetensors"))
    cfg = dict(O.LITE_2B)
    sd = O.synthetic_state_dict(O.DitConfig(**cfg), seed=meta["weights_seed"])
    for k in sd:
        if k.endswith(("query_norm.weight", "key_norm.weight")):
            sd[k] = torch.full((64,), float(meta["qk_gain"]))
    T, H, W = c["latent"]
    g = torch.Generator().manual_seed(c["xseed"])
    te = {"text_embeds": torch.randn(c["L"], 3584, generator=g), "pooled_embed": torch.randn(1, 768, generator=g)}
    ne = {"text_embeds": torch.randn(c["Lnull"], 3584, generator=g), "pooled_embed": torch.randn(1, 768, generator=g)}
    pos = [torch.arange(T), torch.arange(H // 2), torch.arange(W // 2)]
    noise = torch.randn(T, H, W, 16, generator=torch.Generator().manual_seed(c["seed"]))
    rel = lambda a, b: float((a - b).norm() / b.norm())   # noqa: E731
    res = {"config": "c1: latent (13,32,32), 32 visual blocks x 16 steps, guidance 1", "yardstick_bf16_oracle_vs_reference": c["bf16_oracle_vs_ref_final"]}
    for fold in (True,):
        O.LN_FOLD = fold
        t0 = time.time()
        with torch.no_grad():
            fin = O.generate(sd, O.DitConfig(**cfg), noise, c["steps"], te, ne, pos, torch.arange(c["L"]), torch.arange(c["Lnull"]), c["w"], c["s"], (1.0, 2.0, 2.0), None, "bf16")
        O.LN_FOLD = False
        got = fin.reshape(-1)[G["sample_idx"]]
        res["lnfold_vs_reference_fp32"] = rel(got, G["final_ref"])
        res["lnfold_vs_unfolded_bf16_oracle"] = rel(got, G["final_bf16_oracle"])
        res["seconds"] = time.time() - t0
        print(res, flush=True)
    res["ratio_to_yardstick"] = res["lnfold_vs_reference_fp32"] / res["yardstick_bf16_oracle_vs_reference"]
    json.dump(res, open(os.path.join(ROOT, "profiles", "r06_lnfold_yardstick.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
