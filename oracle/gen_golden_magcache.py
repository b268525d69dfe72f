"""Generate golden vectors for the MagCache forward FROM THE REFERENCE ITSELF.

TEST INFRASTRUCTURE — run once in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_magcache.py

Imports the reference's `kandinsky/magcache_utils.py` (its `@torch.compile` decorator replaced by the identity: there is
no inductor backend to run here and the decorator does not change the arithmetic) under the patches of
oracle/_ref_import.py, installs it on the tiny DiT of tests/golden/dit_tiny.safetensors with
`set_magcache_params`, and runs the reference's own `generate`.  Stores data only: the ratio arrays used, the
interpolated ratio tables the reference derived, the per-call skip decisions and the final latents.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import import_reference  # noqa: E402
from gen_golden import TINY, conf_ns  # noqa: E402

import torch  # noqa: E402
from safetensors.torch import load_file, save_file  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    r = import_reference()
    torch.compile = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    import kandinsky.magcache_utils as kmag

    g = load_file(os.path.join(OUT, "dit_tiny.safetensors"))
    confs = json.load(open(os.path.join(OUT, "configs_parsed.json")))
    dit = r.dit.DiffusionTransformer3D(**TINY).eval()
    dit.load_state_dict({k[2:]: v for k, v in g.items() if k.startswith("w.")})
    te = {"text_embeds": g["fwd.text"], "pooled_embed": g["fwd.pooled"]}
    ne = {"text_embeds": g["gen.null_text"], "pooled_embed": g["gen.null_pooled"]}
    pos = [torch.arange(3), torch.arange(4), torch.arange(6)]

    # count the decoder-block evaluations of every forward: 0 = the 32(2) visual blocks were skipped
    calls = []
    blk0 = dit.visual_transformer_blocks[0]
    blk0.register_forward_hook(lambda m, a, o: calls.__setitem__(-1, 1))
    orig_before = dit.before_visual_transformer_blocks

    def spy_before(*a, **k):
        calls.append(0)
        return orig_before(*a, **k)
    dit.before_visual_transformer_blocks = spy_before

    sft = confs["config_5s_sft.yaml"]["magcache"]["mag_ratios"]
    nocfg = confs["config_5s_nocfg.yaml"]["magcache"]["mag_ratios"]
    hand = [1.0, 1.0, 0.99, 1.01, 0.97, 0.96, 1.02, 0.9, 0.98, 1.0, 1.0, 0.99, 0.95, 1.05, 1.0, 1.0, 0.99, 0.98]  # len 18 -> 10 steps exact
    cases = [  # (tag, ratios, num_steps, guidance_weight, scheduler_scale)
        ("sft_50", sft, 50, 5.0, 5.0),          # len(ratios)+2 == 2*steps: no interpolation
        ("sft_12", sft, 12, 3.0, 5.0),          # interpolated tables, CFG
        ("nocfg_50", nocfg, 50, 1.0, 5.0),
        ("nocfg_9", nocfg, 9, 1.0, 10.0),       # interpolated, w = 1 (cnt advances by 2, slot 0 only)
        ("hand_10", hand, 10, 2.0, 5.0),
    ]
    T, meta = {}, {"cases": []}
    with torch.no_grad():
        for tag, ratios, steps, w, s in cases:
            no_cfg = abs(w - 1.0) <= 1e-6
            kmag.set_magcache_params(dit, list(ratios), steps, no_cfg)
            calls.clear()
            out = r.gen.generate(dit, "cpu", (3, 8, 12, 16), steps, te, ne, pos, torch.arange(7), torch.arange(4),
                                 w, s, conf_ns(dict(type="flash")), seed=6554)
            assert dit.cnt == 0, "the reference resets its counter after num_steps*2 calls"
            T[f"mag.{tag}.final"] = out
            T[f"mag.{tag}.table"] = torch.from_numpy(np.asarray(dit.mag_ratios, dtype=np.float64))
            meta["cases"].append({"tag": tag, "ratios": list(map(float, ratios)), "num_steps": steps, "guidance_weight": w,
                                  "scheduler_scale": s, "no_cfg": no_cfg, "ran_blocks": list(calls)})
            print(tag, "forwards", len(calls), "skipped", len(calls) - sum(calls))
    save_file({k: v.contiguous() for k, v in T.items()}, os.path.join(OUT, "magcache_tiny.safetensors"))
    with open(os.path.join(OUT, "magcache_meta.json"), "w") as f:
        json.dump(meta, f)


if __name__ == "__main__":
    main()
