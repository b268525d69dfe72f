"""Generate golden vectors for the DiT / sampler path FROM THE REFERENCE ITSELF.

TEST INFRASTRUCTURE — run once in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

Imports the reference python under the patches of oracle/_ref_import.py (fp32 oracle mode,
SURVEY.md Appendix C) and writes *data only* (inputs, weights of a tiny config, outputs of the
reference's own functions) to tests/golden/.  Nothing from the reference's source text is stored.
"""
import json
import os
import sys
from types import SimpleNamespace as NS

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import import_reference, REF  # noqa: E402

import torch  # noqa: E402
import yaml  # noqa: E402
from safetensors.torch import save_file  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)

TINY = dict(in_visual_dim=16, in_text_dim=96, in_text_dim2=48, time_dim=64, out_visual_dim=16,
            patch_size=(1, 2, 2), model_dim=128, ff_dim=256, num_text_blocks=1, num_visual_blocks=2,
            axes_dims=(16, 24, 24), visual_cond=True)
LITE = dict(in_visual_dim=16, out_visual_dim=16, time_dim=512, patch_size=(1, 2, 2), model_dim=1792,
            ff_dim=7168, num_text_blocks=2, num_visual_blocks=32, axes_dims=(16, 24, 24),
            visual_cond=True, in_text_dim=3584, in_text_dim2=768)


def conf_ns(attn):
    return NS(model=NS(dit_params=NS(patch_size=(1, 2, 2)), attention=NS(**attn)),
              metrics=NS(scale_factor=(1.0, 2.0, 2.0)))


def main():
    r = import_reference()
    knn, kdit, kutils, kgen = r.nn, r.dit, r.utils, r.gen
    torch.manual_seed(1234)
    T = {}  # tensors to save
    meta = {"tiny_config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in TINY.items()}}

    dit = kdit.DiffusionTransformer3D(**TINY).eval()
    with torch.no_grad():
        for n, p in dit.named_parameters():
            if "modulation" in n:
                p.normal_(0, 0.05)
            elif n.endswith("norm.weight"):
                p.normal_(1.0, 0.1)
            elif n.endswith("norm.bias"):
                p.normal_(0, 0.1)
            elif n.endswith(".bias"):
                p.normal_(0, 0.05)
    for k, v in dit.state_dict().items():
        T["w." + k] = v.detach().clone()

    D, H, hd = 128, 2, 64
    with torch.no_grad():
        # ---------------- G1 per-op ----------------
        T["op.get_freqs.8"] = kutils.get_freqs(8)
        T["op.get_freqs.12"] = kutils.get_freqs(12)
        T["op.get_freqs.896"] = kutils.get_freqs(896)
        pos = [torch.arange(3), torch.arange(4), torch.arange(6)]
        rope3 = dit.visual_rope_embeddings((3, 4, 6), pos, (1.0, 2.0, 2.0))   # (3,4,6,1,32,2,2)
        T["op.rope3d"] = rope3.squeeze(-4).contiguous()
        tpos = torch.tensor([0, 1, 2, 5, 9, 10, 11])
        rope1 = dit.text_rope_embeddings(tpos)                                # (7,1,32,2,2)
        T["op.rope1d.pos"] = tpos
        T["op.rope1d"] = rope1.squeeze(-4).contiguous()
        xq = torch.randn(72, H, hd)
        T["op.rotary.x"] = xq
        T["op.rotary.out"] = knn.apply_rotary(xq, rope3.flatten(0, 2))
        t_in = torch.tensor([731.25])
        T["op.time.in"] = t_in
        T["op.time.out"] = dit.time_embeddings(t_in)
        temb = torch.randn(1, 64)
        T["op.mod.in"] = temb
        T["op.mod.out9"] = dit.visual_transformer_blocks[0].visual_modulation(temb)
        xs = torch.randn(72, D) * 2 + 0.3
        sc, sh, gt = torch.randn(1, D) * 0.3, torch.randn(1, D) * 0.3, torch.randn(1, D)
        T["op.ssn.x"], T["op.ssn.scale"], T["op.ssn.shift"] = xs, sc, sh
        T["op.ssn.out"] = knn.apply_scale_shift_norm(dit.visual_transformer_blocks[0].self_attention_norm, xs, sc, sh)
        xo = torch.randn(72, D)
        T["op.gate.out_in"], T["op.gate.gate"] = xo, gt
        T["op.gate.out"] = knn.apply_gate_sum(xs, xo, gt)
        sa = dit.visual_transformer_blocks[0].self_attention
        q, k, v = sa.get_qkv(xs)
        qn, kn = sa.norm_qk(q, k)
        T["op.qkv.q"], T["op.qkv.k"], T["op.qkv.v"] = q, k, v
        T["op.normqk.q"], T["op.normqk.k"] = qn, kn
        T["op.selfattn.out"] = sa(xs, rope3.flatten(0, 2))
        text7 = torch.randn(7, D)
        T["op.cross.cond"] = text7
        T["op.cross.out"] = dit.visual_transformer_blocks[0].cross_attention(xs, text7)
        T["op.ff.out"] = dit.visual_transformer_blocks[0].feed_forward(xs)
        T["op.encattn.out"] = dit.text_transformer_blocks[0].self_attention(text7, rope1)
        xv = torch.randn(3, 8, 12, 33)
        T["op.visemb.x"] = xv
        T["op.visemb.out"] = dit.visual_embeddings(xv)
        ve = torch.randn(3, 4, 6, D)
        T["op.outlayer.x"] = ve
        T["op.outlayer.out"] = dit.out_layer(ve, None, temb)
        traw = torch.randn(7, 96)
        T["op.textemb.x"] = traw
        T["op.textemb.out"] = dit.text_embeddings(traw)

        # ---------------- G2 blocks ----------------
        T["blk.enc.out"] = dit.text_transformer_blocks[0](text7, temb, rope1)
        T["blk.dec.out"] = dit.visual_transformer_blocks[0](xs, text7, temb, rope3.flatten(0, 2), None)

        # ---------------- G3 full forward dense ----------------
        x = torch.randn(3, 8, 12, 33)
        te, pe = torch.randn(7, 96), torch.randn(1, 48)
        tm = torch.tensor([612.5])
        T["fwd.x"], T["fwd.text"], T["fwd.pooled"], T["fwd.time"] = x, te, pe, tm
        hooks, taps = [], {}
        for i, b in enumerate(dit.visual_transformer_blocks):
            hooks.append(b.register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"fwd.vblock{i}", o.clone())))
        hooks.append(dit.text_transformer_blocks[0].register_forward_hook(
            lambda m, a, o: taps.__setitem__("fwd.tblock0", o.clone())))
        T["fwd.out"] = dit(x, te, pe, tm, pos, torch.arange(7), scale_factor=(1.0, 2.0, 2.0))
        T.update(taps)
        for h in hooks:
            h.remove()

        # ---------------- NABLA: mask + block + forward ----------------
        nshape = (6, 16, 16)  # patched -> 24 blocks of 64 tokens
        attn = dict(type="nabla", P=0.7, wT=3, wH=1, wW=1, add_sta=True, method="topcdf")
        meta["nabla_attention"] = attn
        meta["nabla_patched_shape"] = list(nshape)
        xn = torch.randn(6, 32, 32, 33)
        sparse = kgen.get_sparse_params(conf_ns(attn), {"visual": xn}, "cpu")
        T["nabla.sta"] = sparse["sta_mask"][0, 0].to(torch.uint8)
        npos = [torch.arange(6), torch.arange(16), torch.arange(16)]
        # nablaT_v2 on fixed q/k
        qn_ = torch.randn(1, H, 1536, hd)
        kn_ = torch.randn(1, H, 1536, hd) + 0.5 * qn_
        bm = kutils.nablaT_v2(qn_, kn_, sparse["sta_mask"], thr=0.7)
        T["nabla.q"], T["nabla.k"] = qn_[0].transpose(0, 1).contiguous(), kn_[0].transpose(0, 1).contiguous()
        T["nabla.mask"] = bm.to_dense()[0].to(torch.uint8)
        meta["nabla_mask_density"] = float(bm.to_dense().float().mean())
        T["nabla.fwd.x"] = xn
        captured = []
        orig_nabla = kutils.nablaT_v2

        def spy(q, k, sta, thr=0.9):
            out = orig_nabla(q, k, sta, thr=thr)
            captured.append(out.to_dense()[0].to(torch.uint8))
            return out
        knn.nablaT_v2 = spy
        T["nabla.fwd.out"] = dit(xn, te, pe, tm, npos, torch.arange(7), scale_factor=(1.0, 2.0, 2.0),
                                 sparse_params=sparse)
        knn.nablaT_v2 = orig_nabla
        for i, c in enumerate(captured):
            T[f"nabla.fwd.mask{i}"] = c
        meta["nabla_fwd_mask_density"] = [float(c.float().mean()) for c in captured]

        # ---------------- G6 fractal permutation ----------------
        idx = torch.arange(6 * 16 * 16).reshape(6, 16, 16, 1).float()
        fl, _ = kutils.fractal_flatten(idx, idx, (6, 16, 16), block_mask=True)
        T["fractal.perm.6x16x16"] = fl[:, 0].long()
        back = kutils.fractal_unflatten(fl, (6, 16, 16), block_mask=True)
        assert torch.equal(back, idx)

        # ---------------- G5 STA masks ----------------
        sta_cases = [(3, 2, 2, 3, 3, 3), (6, 2, 2, 3, 1, 1), (5, 3, 4, 3, 3, 3), (61, 4, 6, 11, 3, 3), (7, 4, 6, 11, 3, 3)]
        meta["sta_cases"] = sta_cases
        for c in sta_cases:
            m = kutils.fast_sta_nabla(*c, device="cpu")
            T["sta." + "_".join(map(str, c))] = torch.from_numpy(np.packbits(m.numpy().astype(np.uint8).reshape(-1)))

        # ---------------- G4 generate trajectories ----------------
        class Spy(torch.nn.Module):
            def __init__(self, m):
                super().__init__()
                self.m, self.visual_cond, self.calls = m, m.visual_cond, []

            def forward(self, x, *a, **kw):
                self.calls.append((x[..., :16].clone(), a[2].clone()))
                return self.m(x, *a, **kw)

        te_d = {"text_embeds": te, "pooled_embed": pe}
        nte, npe = torch.randn(4, 96), torch.randn(1, 48)
        ne_d = {"text_embeds": nte, "pooled_embed": npe}
        T["gen.null_text"], T["gen.null_pooled"] = nte, npe
        gen_cases = [(4, 5.0, 1.0), (4, 5.0, 5.0), (16, 5.0, 1.0), (3, 10.0, 3.0)]
        meta["gen_cases"] = gen_cases
        seed = 6554
        meta["gen_seed"] = seed
        T["gen.noise"] = torch.randn(3, 8, 12, 16, generator=torch.Generator().manual_seed(seed))
        for (steps, s, w) in gen_cases:
            spy_m = Spy(dit)
            out = kgen.generate(spy_m, "cpu", (3, 8, 12, 16), steps, te_d, ne_d, pos, torch.arange(7),
                                torch.arange(4), w, s, conf_ns(dict(type="flash")), seed=seed)
            tag = f"gen.{steps}_{s}_{w}"
            T[tag + ".final"] = out
            stride = 1 if abs(w - 1.0) <= 1e-6 else 2
            lat = [c[0] for c in spy_m.calls[::stride]]
            assert torch.equal(lat[0], T["gen.noise"])
            T[tag + ".latents"] = torch.stack(lat[1:] + [out])
            T[tag + ".times"] = torch.cat([c[1] for c in spy_m.calls[::stride]])
        # NABLA generate, 2 steps with CFG
        T["gen.nabla.noise"] = torch.randn(6, 32, 32, 16, generator=torch.Generator().manual_seed(seed))
        out = kgen.generate(dit, "cpu", (6, 32, 32, 16), 2, te_d, ne_d, npos, torch.arange(7), torch.arange(4),
                            2.0, 5.0, conf_ns(attn), seed=seed)
        T["gen.nabla.final"] = out

    save_file({k: v.contiguous() for k, v in T.items()}, os.path.join(OUT, "dit_tiny.safetensors"))

    # ---------------- G9 state_dict manifest of the full 2B Lite config ----------------
    with torch.device("meta"):
        full = kdit.DiffusionTransformer3D(**LITE)
    manifest = {k: list(v.shape) for k, v in full.state_dict().items()}
    meta["lite_num_params"] = int(sum(int(np.prod(s)) for s in manifest.values()))
    with open(os.path.join(OUT, "dit_lite_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0)

    # ---------------- G10 parsed configs (data) ----------------
    confs = {}
    for fn in sorted(os.listdir(os.path.join(REF, "configs"))):
        with open(os.path.join(REF, "configs", fn)) as f:
            confs[fn] = yaml.safe_load(f)
    with open(os.path.join(OUT, "configs_parsed.json"), "w") as f:
        json.dump(confs, f, indent=0)

    with open(os.path.join(OUT, "dit_tiny_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    tot = sum(os.path.getsize(os.path.join(OUT, p)) for p in os.listdir(OUT))
    print("golden written:", OUT, f"{tot/1e6:.2f} MB", "nabla densities", meta["nabla_mask_density"],
          meta["nabla_fwd_mask_density"])


if __name__ == "__main__":
    main()
