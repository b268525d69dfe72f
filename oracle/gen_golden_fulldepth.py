"""Golden vectors at FULL DEPTH and FULL SCHEDULE — the whole 2B-Lite network (32 visual blocks, 2 text blocks, D = 1792, 28 heads,
FF = 7168) — FROM THE REFERENCE ITSELF (fp32 under the patches of oracle/_ref_import.py).

TEST INFRASTRUCTURE — run once in the build container (needs /root/reference; ~15 min of fp32 on 8 host cores, ~9 GB of RAM):

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_fulldepth.py          # all three parts
    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_fulldepth.py c1|f32|t50
    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_fulldepth.py c2        # ~1 h: two 32-block forwards at 47 616 tokens
    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_fulldepth.py n1        # ~45 min: NABLA at full depth, guidance 1 and 5

VERDICT r3 "missing #1 / weak #1": every full-width parity test until round 4 ran 1-2 of the 32 visual blocks, and the only 32-block
execution (bench.py) was checked for finiteness.  north_star's correctness clause is about the FINAL LATENT of the sampler.

 c1   BASELINE config 1 IN FULL through the reference's own generate() (generation_utils.py:80-129) and DiffusionTransformer3D.forward
      (dit.py:155-181, all 32 visual_transformer_blocks :175-177): latent (13, 32, 32, 16) = 3328 tokens, NFE 16, guidance 1
      (configs/config_5s_distil.yaml: num_steps 16, guidance_weight 1.0), scheduler_scale 5, text 256 x 3584 + pooled 768,
      weights oracle.synthetic_state_dict(LITE_2B, seed = 7) with every QK-norm gain 1.5.  Stored (data only): 16384 sampled values of
      the final latent + whole-tensor sums; 2048 samples of the latent entering every step (error growth along the schedule); the same
      from the bf16-island oracle (what the engine is compared with at 1e-2 in the 2-block tests: here its distance from fp32 AT DEPTH is
      the measured bf16 noise floor); per (step, block) RMS of the residual stream (forward hooks on the reference's blocks); per
      (block, head) softmax-bound statistics of the first and the last step — max|q| max|k| log2(e)/8 (the engine's Cauchy-Schwarz bound
      in the exp2 domain), the largest score, the smallest row maximum — from a recording wrapper around the patched FA.
 f32  one 32-block forward on a (5, 16, 16) latent (320 tokens, ragged against every tile size), 37 text tokens: the whole output
      (20480 values) + per-block residual RMS.  Same weights as c1.
 c2   BASELINE config 2 (what bench.py times) at full depth and size: the first 2 Euler steps of the reference's generate(NFE 50) on the
      (31, 64, 96) latent = 47 616 tokens with bench.py's own seeded weights and inputs; 16384 samples of the latent after each step.
      bench.py replays exactly this through k5_sample before its timed region and reports the distance (`parity_check`).
 n1   NABLA AT FULL DEPTH AND OVER THE SCHEDULE (VERDICT r4 "missing #2"): config 1's latent (13, 32, 32) = 3328 tokens = 52 blocks of 64
      (Hp = Wp = 16: two 8 x 8 tiles per side), attention.type nabla, P 0.9, window (11, 3, 3) (configs/config_10s_sft.yaml), NFE 16, guidance
      1 and 5, through the reference's generate() (generation_utils.py:80-129 -> get_sparse_params :10-36 -> nn.py:257-298 in all 32 blocks of
      every forward, dit.py:175-178).  Same weights as c1.  Stored: final latent samples + sums, the latent after every step, and per (step,
      CFG branch, block) the reference's kept-block bitmap (its own nablaT_v2 BlockMask, utils.py:136-163) of 4 heads, bit-packed; the same
      from the bf16-island oracle (its map flips are the bf16 noise floor of a discrete decision).
 t50  the reference's generate() for 50 steps on the tiny model (tests/golden/dit_tiny.safetensors), guidance 1 and 5, without MagCache
      (the MagCache 50-step cases are in magcache_tiny.safetensors: sft_50 / nocfg_50).

The consumer (tests/test_gpu_fulldepth.py) regenerates weights and inputs from the seeds recorded in dit_fulldepth_meta.json.
"""
import json
import math
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from _ref_import import import_reference  # noqa: E402

import torch  # noqa: E402
from safetensors.torch import load_file, save_file  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
REAL_BF16 = torch.bfloat16      # import_reference() aliases torch.bfloat16 := float32 (the reference's fp32 mode); the bf16-island oracle needs the real one


class real_bf16:
    def __enter__(self):
        self.saved, torch.bfloat16 = torch.bfloat16, REAL_BF16
    def __exit__(self, *a):
        torch.bfloat16 = self.saved


WSEED, GAIN = 7, 1.5
C1 = dict(latent=(13, 32, 32), L=256, Lnull=32, steps=16, w=1.0, s=5.0, seed=6554, xseed=31)
F32 = dict(latent=(5, 16, 16), L=37, time=875.0, xseed=11)
META = os.path.join(OUT, "dit_fulldepth_meta.json")


def weights(O):
    cfgd = dict(O.LITE_2B)
    assert cfgd["num_visual_blocks"] == 32 and cfgd["num_text_blocks"] == 2
    sd = O.synthetic_state_dict(O.DitConfig(**cfgd), seed=WSEED)
    for k in sd:
        if k.endswith(("query_norm.weight", "key_norm.weight")):
            sd[k] = torch.full((64,), GAIN)
    return cfgd, sd


def c1_inputs():
    g = torch.Generator().manual_seed(C1["xseed"])
    te = {"text_embeds": torch.randn(C1["L"], 3584, generator=g), "pooled_embed": torch.randn(1, 768, generator=g)}
    ne = {"text_embeds": torch.randn(C1["Lnull"], 3584, generator=g), "pooled_embed": torch.randn(1, 768, generator=g)}
    T, H, W = C1["latent"]
    return te, ne, [torch.arange(T), torch.arange(H // 2), torch.arange(W // 2)]


def load_meta():
    return json.load(open(META)) if os.path.exists(META) else {}


def save_meta(m):
    json.dump(m, open(META, "w"), indent=1)


def part_c1(O, r, dit, cfgd, sd):
    from types import SimpleNamespace as NS
    T, H, W = C1["latent"]
    N = T * (H // 2) * (W // 2)
    te, ne, pos = c1_inputs()
    conf = NS(model=NS(dit_params=NS(patch_size=(1, 2, 2)), attention=NS(type="flash")), metrics=NS(scale_factor=(1.0, 2.0, 2.0)))
    # ---- instrumentation (reads only; the arithmetic is the reference's) ----
    step_in, resid_rms, bound_stats = [], [], {}
    cur = {"step": -1, "blk": 0}
    for i, blk in enumerate(dit.visual_transformer_blocks):
        blk.register_forward_hook(lambda m, a, o, i=i: resid_rms[-1].__setitem__(i, float(o.float().pow(2).mean().sqrt())))
    fa0 = r.nn.FA

    def fa_rec(q, k, v):
        if q.shape[1] == N and k.shape[1] == N:          # visual self-attention (text: L x L, cross: N x L)
            if cur["step"] in (0, C1["steps"] - 1):
                c = 0.125 * math.log2(math.e)
                qh, kh = q[0].transpose(0, 1).float(), k[0].transpose(0, 1).float()      # [H][N][64]
                qn, kn = qh.norm(dim=-1).amax(dim=-1), kh.norm(dim=-1).amax(dim=-1)
                rmax_hi, rmax_lo, rmin_lo = [], [], []
                for h in range(qh.shape[0]):
                    s_ = (qh[h] @ kh[h].t()) * c
                    rm = s_.amax(dim=-1)
                    rmax_hi.append(float(rm.max())); rmax_lo.append(float(rm.min())); rmin_lo.append(float(s_.min()))
                bound_stats.setdefault(cur["step"], []).append(
                    {"bound": [float(x) for x in (qn * kn * c)], "score_max": rmax_hi, "row_max_min": rmax_lo, "score_min": rmin_lo})
            cur["blk"] += 1
        return fa0(q, k, v)
    r.nn.FA = fa_rec

    class Spy(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m, self.visual_cond = m, m.visual_cond
        def forward(self, x, *a, **k):
            step_in.append(x[..., :16].clone())
            resid_rms.append([0.0] * len(dit.visual_transformer_blocks))
            cur["step"] += 1; cur["blk"] = 0
            return self.m(x, *a, **k)
    t0 = time.time()
    with torch.no_grad():
        final = r.gen.generate(Spy(dit), "cpu", (T, H, W, 16), C1["steps"], te, ne, pos, torch.arange(C1["L"]), torch.arange(C1["Lnull"]),
                               C1["w"], C1["s"], conf, seed=C1["seed"]).float()
    r.nn.FA = fa0
    t_ref = time.time() - t0
    noise = torch.randn(T, H, W, 16, generator=torch.Generator().manual_seed(C1["seed"]))
    assert len(step_in) == C1["steps"] and torch.equal(step_in[0], noise)
    print(f"c1: reference generate() {t_ref:.0f} s; final latent moved {float((final - noise).norm() / noise.norm()):.3f} from the noise", flush=True)
    # ---- the bf16-island oracle on the same weights (the engine's parity target; its distance from fp32 at depth = the bf16 floor) ----
    t0 = time.time()
    with real_bf16():
        fin16, traj16 = O.generate(sd, O.DitConfig(**cfgd), noise, C1["steps"], te, ne, pos, torch.arange(C1["L"]), torch.arange(C1["Lnull"]),
                                   C1["w"], C1["s"], (1.0, 2.0, 2.0), None, "bf16", return_trajectory=True)
    t_16 = time.time() - t0
    rel = lambda a, b: float((a - b).norm() / b.norm())   # noqa: E731
    print(f"c1: bf16-island oracle {t_16:.0f} s; vs reference fp32 {rel(fin16, final):.3e} on the final latent", flush=True)
    idx = torch.randperm(final.numel(), generator=torch.Generator().manual_seed(8))[:16384].sort().values
    idx_s = torch.randperm(final.numel(), generator=torch.Generator().manual_seed(9))[:2048].sort().values
    steps_ref = torch.stack([x.reshape(-1)[idx_s] for x in step_in[1:]] + [final.reshape(-1)[idx_s]])      # latent AFTER step i, i = 0..15
    steps_16 = torch.stack([x.reshape(-1)[idx_s] for x in traj16])
    save_file({"sample_idx": idx, "final_ref": final.reshape(-1)[idx].contiguous(), "final_bf16_oracle": fin16.reshape(-1)[idx].contiguous(),
               "step_idx": idx_s, "steps_ref": steps_ref.contiguous(), "steps_bf16_oracle": steps_16.contiguous()},
              os.path.join(OUT, "dit_fulldepth_c1.safetensors"))
    m = load_meta()
    m["weights_seed"], m["qk_gain"] = WSEED, GAIN
    m["c1"] = dict(C1, latent=list(C1["latent"]), tokens=N, final_sum=float(final.double().sum()), final_sumsq=float(final.double().pow(2).sum()),
                   moved_from_noise=rel(final, noise),
                   bf16_oracle_vs_ref_final=rel(fin16, final),
                   bf16_oracle_vs_ref_per_step=[rel(steps_16[i], steps_ref[i]) for i in range(C1["steps"])],
                   update_bf16_oracle_vs_ref=rel(fin16 - noise, final - noise),
                   resid_rms_per_step_block=resid_rms, softmax_bound_stats={str(k): v for k, v in bound_stats.items()},
                   seconds_reference=round(t_ref, 1), seconds_bf16_oracle=round(t_16, 1))
    save_meta(m)


def part_f32(O, r, dit, cfgd, sd):
    T, H, W = F32["latent"]
    g = torch.Generator().manual_seed(F32["xseed"])
    x = torch.randn(T, H, W, 16, generator=g)
    text, pooled = torch.randn(F32["L"], 3584, generator=g), torch.randn(1, 768, generator=g)
    pos = [torch.arange(T), torch.arange(H // 2), torch.arange(W // 2)]
    rms = [0.0] * 32
    hooks = [blk.register_forward_hook(lambda m, a, o, i=i: rms.__setitem__(i, float(o.float().pow(2).mean().sqrt())))
             for i, blk in enumerate(dit.visual_transformer_blocks)]
    xin = torch.cat([x, torch.zeros(T, H, W, 17)], dim=-1)
    with torch.no_grad():
        out = dit(xin, text, pooled, torch.tensor([F32["time"]]), pos, torch.arange(F32["L"]), scale_factor=(1.0, 2.0, 2.0)).float()
    for h in hooks:
        h.remove()
    taps = {}
    with real_bf16():
        o16 = O.dit_forward(sd, O.DitConfig(**cfgd), xin, text, pooled, torch.tensor([F32["time"]]), pos, torch.arange(F32["L"]), (1.0, 2.0, 2.0), None, "bf16", taps=taps)
    o32 = O.dit_forward(sd, O.DitConfig(**cfgd), xin, text, pooled, torch.tensor([F32["time"]]), pos, torch.arange(F32["L"]), (1.0, 2.0, 2.0), None, "fp32")
    rel = lambda a, b: float((a - b).norm() / b.norm())   # noqa: E731
    print(f"f32: fp32 oracle vs reference {rel(o32, out):.2e}; bf16-island oracle vs reference {rel(o16, out):.3e}", flush=True)
    assert rel(o32, out) < 1e-4
    save_file({"out_ref": out.contiguous(), "out_bf16_oracle": o16.float().contiguous()}, os.path.join(OUT, "dit_fulldepth_f32.safetensors"))
    m = load_meta()
    m["weights_seed"], m["qk_gain"] = WSEED, GAIN
    m["f32"] = dict(F32, latent=list(F32["latent"]), resid_rms_per_block=rms, fp32_oracle_vs_ref=rel(o32, out), bf16_oracle_vs_ref=rel(o16, out),
                    resid_rms_bf16_oracle=[float(v.float().pow(2).mean().sqrt()) for v in taps["visual_blocks"]])
    save_meta(m)


C2 = dict(latent=(31, 64, 96), L=256, Lnull=32, nfe=50, steps_kept=2, w=1.0, s=5.0, seed=6554, xseed=6555, wseed=0, gain=1.0)


def c2_inputs():
    """what bench.py feeds the engine: bf16-representable text embeddings from a CPU generator (the reference sees the same values as fp32)"""
    g = torch.Generator().manual_seed(C2["xseed"])
    q = lambda t: t.to(REAL_BF16).float()   # noqa: E731
    te = {"text_embeds": q(torch.randn(C2["L"], 3584, generator=g)), "pooled_embed": q(torch.randn(1, 768, generator=g))}
    ne = {"text_embeds": q(torch.randn(C2["Lnull"], 3584, generator=g)), "pooled_embed": q(torch.randn(1, 768, generator=g))}
    T, H, W = C2["latent"]
    return te, ne, [torch.arange(T), torch.arange(H // 2), torch.arange(W // 2)]


def part_c2(O, r):
    """BASELINE config 2 — what bench.py TIMES — in full depth at its own size: the first `steps_kept` Euler steps of the reference's
    generate() with NFE 50 on the (31, 64, 96) latent = 47 616 tokens, 32 visual blocks, weights synthetic_state_dict(seed 0) = bench.py's
    (DiffusionTransformer3D.init_synthetic(host_rng=True) draws the same per-tensor CPU streams).  ~25 min of fp32 per forward on 8 cores."""
    from types import SimpleNamespace as NS
    torch.set_num_threads(int(os.environ.get("K5_GOLDEN_THREADS", "7")))
    cfgd = dict(O.LITE_2B)
    sd = O.synthetic_state_dict(O.DitConfig(**cfgd), seed=C2["wseed"])
    dit = r.dit.DiffusionTransformer3D(**cfgd).eval()
    dit.load_state_dict(sd, strict=True, assign=True)
    T, H, W = C2["latent"]
    te, ne, pos = c2_inputs()
    conf = NS(model=NS(dit_params=NS(patch_size=(1, 2, 2)), attention=NS(type="flash")), metrics=NS(scale_factor=(1.0, 2.0, 2.0)))
    seen = []

    class Spy(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m, self.visual_cond = m, m.visual_cond
        def forward(self, x, *a, **k):
            seen.append(x[..., :16].clone())
            print(f"c2: forward {len(seen)} starts at {time.time() - t0:.0f} s", flush=True)
            if len(seen) == C2["steps_kept"] + 1:
                raise StopIteration
            return self.m(x, *a, **k)
    t0 = time.time()
    with torch.no_grad():
        try:
            r.gen.generate(Spy(dit), "cpu", (T, H, W, 16), C2["nfe"], te, ne, pos, torch.arange(C2["L"]), torch.arange(C2["Lnull"]),
                           C2["w"], C2["s"], conf, seed=C2["seed"])
        except StopIteration:
            pass
    noise = torch.randn(T, H, W, 16, generator=torch.Generator().manual_seed(C2["seed"]))
    assert torch.equal(seen[0], noise)
    idx = torch.randperm(noise.numel(), generator=torch.Generator().manual_seed(10))[:16384].sort().values
    out = {"sample_idx": idx}
    m = load_meta()
    m["c2"] = dict(C2, latent=list(C2["latent"]), tokens=T * (H // 2) * (W // 2), seconds=round(time.time() - t0, 1), after_step=[])
    for i in range(1, C2["steps_kept"] + 1):
        lat = seen[i].float()
        out[f"latent_after_{i}"] = lat.reshape(-1)[idx].contiguous()
        m["c2"]["after_step"].append({"latent_sumsq": float(lat.double().pow(2).sum()), "update_sumsq": float((lat - noise).double().pow(2).sum())})
    save_file(out, os.path.join(OUT, "dit_fulldepth_c2.safetensors"))
    save_meta(m)
    print("c2 done", m["c2"]["after_step"], flush=True)


N1 = dict(latent=(13, 32, 32), L=256, Lnull=32, steps=16, s=5.0, seed=6554, xseed=31, P=0.9, win=(11, 3, 3), heads=(0, 9, 18, 27), guidance=(1.0, 5.0))


def part_n1(O, r, dit, cfgd, sd):
    """NABLA at full depth over the whole schedule, guidance 1 and 5, from the reference's generate(); see the module docstring."""
    import numpy as np
    from types import SimpleNamespace as NS
    T, H, W = N1["latent"]
    N, nb = T * (H // 2) * (W // 2), T * (H // 2) * (W // 2) // 64
    te, ne, pos = c1_inputs()                      # same prompt tensors as c1 (xseed 31)
    wT, wH, wW = N1["win"]
    conf = NS(model=NS(dit_params=NS(patch_size=(1, 2, 2)), attention=NS(type="nabla", P=N1["P"], wT=wT, wH=wH, wW=wW, add_sta=True, method="topcdf")),
              metrics=NS(scale_factor=(1.0, 2.0, 2.0)))
    heads = list(N1["heads"])
    noise = torch.randn(T, H, W, 16, generator=torch.Generator().manual_seed(N1["seed"]))
    idx = torch.randperm(noise.numel(), generator=torch.Generator().manual_seed(8))[:16384].sort().values
    idx_s = torch.randperm(noise.numel(), generator=torch.Generator().manual_seed(9))[:2048].sort().values
    rel = lambda a, b: float((a - b).norm() / b.norm())   # noqa: E731
    out, m = {"sample_idx": idx, "step_idx": idx_s}, load_meta()
    m["n1"] = dict(N1, latent=list(N1["latent"]), win=list(N1["win"]), heads=heads, guidance=list(N1["guidance"]), tokens=N, blocks64=nb, runs={})
    flex0 = r.nn.flex_attention
    for w in N1["guidance"]:
        tag = f"w{w:g}"
        fwd_per_step = 1 if abs(w - 1.0) < 1e-6 else 2
        maps, step_in, dens = [], [], []

        def flex_rec(q, k, v, block_mask=None):
            d = block_mask.to_dense()[0].bool()                    # [H][nb][nb]: the reference's own map
            maps.append(np.packbits(d[heads].numpy().reshape(-1)))
            dens.append(float(d.float().mean()))
            return flex0(q, k, v, block_mask=block_mask)
        r.nn.flex_attention = flex_rec

        class Spy(torch.nn.Module):
            def __init__(self, mod):
                super().__init__()
                self.m, self.visual_cond, self.calls = mod, mod.visual_cond, 0
            def forward(self, x, *a, **k):
                if self.calls % fwd_per_step == 0:
                    step_in.append(x[..., :16].clone())
                self.calls += 1
                return self.m(x, *a, **k)
        t0 = time.time()
        with torch.no_grad():
            final = r.gen.generate(Spy(dit), "cpu", (T, H, W, 16), N1["steps"], te, ne, pos, torch.arange(N1["L"]), torch.arange(N1["Lnull"]),
                                   w, N1["s"], conf, seed=N1["seed"]).float()
        r.nn.flex_attention = flex0
        t_ref = time.time() - t0
        assert torch.equal(step_in[0], noise) and len(maps) == N1["steps"] * fwd_per_step * 32
        print(f"n1 {tag}: reference generate() {t_ref:.0f} s; moved {rel(final, noise):.3f} from the noise; kept density {sum(dens) / len(dens):.3f}", flush=True)
        # ---- the bf16-island oracle, with ITS maps (a discrete decision on bf16 logits: how many entries flip is the noise floor) ----
        omaps = []
        nbm0 = O.nabla_block_mask

        def nbm_rec(q, k, sta, thr, mode):
            bm = nbm0(q, k, sta, thr, mode)
            omaps.append(np.packbits(bm[heads].numpy().reshape(-1)))
            return bm
        O.nabla_block_mask = nbm_rec
        t0 = time.time()
        with real_bf16():
            fin16, traj16 = O.generate(sd, O.DitConfig(**cfgd), noise, N1["steps"], te, ne, pos, torch.arange(N1["L"]), torch.arange(N1["Lnull"]),
                                       w, N1["s"], (1.0, 2.0, 2.0), {"type": "nabla", "P": N1["P"], "wT": wT, "wH": wH, "wW": wW}, "bf16", return_trajectory=True)
        O.nabla_block_mask = nbm0
        t_16 = time.time() - t0
        assert len(omaps) == len(maps)
        R, Q = np.stack(maps), np.stack(omaps)                      # [steps * branches * 32][bits]
        nbits = len(heads) * nb * nb
        flips = np.unpackbits(R ^ Q, axis=1)[:, :nbits].sum(axis=1).reshape(N1["steps"], fwd_per_step, 32)
        print(f"n1 {tag}: bf16-island oracle {t_16:.0f} s; vs reference fp32 {rel(fin16, final):.3e} on the final latent; map entries that differ "
              f"(of {nbits} per map): block 0 {flips[:, :, 0].mean():.1f}, block 31 {flips[:, :, 31].mean():.1f}, worst {int(flips.max())}", flush=True)
        steps_ref = torch.stack([x.reshape(-1)[idx_s] for x in step_in[1:]] + [final.reshape(-1)[idx_s]])
        out.update({f"{tag}.final_ref": final.reshape(-1)[idx].contiguous(), f"{tag}.final_bf16_oracle": fin16.reshape(-1)[idx].contiguous(),
                    f"{tag}.steps_ref": steps_ref.contiguous(), f"{tag}.steps_bf16_oracle": torch.stack([x.reshape(-1)[idx_s] for x in traj16]).contiguous(),
                    f"{tag}.maps_ref": torch.from_numpy(R.reshape(N1["steps"], fwd_per_step, 32, -1).copy()),
                    f"{tag}.maps_bf16_oracle": torch.from_numpy(Q.reshape(N1["steps"], fwd_per_step, 32, -1).copy())})
        m["n1"]["runs"][tag] = dict(w=w, forwards_per_step=fwd_per_step, final_sum=float(final.double().sum()), final_sumsq=float(final.double().pow(2).sum()),
                                    moved_from_noise=rel(final, noise), bf16_oracle_vs_ref_final=rel(fin16, final),
                                    update_bf16_oracle_vs_ref=rel(fin16 - noise, final - noise),
                                    bf16_oracle_vs_ref_per_step=[rel(out[f"{tag}.steps_bf16_oracle"][i], steps_ref[i]) for i in range(N1["steps"])],
                                    kept_density_ref=sum(dens) / len(dens), kept_density_per_block_ref=[float(np.mean(dens[b::32])) for b in range(32)],
                                    map_bits=nbits, map_flips_bf16_oracle_vs_ref_per_block=[float(flips[:, :, b].mean()) for b in range(32)],
                                    map_flips_bf16_oracle_vs_ref_per_step=[float(flips[i].mean()) for i in range(N1["steps"])],
                                    seconds_reference=round(t_ref, 1), seconds_bf16_oracle=round(t_16, 1))
        save_file(out, os.path.join(OUT, "dit_fulldepth_n1.safetensors"))
        save_meta(m)


def part_t50(r):
    from gen_golden import TINY, conf_ns
    g = load_file(os.path.join(OUT, "dit_tiny.safetensors"))
    dit = r.dit.DiffusionTransformer3D(**TINY).eval()
    dit.load_state_dict({k[2:]: v for k, v in g.items() if k.startswith("w.")})
    te = {"text_embeds": g["fwd.text"], "pooled_embed": g["fwd.pooled"]}
    ne = {"text_embeds": g["gen.null_text"], "pooled_embed": g["gen.null_pooled"]}
    pos = [torch.arange(3), torch.arange(4), torch.arange(6)]
    T = {}
    with torch.no_grad():
        for w in (1.0, 5.0):
            T[f"gen50.{w}"] = r.gen.generate(dit, "cpu", (3, 8, 12, 16), 50, te, ne, pos, torch.arange(7), torch.arange(4), w, 5.0,
                                             conf_ns(dict(type="flash")), seed=6554).float().contiguous()
    save_file(T, os.path.join(OUT, "dit_tiny_50steps.safetensors"))
    print("t50:", {k: float(v.norm()) for k, v in T.items()})


def main():
    parts = sys.argv[1:] or ["t50", "f32", "c1"]
    import k5_oracle as O          # before import_reference(): it aliases torch.bfloat16 for the reference's fp32 mode
    cfgd = sd = None
    if "f32" in parts or "c1" in parts or "n1" in parts:
        t0 = time.time()
        cfgd, sd = weights(O)
        print(f"weights: {sum(v.numel() for v in sd.values()) / 1e9:.3f} B parameters in {time.time() - t0:.0f} s", flush=True)
    r = import_reference()
    if "t50" in parts:
        part_t50(r)
    if "c2" in parts:
        part_c2(O, r)
    if sd is not None:
        dit = r.dit.DiffusionTransformer3D(**cfgd).eval()
        dit.load_state_dict(sd, strict=True, assign=True)
        if "f32" in parts:
            part_f32(O, r, dit, cfgd, sd)
        if "c1" in parts:
            part_c1(O, r, dit, cfgd, sd)
        if "n1" in parts:
            torch.set_num_threads(int(os.environ.get("K5_GOLDEN_THREADS", "8")))
            part_n1(O, r, dit, cfgd, sd)


if __name__ == "__main__":
    main()
