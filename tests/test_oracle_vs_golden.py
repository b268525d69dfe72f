"""Pins oracle/k5_oracle.py (own CPU restatement) against vectors produced by the reference's own
code (oracle/gen_golden.py, fp32 oracle mode).  CPU only."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import k5_oracle as O

TOL = dict(atol=2e-5, rtol=2e-5)


@pytest.fixture(scope="module")
def cfg(golden_meta):
    c = dict(golden_meta["tiny_config"])
    c["patch_size"] = tuple(c["patch_size"])
    c["axes_dims"] = tuple(c["axes_dims"])
    return O.DitConfig(**c)


def close(a, b, **kw):
    tol = dict(TOL)
    tol.update(kw)
    torch.testing.assert_close(a.float(), b.float(), **tol)


def test_get_freqs(golden):
    for n in (8, 12, 896):
        close(O.get_freqs(n), golden[f"op.get_freqs.{n}"], atol=0, rtol=1e-6)


def _rope_mats(args):
    c, s = torch.cos(args), torch.sin(args)
    return torch.stack([c, -s, s, c], dim=-1).reshape(*args.shape, 2, 2)


def test_rope_tables(golden, cfg):
    pos = [torch.arange(3), torch.arange(4), torch.arange(6)]
    a = O.rope_3d_args((3, 4, 6), pos, cfg.axes_dims, (1.0, 2.0, 2.0))
    close(_rope_mats(a), golden["op.rope3d"], atol=1e-6)
    a1 = O.rope_1d_args(golden["op.rope1d.pos"], cfg.head_dim)
    close(_rope_mats(a1), golden["op.rope1d"], atol=1e-6)


def test_apply_rotary(golden, cfg):
    pos = [torch.arange(3), torch.arange(4), torch.arange(6)]
    a = O.rope_3d_args((3, 4, 6), pos, cfg.axes_dims, (1.0, 2.0, 2.0)).reshape(72, -1)
    out = O.apply_rotary(golden["op.rotary.x"], torch.cos(a), torch.sin(a), "fp32")
    close(out, golden["op.rotary.out"])


def test_time_embeddings(golden, tiny_sd, cfg):
    close(O.time_embeddings(tiny_sd, golden["op.time.in"], cfg), golden["op.time.out"], atol=1e-4, rtol=1e-4)


def test_modulation_chunk_order(golden, tiny_sd):
    out = O.modulation(tiny_sd, "visual_transformer_blocks.0.visual_modulation", golden["op.mod.in"])
    close(out, golden["op.mod.out9"])


def test_scale_shift_norm_and_gate(golden):
    out = O.scale_shift_norm(golden["op.ssn.x"], golden["op.ssn.scale"], golden["op.ssn.shift"], "fp32")
    close(out, golden["op.ssn.out"])
    g = O.gate_sum(golden["op.ssn.x"], golden["op.gate.out_in"], golden["op.gate.gate"], "fp32")
    close(g, golden["op.gate.out"])


def test_qkv_norm(golden, tiny_sd, cfg):
    p = "visual_transformer_blocks.0.self_attention"
    x = golden["op.ssn.x"]
    q = O._linear(x, tiny_sd[p + ".to_query.weight"], tiny_sd[p + ".to_query.bias"], "fp32").reshape(72, 2, 64)
    close(q, golden["op.qkv.q"])
    close(O.rms_norm_heads(q, tiny_sd[p + ".query_norm.weight"], "fp32"), golden["op.normqk.q"])
    k = O._linear(x, tiny_sd[p + ".to_key.weight"], tiny_sd[p + ".to_key.bias"], "fp32").reshape(72, 2, 64)
    close(O.rms_norm_heads(k, tiny_sd[p + ".key_norm.weight"], "fp32"), golden["op.normqk.k"])


def test_attention_modules(golden, tiny_sd, cfg):
    pos = [torch.arange(3), torch.arange(4), torch.arange(6)]
    a = O.rope_3d_args((3, 4, 6), pos, cfg.axes_dims, (1.0, 2.0, 2.0)).reshape(72, -1)
    x = golden["op.ssn.x"]
    out = O.self_attention(tiny_sd, "visual_transformer_blocks.0.self_attention", x, torch.cos(a), torch.sin(a), cfg, "fp32")
    close(out, golden["op.selfattn.out"])
    out = O.cross_attention(tiny_sd, "visual_transformer_blocks.0.cross_attention", x, golden["op.cross.cond"], cfg, "fp32")
    close(out, golden["op.cross.out"])
    a1 = O.rope_1d_args(golden["op.rope1d.pos"], cfg.head_dim)
    out = O.self_attention(tiny_sd, "text_transformer_blocks.0.self_attention", golden["op.cross.cond"],
                           torch.cos(a1), torch.sin(a1), cfg, "fp32")
    close(out, golden["op.encattn.out"])


def test_feed_forward(golden, tiny_sd):
    close(O.feed_forward(tiny_sd, "visual_transformer_blocks.0.feed_forward", golden["op.ssn.x"], "fp32"),
          golden["op.ff.out"])


def test_visual_embeddings_patch_order(golden, tiny_sd, cfg):
    close(O.visual_embeddings(tiny_sd, golden["op.visemb.x"], cfg, "fp32"), golden["op.visemb.out"])


def test_text_embeddings(golden, tiny_sd):
    close(O.text_embeddings(tiny_sd, "text_embeddings", golden["op.textemb.x"], "fp32"), golden["op.textemb.out"])


def test_out_layer_unpatch_order(golden, tiny_sd, cfg):
    x = golden["op.outlayer.x"]
    o = O.out_layer(tiny_sd, x.reshape(-1, x.shape[-1]), golden["op.mod.in"], cfg, "fp32")
    close(O.unpatchify(o.reshape(3, 4, 6, -1), cfg.patch_size), golden["op.outlayer.out"])


def test_blocks(golden, tiny_sd, cfg):
    a1 = O.rope_1d_args(golden["op.rope1d.pos"], cfg.head_dim)
    out = O.encoder_block(tiny_sd, "text_transformer_blocks.0", golden["op.cross.cond"], golden["op.mod.in"],
                          torch.cos(a1), torch.sin(a1), cfg, "fp32")
    close(out, golden["blk.enc.out"])
    pos = [torch.arange(3), torch.arange(4), torch.arange(6)]
    a = O.rope_3d_args((3, 4, 6), pos, cfg.axes_dims, (1.0, 2.0, 2.0)).reshape(72, -1)
    out = O.decoder_block(tiny_sd, "visual_transformer_blocks.0", golden["op.ssn.x"], golden["op.cross.cond"],
                          golden["op.mod.in"], torch.cos(a), torch.sin(a), cfg, "fp32")
    close(out, golden["blk.dec.out"])


def test_full_forward_dense(golden, tiny_sd, cfg):
    pos = [torch.arange(3), torch.arange(4), torch.arange(6)]
    taps = {}
    out = O.dit_forward(tiny_sd, cfg, golden["fwd.x"], golden["fwd.text"], golden["fwd.pooled"], golden["fwd.time"],
                        pos, torch.arange(7), (1.0, 2.0, 2.0), None, "fp32", taps)
    close(taps["text_out"], golden["fwd.tblock0"], atol=1e-4, rtol=1e-4)
    for i in range(2):
        close(taps["visual_blocks"][i], golden[f"fwd.vblock{i}"], atol=1e-4, rtol=1e-4)
    close(out, golden["fwd.out"], atol=1e-4, rtol=1e-4)


def test_sta_masks(golden, golden_meta):
    for c in golden_meta["sta_cases"]:
        m = O.fast_sta(*c)
        packed = np.packbits(m.numpy().astype(np.uint8).reshape(-1))
        assert np.array_equal(packed, golden["sta." + "_".join(map(str, c))].numpy()), c


def test_fractal_perm(golden):
    assert torch.equal(O.fractal_perm((6, 16, 16)), golden["fractal.perm.6x16x16"])


def test_nabla_mask(golden, golden_meta):
    sta = golden["nabla.sta"].bool()
    assert torch.equal(O.fast_sta(6, 2, 2, 3, 1, 1), sta)
    m = O.nabla_block_mask(golden["nabla.q"], golden["nabla.k"], sta, golden_meta["nabla_attention"]["P"], "fp32")
    ref = golden["nabla.mask"].bool()
    assert (m != ref).sum().item() == 0
    assert 0.3 < ref.float().mean().item() < 0.95  # the mask is non-trivial


def test_full_forward_nabla(golden, golden_meta, tiny_sd, cfg):
    attn = golden_meta["nabla_attention"]
    sparse = O.get_sparse_params(attn, (6, 32, 32), cfg.patch_size)
    pos = [torch.arange(6), torch.arange(16), torch.arange(16)]
    taps = {}
    out = O.dit_forward(tiny_sd, cfg, golden["nabla.fwd.x"], golden["fwd.text"], golden["fwd.pooled"],
                        golden["fwd.time"], pos, torch.arange(7), (1.0, 2.0, 2.0), sparse, "fp32", taps)
    for i, m in enumerate(taps["nabla_masks"]):
        assert torch.equal(m, golden[f"nabla.fwd.mask{i}"].bool()), i
    close(out, golden["nabla.fwd.out"], atol=1e-4, rtol=1e-4)


def test_sigma_schedule_and_trajectories(golden, golden_meta, tiny_sd, cfg):
    pos = [torch.arange(3), torch.arange(4), torch.arange(6)]
    te = {"text_embeds": golden["fwd.text"], "pooled_embed": golden["fwd.pooled"]}
    ne = {"text_embeds": golden["gen.null_text"], "pooled_embed": golden["gen.null_pooled"]}
    for steps, s, w in golden_meta["gen_cases"]:
        tag = f"gen.{steps}_{s}_{w}"
        sig = O.sigma_schedule(steps, s)
        close(sig[:-1] * 1000, golden[tag + ".times"], atol=1e-4, rtol=1e-6)
        if steps > 4:
            continue
        final, traj = O.generate(tiny_sd, cfg, golden["gen.noise"], steps, te, ne, pos, torch.arange(7),
                                 torch.arange(4), w, s, (1.0, 2.0, 2.0), None, "fp32", return_trajectory=True)
        close(torch.stack(traj), golden[tag + ".latents"], atol=2e-4, rtol=2e-4)
        close(final, golden[tag + ".final"], atol=2e-4, rtol=2e-4)


def test_generate_16_steps(golden, tiny_sd, cfg):
    pos = [torch.arange(3), torch.arange(4), torch.arange(6)]
    te = {"text_embeds": golden["fwd.text"], "pooled_embed": golden["fwd.pooled"]}
    final = O.generate(tiny_sd, cfg, golden["gen.noise"], 16, te, te, pos, torch.arange(7), torch.arange(7),
                       1.0, 5.0, (1.0, 2.0, 2.0), None, "fp32")
    close(final, golden["gen.16_5.0_1.0.final"], atol=5e-4, rtol=5e-4)


def test_generate_nabla_cfg(golden, golden_meta, tiny_sd, cfg):
    pos = [torch.arange(6), torch.arange(16), torch.arange(16)]
    te = {"text_embeds": golden["fwd.text"], "pooled_embed": golden["fwd.pooled"]}
    ne = {"text_embeds": golden["gen.null_text"], "pooled_embed": golden["gen.null_pooled"]}
    final = O.generate(tiny_sd, cfg, golden["gen.nabla.noise"], 2, te, ne, pos, torch.arange(7), torch.arange(4),
                       2.0, 5.0, (1.0, 2.0, 2.0), golden_meta["nabla_attention"], "fp32")
    close(final, golden["gen.nabla.final"], atol=3e-4, rtol=3e-4)


def test_manifest_matches_reference_state_dict(cfg):
    with open(os.path.join(os.path.dirname(__file__), "golden", "dit_lite_manifest.json")) as f:
        ref = json.load(f)
    mine = O.state_dict_manifest(O.DitConfig(**O.LITE_2B))
    assert list(mine.keys()) == list(ref.keys())
    for k in ref:
        assert list(mine[k]) == ref[k], k
    assert sum(math.prod(s) for s in mine.values()) == 2_007_702_848   # the "2B" of T2V Lite: 2.008e9 parameters


def test_bf16_mode_is_close_to_fp32(golden, tiny_sd, cfg):
    """The bf16-island mode must stay within bf16 noise of the fp32 oracle (sanity of rounding points)."""
    pos = [torch.arange(3), torch.arange(4), torch.arange(6)]
    a = O.dit_forward(tiny_sd, cfg, golden["fwd.x"], golden["fwd.text"], golden["fwd.pooled"], golden["fwd.time"],
                      pos, torch.arange(7), (1.0, 2.0, 2.0), None, "bf16")
    ref = golden["fwd.out"]
    rel = (a - ref).norm() / ref.norm()
    assert rel < 2e-2, rel
    assert torch.equal(a, a.bfloat16().float())  # velocity is bf16-valued


# ------------------------------------------------------------------------------------------ MagCache
@pytest.fixture(scope="module")
def mag_golden():
    from safetensors.torch import load_file
    import json
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return load_file(os.path.join(here, "magcache_tiny.safetensors")), json.load(open(os.path.join(here, "magcache_meta.json")))


def test_magcache_tables_and_skip_pattern_match_reference(mag_golden):
    """Ratio interpolation (np.round half-to-even) and every skip / run decision of the reference's magcache_forward."""
    T, meta = mag_golden
    for c in meta["cases"]:
        mc = O.MagCache(c["ratios"], c["num_steps"], c["no_cfg"])
        assert np.array_equal(mc.mag_ratios, T[f"mag.{c['tag']}.table"].numpy()), c["tag"]
        ran = []
        for _ in c["ran_blocks"]:
            ran.append(0 if mc.decide() else 1)
            mc.advance()
        assert ran == c["ran_blocks"], c["tag"]
        assert mc.cnt == 0
        assert 0 < sum(ran) < len(ran)


@pytest.mark.parametrize("tag", ["sft_12", "nocfg_9", "hand_10", "sft_50"])
def test_magcache_generate_matches_reference(mag_golden, golden, tiny_sd, cfg, tag):
    T, meta = mag_golden
    c = [c for c in meta["cases"] if c["tag"] == tag][0]
    te = {"text_embeds": golden["fwd.text"], "pooled_embed": golden["fwd.pooled"]}
    ne = {"text_embeds": golden["gen.null_text"], "pooled_embed": golden["gen.null_pooled"]}
    pos = [torch.arange(3), torch.arange(4), torch.arange(6)]
    mc = O.MagCache(c["ratios"], c["num_steps"], c["no_cfg"])
    final = O.generate(tiny_sd, cfg, golden["gen.noise"], c["num_steps"], te, ne, pos, torch.arange(7), torch.arange(4),
                       c["guidance_weight"], c["scheduler_scale"], magcache=mc)
    assert mc.ran_blocks == c["ran_blocks"]
    close(final, T[f"mag.{tag}.final"], atol=5e-4, rtol=5e-4)


# ------------------------------------------------------------------------------------------ full 2B-Lite width
@pytest.fixture(scope="module")
def fullwidth():
    import json
    from safetensors.torch import load_file
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return load_file(os.path.join(here, "dit_fullwidth.safetensors")), json.load(open(os.path.join(here, "dit_fullwidth_meta.json")))


def test_full_width_forward_matches_reference(fullwidth):
    """D = 1792 / 28 heads / FF 7168 / text 3584+768, 1 text + 2 visual blocks, weights regenerated from the documented
    seed: the oracle reproduces the reference's sampled outputs and whole-tensor sums (oracle/gen_golden_fullwidth.py)."""
    T, meta = fullwidth
    c = dict(meta["config"])
    c["patch_size"], c["axes_dims"] = tuple(c["patch_size"]), tuple(c["axes_dims"])
    cfg = O.DitConfig(**c)
    sd = O.synthetic_state_dict(cfg, seed=meta["weights_seed"])
    x = torch.cat([T["x"], torch.zeros(5, 16, 16, 17)], dim=-1)
    pos = [torch.arange(5), torch.arange(8), torch.arange(8)]
    out = O.dit_forward(sd, cfg, x, T["text"], T["pooled"], T["time"], pos, torch.arange(37), (1.0, 2.0, 2.0), None, "fp32")
    assert list(out.shape) == meta["out_shape"]
    close(out.reshape(-1)[T["sample_idx"]], T["sample_val"], atol=2e-4, rtol=2e-4)
    assert abs(float(out.double().sum()) - meta["out_sum"]) <= 2e-2 * abs(meta["out_sum"]) ** 0.5 + 0.05
    assert abs(float(out.double().pow(2).sum()) - meta["out_sumsq"]) <= 1e-3 * meta["out_sumsq"]
