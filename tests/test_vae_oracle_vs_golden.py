"""Pins oracle/vae_oracle.py against vectors produced by the reference's own kandinsky/models/vae.py
(oracle/gen_golden_vae.py; fp32, CPU).  The mid-block Attention is diffusers code outside the reference tree
(shimmed by its definition when the vectors were generated): parity for it is pinned only through the call site."""
import json
import os

import pytest
import torch
from safetensors.torch import load_file

from oracle import vae_oracle as V

G = os.path.join(os.path.dirname(__file__), "golden")
TOL = dict(atol=3e-5, rtol=3e-5)


@pytest.fixture(scope="module")
def g():
    return load_file(os.path.join(G, "vae_tiny.safetensors"))


@pytest.fixture(scope="module")
def meta():
    return json.load(open(os.path.join(G, "vae_meta.json")))


@pytest.fixture(scope="module")
def sd(g):
    return {k[2:]: v for k, v in g.items() if k.startswith("w.")}


def close(a, b, **kw):
    t = dict(TOL)
    t.update(kw)
    torch.testing.assert_close(a.float(), b.float(), **t)


def test_causal_conv_and_resnets(g, sd):
    close(V.causal_conv3d(sd, "decoder.mid_block.resnets.0.conv1", g["m.conv.x"], "fp32"), g["m.conv.out"])
    close(V.resnet_block(sd, "decoder.mid_block.resnets.0", g["m.conv.x"], 8, "fp32"), g["m.resnet.out"])
    close(V.resnet_block(sd, "decoder.up_blocks.3.resnets.0", g["m.resnet_sc.x"], 8, "fp32"), g["m.resnet_sc.out"])


def test_upsamplers(g, sd):
    close(V.upsample(sd, "decoder.up_blocks.0.upsamplers.0", g["m.conv.x"], (1, 2, 2), "fp32"), g["m.up_hw.out"])
    close(V.upsample(sd, "decoder.up_blocks.1.upsamplers.0", g["m.conv.x"], (2, 2, 2), "fp32"), g["m.up_thw.out"])
    close(V.upsample(sd, "decoder.up_blocks.1.upsamplers.0", g["m.up_thw1.x"], (2, 2, 2), "fp32"), g["m.up_thw1.out"])
    assert g["m.up_thw.out"].shape[2] == 5  # 3 frames -> 1 + 2*2


def test_mask_and_mid_block(g, sd):
    assert torch.equal(V.causal_attention_mask(3, 4), g["m.mask.3x4"])
    x = g["m.conv.x"]
    h = V.resnet_block(sd, "decoder.mid_block.resnets.0", x, 8, "fp32")
    h = V.mid_attention(sd, "decoder.mid_block.attentions.0", h, 8, "fp32")
    h = V.resnet_block(sd, "decoder.mid_block.resnets.1", h, 8, "fp32")
    close(h, g["m.mid.out"], atol=1e-4, rtol=1e-4)


def test_decoder_and_decode(g, sd, meta):
    out = V.decoder_forward(sd, g["d.z"], meta["config"], "fp32")
    assert tuple(out.shape) == (1, 3, 9, 48, 40)
    close(out, g["d.decoder"], atol=2e-4, rtol=2e-4)


@pytest.mark.parametrize("case,key", [("temporal_tiling_case", "tt"), ("spatial_tiling_case", "st"), ("both_tiling_case", "bt")])
def test_tiled_decodes(g, sd, meta, case, key):
    c = meta[case]
    out = V.tiled_decode(sd, g[f"d.{key}.z"], meta["config"], c["tile"], c["stride"], "fp32")
    assert out.shape == g[f"d.{key}.out"].shape
    close(out, g[f"d.{key}.out"], atol=3e-4, rtol=3e-4)


def test_blends(g):
    for name, dim, ext in (("t", 2, 3), ("v", 3, 4), ("h", 4, 5)):
        close(V.blend(g["b.a"], g["b.b"].clone(), ext, dim, "fp32"), g["b." + name], atol=1e-6, rtol=1e-6)


def test_tiling_policy(meta):
    tt = {int(k): tuple(v) for k, v in meta["opt_temporal_tiling"].items()}
    st = {int(k): tuple(v) for k, v in meta["opt_spatial_tiling"].items()}
    for key, (tile, stride) in meta["dec_tiling"].items():
        shape = [int(x) for x in key.split("x")]
        a, b = V.get_dec_optimal_tiling(shape, tt, st)
        assert list(a) == tile and list(b) == stride, key
    # BASELINE config 2: 5 s 768x512 -> 17-frame tiles, stride 8 (14 temporal tiles of 5 latent frames)
    assert meta["dec_tiling"]["1x16x31x64x96"] == [[1, 17, 512, 768], [8, 512, 768]]
    assert len(range(0, 31 - 4 + 1, 2)) == 14


def test_bf16_mode_close_to_fp32(g, sd, meta):
    a = V.decoder_forward(sd, g["d.z"], meta["config"], "bf16")
    rel = (a - g["d.decoder"]).norm() / g["d.decoder"].norm()
    assert rel < 3e-2, rel


def test_production_width_fixture_pins_the_oracle():
    """tests/golden/vae_fullwidth.safetensors (oracle/gen_golden_vae_fullwidth.py, build container, ~6 min per evaluation): the
    REFERENCE decoder at 128/256/512/512 channels on the production (5, 64, 96) latent tile, and oracle/vae_oracle.py on the same
    seeded weights in fp32 and bf16 mode — 32768 sampled outputs each.  Re-running a 119-TFLOP decode on the host is not a unit test;
    what is checked here is the committed evidence: the fp32 oracle reproduces the reference at production width (<= 1e-5 relative,
    measured 1.7e-6), the bf16-autocast restatement sits at the expected ~1e-2, and the weights recipe the GPU test regenerates is
    the one the fixture was made with (first tensors' checksums)."""
    import json
    import os
    import torch
    from safetensors.torch import load_file
    from oracle import vae_oracle as V
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g, fm = load_file(os.path.join(here, "vae_fullwidth.safetensors")), json.load(open(os.path.join(here, "vae_fullwidth_meta.json")))
    ref, o32, ob = g["ref.sample_val"], g["oracle32.sample_val"], g["oraclebf16.sample_val"]
    assert ref.numel() == 32768 and fm["tile"] == [5, 64, 96] and fm["ref"]["out_shape"] == [1, 3, 17, 512, 768]
    assert ((o32 - ref).norm() / ref.norm()).item() <= 1e-5
    assert abs(fm["oracle32"]["out_sumsq"] - fm["ref"]["out_sumsq"]) <= 1e-6 * fm["ref"]["out_sumsq"]
    assert 2e-3 <= ((ob - ref).norm() / ref.norm()).item() <= 2e-2
    assert fm["load_state_dict"] == "<All keys matched successfully>"
    idx = g["sample_idx"]
    assert torch.equal(idx, torch.randperm(3 * 17 * 512 * 768, generator=torch.Generator().manual_seed(fm["index_seed"]))[:32768].sort().values)
    man = json.load(open(os.path.join(here, "vae_meta.json")))["full_manifest"]
    small = {k: v for k, v in man.items() if k in ("post_quant_conv.weight", "decoder.conv_in.conv.bias", "decoder.conv_norm_out.weight")}
    sd = V.synthetic_decoder_state_dict(man, fm["weights_seed"])
    assert len(sd) == len(man) == 140 and all(tuple(sd[k].shape) == tuple(man[k]) for k in small)
