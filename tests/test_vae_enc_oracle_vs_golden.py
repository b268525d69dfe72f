"""The CPU oracle of the VAE ENCODE path (oracle/vae_oracle.py: encoder_forward, tiled_encode) against vectors produced by
the reference's own vae.py (oracle/gen_golden_vae_enc.py -> tests/golden/vae_enc_tiny.safetensors) — SURVEY §8 f4."""
import json
import os

import pytest
import torch
from safetensors.torch import load_file

from oracle import vae_oracle as V

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def g():
    return load_file(os.path.join(HERE, "vae_enc_tiny.safetensors"))


@pytest.fixture(scope="module")
def meta():
    return json.load(open(os.path.join(HERE, "vae_enc_meta.json")))


@pytest.fixture(scope="module")
def sd(g):
    return {k[2:]: v for k, v in g.items() if k.startswith("w.")}


def close(a, b, tol=2e-5):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    assert err <= tol * max(1.0, b.abs().max().item()), err


def test_strided_causal_conv(g, sd):
    close(V.causal_conv3d_strided(sd, "encoder.down_blocks.0.downsamplers.0.conv", g["m.down_hw.x"], (1, 2, 2), "fp32"), g["m.down_hw.out"])
    close(V.causal_conv3d_strided(sd, "encoder.down_blocks.1.downsamplers.0.conv", g["m.down_thw.x"], (2, 2, 2), "fp32"), g["m.down_thw.out"])
    close(V.causal_conv3d_strided(sd, "encoder.down_blocks.1.downsamplers.0.conv", g["m.down_thw1.x"], (2, 2, 2), "fp32"), g["m.down_thw1.out"])
    assert V.down_schedule(4) == [(1, 2, 2), (2, 2, 2), (2, 2, 2), None]


def test_encoder_forward(g, sd, meta):
    cfg = meta["config"]
    close(V.encoder_forward(sd, g["e.x"], cfg, "fp32"), g["e.moments"], 5e-5)
    close(V.encoder_forward(sd, g["e.img.x"], cfg, "fp32"), g["e.img.moments"], 5e-5)
    assert tuple(g["e.moments"].shape) == (1, 32, 3, 4, 3)     # (9-1)/4+1 frames, 32/8 x 24/8, mean | logvar


@pytest.mark.parametrize("case", ["tt", "st", "bt"])
def test_tiled_encode(g, sd, meta, case):
    c = meta[{"tt": "temporal_tiling_case", "st": "spatial_tiling_case", "bt": "both_tiling_case"}[case]]
    out = V.tiled_encode(sd, g[f"e.{case}.x"], meta["config"], c["tile"], c["stride"], "fp32")
    close(out, g[f"e.{case}.out"], 5e-5)


def test_bf16_mode_close_to_fp32(g, sd, meta):
    a = V.encoder_forward(sd, g["e.x"], meta["config"], "bf16")
    rel = ((a - g["e.moments"]).norm() / g["e.moments"].norm()).item()
    assert rel < 3e-2, rel


def test_gaussian_moments():
    h = torch.randn(1, 8, 2, 3, 3) * 20
    mean, logvar, std = V.gaussian_moments(h)
    assert torch.equal(mean, h[:, :4]) and float(logvar.max()) <= 20.0 and float(logvar.min()) >= -30.0
    assert torch.allclose(std, torch.exp(0.5 * logvar))
