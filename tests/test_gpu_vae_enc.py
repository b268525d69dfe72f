"""GPU parity of the VAE ENCODE path (C ABI k5_vae_encode_tile / k5_conv3d_strided_bf16 + the host mirror's
tiling) against oracle/vae_oracle.py in bf16-autocast mode — the oracle itself is pinned to the reference's own encoder
(tests/test_vae_enc_oracle_vs_golden.py).  Tolerance as for the decoder: relative L2 <= 2e-2 on the moments of a tile
(bf16 activations through ~25 conv / GroupNorm layers), blends bit-exact."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vae_oracle as V  # noqa: E402

CFG = dict(latent_channels=16, out_channels=3, block_out_channels=(64, 64, 128, 128), layers_per_block=2, norm_num_groups=16)


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def bfr(x):
    return x.bfloat16().float()


@pytest.fixture(scope="module")
def vae():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X")
    from kandinsky.models.vae import AutoencoderKLHunyuanVideo
    m = AutoencoderKLHunyuanVideo(**CFG)
    g = torch.Generator().manual_seed(15)
    sd = {}
    for k, p in m.state_dict().items():
        if "norm" in k and k.endswith("weight"):
            sd[k] = 1.0 + 0.2 * torch.randn(p.shape, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.1 * torch.randn(p.shape, generator=g)
        else:
            sd[k] = torch.randn(p.shape, generator=g) * (1.2 / (p[0].numel() ** 0.5))
    m.load_state_dict(sd, assign=True)
    return m.to("cuda:0"), sd


@pytest.mark.parametrize("st,dims,Cin,Cout", [((1, 2, 2), (5, 9, 8), 64, 64), ((2, 2, 2), (5, 8, 6), 128, 128), ((2, 2, 2), (1, 8, 6), 64, 72),
                                              ((2, 2, 2), (9, 33, 47), 128, 256)])
def test_strided_causal_conv_kernel(st, dims, Cin, Cout):
    """HunyuanVideoDownsampleCausal3D: replicate pad (W 1,1; H 1,1; T 2,0), stride (st_t, 2, 2), padding 0 — odd extents,
    a single frame, and a shape with several 128-row tiles."""
    from kandinsky import _engine as E
    torch.manual_seed(2)
    Ts, Hs, Ws = dims
    x = bfr(torch.randn(1, Cin, Ts, Hs, Ws))
    w = bfr(torch.randn(Cout, Cin, 3, 3, 3) * 0.05)
    b = bfr(torch.randn(Cout) * 0.1)
    ref = V.causal_conv3d_strided({"c.conv.weight": w, "c.conv.bias": b}, "c", x, st, "bf16")
    To, Ho, Wo = ref.shape[2:]
    assert (To, Ho, Wo) == ((Ts - 1) // st[0] + 1, (Hs - 1) // 2 + 1, (Ws - 1) // 2 + 1)
    xd = x[0].permute(1, 2, 3, 0).contiguous().cuda().bfloat16()
    wd = w.permute(0, 2, 3, 4, 1).reshape(Cout, 27 * Cin).contiguous().cuda().bfloat16()
    bd = b.cuda()
    out = torch.empty(To * Ho * Wo, Cout, dtype=torch.bfloat16, device="cuda")
    E.check(E.lib().k5_conv3d_strided_bf16(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), Ts, Hs, Ws, Cin, Cout, st[0], st[1],
                                           Cout, E.stream_ptr()))
    torch.cuda.synchronize()
    want = ref[0].permute(1, 2, 3, 0).reshape(-1, Cout)
    err = (out.float().cpu() - want).abs().max().item()
    assert err <= 2.0 ** -6 * max(1.0, want.abs().max().item()), (st, dims, err)


@pytest.mark.parametrize("shape", [(9, 32, 48), (1, 32, 32), (17, 16, 24)])
def test_encode_tile_vs_oracle(vae, shape):
    """quant_conv(encoder(x)) of one tile: a 9-frame clip, a single image (the I2V conditioning frame), 17 frames."""
    m, sd = vae
    T, H, W = shape
    x = torch.randn(1, 3, T, H, W, generator=torch.Generator().manual_seed(T)).clamp(-1, 1)
    got = m._encode_tile(x.cuda())
    assert tuple(got.shape) == (1, 32, (T - 1) // 4 + 1, H // 8, W // 8) and got.dtype == torch.bfloat16
    ref16 = V.encoder_forward(sd, x, CFG, "bf16")
    ref32 = V.encoder_forward(sd, x, CFG, "fp32")
    print(f"encode tile {shape}: engine vs bf16 oracle {rel(got, ref16):.3e}; bf16 oracle vs fp32 oracle {rel(ref16, ref32):.3e}")
    assert rel(got, ref16) <= 2e-2, rel(got, ref16)
    assert rel(got, ref32) <= 4e-2, rel(got, ref32)
    again = m._encode_tile(x.cuda())
    assert torch.equal(got, again)                      # deterministic


@pytest.mark.parametrize("case", ["temporal", "spatial", "both"])
def test_tiled_encode_policy_vs_oracle(vae, case):
    """The host mirror's _encode with apply_tiling (temporal tiles of 8+1 frames / stride 4, 48x48 spatial tiles / stride 32)
    against the oracle's tiled_encode driven by the ENGINE's tile encoder: slicing, dropped frames and blends bit-exact."""
    m, sd = vae
    tile, stride, shape = {"temporal": ((1, 9, 16, 16), (4, 16, 16), (17, 16, 16)),
                           "spatial": ((1, 9, 48, 48), (8, 32, 32), (5, 80, 112)),
                           "both": ((1, 9, 48, 48), (4, 32, 32), (17, 80, 80))}[case]
    x = torch.randn(1, 3, *shape, generator=torch.Generator().manual_seed(len(case))).clamp(-1, 1)
    m.apply_tiling(tile, stride)
    got = m._encode(x.cuda())
    ref = V.tiled_encode(sd, x, CFG, tile, stride, "bf16", encode_tile=lambda t: m._encode_tile(t.cuda()).float().cpu())
    assert tuple(got.shape) == (1, 32, (shape[0] - 1) // 4 + 1, shape[1] // 8, shape[2] // 8)
    assert torch.equal(got.float().cpu(), ref)
    full = V.tiled_encode(sd, x, CFG, tile, stride, "bf16")
    assert rel(got, full) <= 2e-2, rel(got, full)


def test_encode_api_and_round_trip(vae):
    """encode(x).latent_dist: mean / logvar split, clamp, mode() / sample(); forward() = decode(mode(encode(x))) runs end to end
    (random weights: only shapes / finiteness); a decode-only load refuses encode loudly."""
    from kandinsky.models.vae import AutoencoderKLHunyuanVideo
    m, sd = vae
    x = torch.randn(1, 3, 5, 32, 32, generator=torch.Generator().manual_seed(3)).clamp(-1, 1).cuda()
    post = m.encode(x).latent_dist
    h = m._encode_tile(x)
    assert post.mean.dtype == h.dtype == torch.bfloat16       # the moments keep the encoder's dtype, as diffusers' class does
    assert torch.equal(post.mean, h[:, :16]) and torch.equal(post.mode(), post.mean)
    assert torch.equal(post.logvar, h[:, 16:].clamp(-30, 20))
    xb = torch.cat([x, x.flip(-1)], 0)                        # batch of 2: per-sample encode, the caller's tiling choice kept
    pb = m.encode(xb, opt_tiling=False).latent_dist.parameters
    assert tuple(pb.shape) == (2, 32) + tuple(h.shape[2:]) and torch.equal(pb[:1], m.encode(x, opt_tiling=False).latent_dist.parameters)
    g = torch.Generator(device="cuda").manual_seed(1)
    s1 = post.sample(generator=g)
    assert s1.shape == post.mean.shape and torch.isfinite(s1).all() and not torch.equal(s1, post.mean)
    rec = m(x).sample
    assert tuple(rec.shape) == (1, 3, 5, 32, 32) and torch.isfinite(rec.float()).all()
    m2 = AutoencoderKLHunyuanVideo(**CFG)
    m2.load_state_dict({k: v for k, v in sd.items() if not (k.startswith("encoder.") or k.startswith("quant_conv"))}, assign=True)
    m2 = m2.to("cuda:0")
    assert torch.isfinite(m2.decode(post.mean).sample.float()).all()         # decode works without the encoder half
    with pytest.raises(RuntimeError, match="without its encoder"):
        m2.encode(x)
