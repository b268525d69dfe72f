"""GPU parity of the VAE decode path (C ABI k5_vae_decode_tile / k5_blend_bf16 + host tiling) against
oracle/vae_oracle.py in bf16-autocast mode.  Tolerance: relative L2 <= 2e-2 on decoded tiles (bf16 activations through
~30 conv / GroupNorm layers; the bf16 oracle itself sits ~1e-2 from the fp32 oracle), blends bit-exact."""
import json
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vae_oracle as V  # noqa: E402

CFG = dict(latent_channels=16, out_channels=3, block_out_channels=(64, 64, 128, 128), layers_per_block=2, norm_num_groups=16)


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def vae():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X")
    from kandinsky.models.vae import AutoencoderKLHunyuanVideo
    m = AutoencoderKLHunyuanVideo(**CFG)
    g = torch.Generator().manual_seed(5)
    sd = {}
    for k, p in m.state_dict().items():
        if "norm" in k and k.endswith("weight"):
            sd[k] = 1.0 + 0.2 * torch.randn(p.shape, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.1 * torch.randn(p.shape, generator=g)
        else:
            sd[k] = torch.randn(p.shape, generator=g) * (1.5 / (p[0].numel() ** 0.5))
    m.load_state_dict(sd, assign=True)
    return m.to("cuda:0"), sd


def bfr(x):
    return x.bfloat16().float()


def test_conv3d_kernel_with_upsample_and_residual(vae):
    import ctypes as C
    from kandinsky import _engine as E
    torch.manual_seed(0)
    Ts, Hs, Ws, Cin, Cout = 3, 5, 7, 64, 72
    x = bfr(torch.randn(1, Cin, Ts, Hs, Ws))
    w = bfr(torch.randn(Cout, Cin, 3, 3, 3) * 0.05)
    b = bfr(torch.randn(Cout) * 0.1)
    sd = {"c.conv.weight": w, "c.conv.bias": b}
    for up_t, up_s in ((1, 1), (1, 2), (2, 2)):
        xin = x
        if up_t > 1 or up_s > 1:  # the upsample half of HunyuanVideoUpsampleCausal3D, conv applied below
            first = torch.nn.functional.interpolate(x[:, :, 0], scale_factor=(up_s, up_s), mode="nearest").unsqueeze(2)
            rest = torch.nn.functional.interpolate(x[:, :, 1:], scale_factor=(up_t, up_s, up_s), mode="nearest")
            xin = torch.cat([first, rest], 2)
        ref = V.causal_conv3d(sd, "c", xin, "bf16")
        To, Ho, Wo = ref.shape[2:]
        resid = bfr(torch.randn(To * Ho * Wo, Cout))
        ref_r = bfr(ref[0].permute(1, 2, 3, 0).reshape(-1, Cout) + resid)
        xd = x[0].permute(1, 2, 3, 0).contiguous().cuda().bfloat16()          # [T][H][W][C]
        wd = w.permute(0, 2, 3, 4, 1).reshape(Cout, 27 * Cin).contiguous().cuda().bfloat16()  # [Cout][27][Cin]
        out = torch.empty(To * Ho * Wo, Cout, dtype=torch.bfloat16, device="cuda")
        rd, bd = resid.cuda().bfloat16(), b.cuda()
        for use_res in (False, True):
            E.check(E.lib().k5_conv3d_bf16(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), Ts, Hs, Ws, Cin, Cout,
                                           up_t, up_s, Cout, rd.data_ptr() if use_res else None, Cout, E.stream_ptr()))
            torch.cuda.synchronize()
            want = ref_r if use_res else ref[0].permute(1, 2, 3, 0).reshape(-1, Cout)
            err = (out.float().cpu() - want).abs().max().item()
            assert err <= 2.0 ** -6 * max(1.0, want.abs().max().item()), (up_t, up_s, use_res, err)


@pytest.mark.parametrize("Cin,Cout,up,dims", [(128, 128, (1, 1), (5, 112, 128)), (128, 256, (2, 2), (3, 60, 64)), (256, 128, (1, 2), (4, 64, 72)),
                                              (512, 512, (2, 2), (3, 36, 48)), (128, 128, (1, 1), (5, 113, 127)),   # M not a multiple of 256, odd width
                                              (256, 128, (1, 2), (4, 64, 64)),    # halo form of the 256 x 128 tile with four slabs (4 weight stages, one barrier per K-tile)
                                              (128, 128, (1, 1), (2, 104, 672)),  # halo form on the 1280 x 768 clips' full-resolution tile width: 13 x 21 patches per frame
                                              (256, 256, (2, 2), (2, 52, 112))])  # ... and an upsampling conv onto 13 x 7 patches (odd counts in both directions)
def test_conv3d_four_wave_kernel(vae, Cin, Cout, up, dims):
    """Shapes in the range of the 4-wave 256-row kernels (conv3d_w4.hip: Cin % 128 == 0, Cout = 128 or % 256, >= 256 tiles): output
    frames of whole 8 x 32 patches take the LDS-halo form (cases 1, 2, 4, 6, 7, 8), the others the per-tap gather (cases 3, 5): source
    offsets (replicate pad, causal T, folded nearest upsample), both tile shapes, residual epilogue, ragged M."""
    from kandinsky import _engine as E
    torch.manual_seed(1)
    Ts, Hs, Ws = dims
    up_t, up_s = up
    x = bfr(torch.randn(1, Cin, Ts, Hs, Ws))
    w = bfr(torch.randn(Cout, Cin, 3, 3, 3) * 0.03)
    b = bfr(torch.randn(Cout) * 0.1)
    xin = x
    if up_t > 1 or up_s > 1:
        first = torch.nn.functional.interpolate(x[:, :, 0], scale_factor=(up_s, up_s), mode="nearest").unsqueeze(2)
        rest = torch.nn.functional.interpolate(x[:, :, 1:], scale_factor=(up_t, up_s, up_s), mode="nearest")
        xin = torch.cat([first, rest], 2)
    ref = V.causal_conv3d({"c.conv.weight": w, "c.conv.bias": b}, "c", xin, "bf16")
    To, Ho, Wo = ref.shape[2:]
    assert (To * Ho * Wo + 255) // 256 * max(1, Cout // 256) >= 256        # really in the 4-wave kernel's range
    want = ref[0].permute(1, 2, 3, 0).reshape(-1, Cout)
    resid = bfr(torch.randn(To * Ho * Wo, Cout))
    xd = x[0].permute(1, 2, 3, 0).contiguous().cuda().bfloat16()
    wd = w.permute(0, 2, 3, 4, 1).reshape(Cout, 27 * Cin).contiguous().cuda().bfloat16()
    out = torch.empty(To * Ho * Wo, Cout, dtype=torch.bfloat16, device="cuda")
    rd, bd = resid.cuda().bfloat16(), b.cuda()
    for use_res in (False, True):
        out.zero_()
        E.check(E.lib().k5_conv3d_bf16(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), Ts, Hs, Ws, Cin, Cout, up_t, up_s,
                                       Cout, rd.data_ptr() if use_res else None, Cout, E.stream_ptr()))
        torch.cuda.synchronize()
        tgt = bfr(want + resid) if use_res else want
        err = (out.float().cpu() - tgt).abs()
        tol = 2.0 ** -6 * tgt.abs().clamp(min=1.0) + (2.0 ** -7 * want.abs() if use_res else 0)   # + a 1-ulp flip of the inner rounding
        assert not (err > tol).any(), (Cin, Cout, up, use_res, float(err.max()), int((err > tol).sum()))
        again = out.clone()
        E.check(E.lib().k5_conv3d_bf16(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), Ts, Hs, Ws, Cin, Cout, up_t, up_s,
                                       Cout, rd.data_ptr() if use_res else None, Cout, E.stream_ptr()))
        assert torch.equal(out, again)                                         # hand-counted waits: bit-reproducible


@pytest.mark.parametrize("Cin,dims", [(128, (3, 6, 300)), (64, (2, 5, 19)), (128, (4, 3, 256)), (192, (2, 4, 40))])
def test_conv3d_three_output_channels(vae, Cin, dims):
    """The decoder's conv_out (Cin -> 3, vae.py:689) has its own kernel (conv3d.hip conv3d_out3_kernel: per source frame a K = Cin GEMM
    onto the 27 (output, kh, kw) columns on MFMA, then a 27-term shift-and-add through LDS): several 8 x 32 tiles with ragged edges in
    both directions, a single partial tile, Cin = 64 and 128; Cin = 192 is outside its range and takes the GEMM-tile kernel.
    Replicate padding in H / W and the causal front padding in T are what the oracle does."""
    from kandinsky import _engine as E
    torch.manual_seed(2)
    Ts, Hs, Ws = dims
    x = bfr(torch.randn(1, Cin, Ts, Hs, Ws))
    w = bfr(torch.randn(3, Cin, 3, 3, 3) * 0.05)
    b = bfr(torch.randn(3) * 0.1)
    ref = V.causal_conv3d({"c.conv.weight": w, "c.conv.bias": b}, "c", x, "bf16")
    want = ref[0].permute(1, 2, 3, 0).reshape(-1, 3)
    xd = x[0].permute(1, 2, 3, 0).contiguous().cuda().bfloat16()
    wd = w.permute(0, 2, 3, 4, 1).reshape(3, 27 * Cin).contiguous().cuda().bfloat16()
    bd = b.cuda()
    out = torch.full((Ts * Hs * Ws, 3), float("nan"), dtype=torch.bfloat16, device="cuda")
    E.check(E.lib().k5_conv3d_bf16(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), Ts, Hs, Ws, Cin, 3, 1, 1, 3, None, 3,
                                   E.stream_ptr()))
    torch.cuda.synchronize()
    err = (out.float().cpu() - want).abs()
    assert not (err > 2.0 ** -6 * want.abs().clamp(min=1.0)).any(), float(err.max())


@pytest.mark.parametrize("M,C,G", [(90, 64, 16), (1000, 128, 16), (3000, 512, 32), (77, 256, 32)])
def test_groupnorm_silu_kernel(M, C, G):
    from kandinsky import _engine as E
    g = torch.Generator().manual_seed(M)
    x = bfr(torch.randn(M, C, generator=g) * 2 + 0.5)
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    ws = torch.empty(E.lib().k5_groupnorm_workspace_size(M, G), dtype=torch.uint8, device="cuda")
    out = torch.empty(M, C, dtype=torch.bfloat16, device="cuda")
    xd, gd, bd = x.cuda().bfloat16(), gamma.cuda(), beta.cuda()
    for silu in (0, 1):
        E.check(E.lib().k5_groupnorm_bf16(xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), out.data_ptr(), M, C, G, 1e-6, silu,
                                          ws.data_ptr(), E.stream_ptr()))
        ref = torch.nn.functional.group_norm(x.t()[None], G, gamma, beta, 1e-6)[0].t()   # stats over (M, C/G) per group
        if silu:
            ref = torch.nn.functional.silu(ref)
        err = (out.float().cpu() - bfr(ref)).abs().max().item()
        assert err <= 2.0 ** -6 * max(1.0, ref.abs().max().item()), (M, C, G, silu, err)


@pytest.mark.parametrize("S,hw", [(300, 100), (4608, 1536), (9000, 3000), (20000, 5000), (20000, 20000)])
def test_causal_softmax_kernel(S, hw):
    """k5_causal_softmax_bf16 (mid-block attention mask vae.py:110-122 + softmax): row i over columns j < (i // hw + 1) * hw, zeros
    elsewhere up to ldp.  S = 300 takes the three-pass kernel, the others the register-resident ones (8 / 16 / 32 values per thread)."""
    from kandinsky import _engine as E
    g = torch.Generator(device="cuda").manual_seed(S)
    ld = (S + 7) // 8 * 8
    sc = torch.randn(S, ld, device="cuda", generator=g) * 3.0
    P = torch.full((S, ld), float("nan"), dtype=torch.bfloat16, device="cuda")
    E.check(E.lib().k5_causal_softmax_bf16(sc.data_ptr(), P.data_ptr(), S, hw, ld, ld, E.stream_ptr()), "k5_causal_softmax_bf16")
    torch.cuda.synchronize()
    rows = torch.arange(S, device="cuda")[:, None]
    cols = torch.arange(ld, device="cuda")[None, :]
    mask = (cols < (rows // hw + 1) * hw) & (cols < S)
    ref = torch.softmax(sc.masked_fill(~mask, float("-inf")), dim=-1)
    assert torch.isfinite(P.float()).all()
    assert (P.float()[~mask] == 0).all()
    assert (P.float() - ref).abs().max().item() <= 2 ** -8 * ref.max().item() + 1e-6          # one bf16 rounding of a probability
    assert (P.float().sum(-1) - 1).abs().max().item() <= 2e-2


@pytest.mark.parametrize("S,hw", [(4608, 1536), (3000, 700), (2048, 2048), (1000, 37), (6144, 6144 // 5 + 1)])
def test_mid_block_attention_kernel_c512(S, hw):
    """k5_vae_attention512_bf16 (one head, d = 512, frame-causal, scores never materialised) against fp32 softmax(scale q k^T + mask) v
    of the same bf16 q, k, v: frame boundaries inside 32-key tiles and 64-query blocks (hw = 700, 37), S not a multiple of 64 / 32,
    a single frame (no masking at all), activation-sized logits (|s| up to ~20 after the 1/sqrt(512) scale)."""
    from kandinsky import _engine as E
    g = torch.Generator(device="cuda").manual_seed(S + hw)
    C = 512
    qk = (torch.randn(S, 2 * C, device="cuda", generator=g) * 1.5).bfloat16()
    v = torch.randn(S, C, device="cuda", generator=g).bfloat16()
    ldvt = (S + 31) // 32 * 32
    vt = torch.full((C, ldvt), 3.0e30, dtype=torch.bfloat16, device="cuda")            # pad columns: finite garbage, weight exactly 0
    vt[:, :S] = v.t()
    out = torch.full((S, C), float("nan"), dtype=torch.bfloat16, device="cuda")
    scale = 1.0 / math.sqrt(C)
    E.check(E.lib().k5_vae_attention512_bf16(qk.data_ptr(), qk.data_ptr() + 2 * C, vt.data_ptr(), out.data_ptr(), S, hw, 2 * C, ldvt, C,
                                             scale, E.stream_ptr()), "k5_vae_attention512_bf16")
    torch.cuda.synchronize()
    q, k = qk[:, :C].float(), qk[:, C:].float()
    rows = torch.arange(S, device="cuda")[:, None]
    cols = torch.arange(S, device="cuda")[None, :]
    sc = (q @ k.t()) * scale
    sc = sc.masked_fill(cols >= (rows // hw + 1) * hw, float("-inf"))
    ref = torch.softmax(sc, dim=-1) @ v.float()
    assert torch.isfinite(out.float()).all()
    err = (out.float() - ref).abs()
    tol = 2.0 ** -6 * ref.abs() + 6e-3         # bf16 probabilities and outputs: two roundings on top of the fp32 reference
    assert not (err > tol).any(), (float(err.max()), int((err > tol).sum()))


def test_decode_tile_vs_oracle(vae):
    m, sd = vae
    z = torch.randn(1, 16, 3, 6, 5, generator=torch.Generator().manual_seed(1))
    out = m._decode_tile(z.cuda())
    assert tuple(out.shape) == (1, 3, 9, 48, 40) and out.dtype == torch.bfloat16
    ref = V.decoder_forward(sd, z, CFG, "bf16")
    ref32 = V.decoder_forward(sd, z, CFG, "fp32")
    assert rel(out, ref) <= 2e-2, rel(out, ref)
    assert rel(out, ref32) <= 4e-2, (rel(out, ref32), rel(ref, ref32))


def test_single_frame_tile(vae):
    m, sd = vae
    z = torch.randn(1, 16, 1, 4, 4, generator=torch.Generator().manual_seed(2))
    out = m._decode_tile(z.cuda())
    assert tuple(out.shape) == (1, 3, 1, 32, 32)
    assert rel(out, V.decoder_forward(sd, z, CFG, "bf16")) <= 2e-2


@pytest.mark.parametrize("shape,tile,stride", [((1, 16, 7, 4, 4), (1, 9, 32, 32), (4, 32, 32)),      # temporal tiles
                                               ((1, 16, 2, 10, 14), (1, 9, 48, 48), (8, 32, 32)),    # spatial tiles
                                               ((1, 16, 5, 10, 10), (1, 9, 48, 48), (4, 32, 32))])   # both
def test_tiled_decode_vs_oracle(vae, shape, tile, stride):
    m, sd = vae
    z = torch.randn(*shape, generator=torch.Generator().manual_seed(3))
    m.apply_tiling(tile, stride)
    out = m._decode(z.cuda()).sample
    ref = V.tiled_decode(sd, z, CFG, tile, stride, "bf16")
    assert out.shape == ref.shape
    assert rel(out, ref) <= 2e-2, rel(out, ref)


def test_blend_kernel_bit_exact(vae):
    m, _ = vae
    g = torch.Generator().manual_seed(4)
    a, b = bfr(torch.randn(1, 3, 5, 6, 7, generator=g)), bfr(torch.randn(1, 3, 5, 6, 7, generator=g))
    for dim, ext in ((2, 3), (3, 4), (4, 5)):
        ref = V.blend(a.clone(), b.clone(), ext, dim, "bf16")
        got = m._blend(a.cuda().bfloat16(), b.cuda().bfloat16(), ext, dim)
        assert torch.equal(got.float().cpu(), ref), dim


def test_temporal_tiling_in_one_pass_equals_the_slice_blend_cat_loop(vae, monkeypatch):
    """Round 6: `_temporal_tiled_decode` decodes a tile straight out of the long latent (k5_vae_decode_tile_strided) and cross-fades + places it in the
    output in one pass (k5_blend_place_bf16) instead of the reference loop's slices, .contiguous() copies, blend_t and torch.cat (vae.py:1144-1204).
    Same numbers, bit for bit, as that loop (K5_VAE_LEGACY_GLUE=1 keeps it callable)."""
    m, _ = vae
    for shape, tile, stride in (((1, 16, 13, 4, 4), (1, 9, 32, 32), (4, 32, 32)), ((1, 16, 9, 4, 6), (1, 17, 32, 48), (8, 32, 48))):
        z = torch.randn(*shape, generator=torch.Generator().manual_seed(8)).cuda()
        m.apply_tiling(tile, stride)
        monkeypatch.setenv("K5_VAE_LEGACY_GLUE", "1")
        old = m._decode(z).sample
        monkeypatch.setenv("K5_VAE_LEGACY_GLUE", "0")
        new = m._decode(z).sample
        assert new.shape == old.shape and torch.equal(new, old), (shape, rel(new, old))


def test_frames_to_uint8_kernel_equals_the_torch_expression():
    from kandinsky.generation_utils import frames_to_uint8
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(1, 3, 5, 16, 24, generator=g) * 0.8).bfloat16().cuda()
    x.view(-1)[:8] = torch.tensor([-1.0, 1.0, -3.0, 3.0, 0.0, -0.0, 0.99609375, -0.99609375], dtype=torch.bfloat16)
    ref = ((x.clamp(-1.0, 1.0) + 1.0) * 127.5).to(torch.uint8)              # generation_utils.py:222-224 of the reference
    assert torch.equal(frames_to_uint8(x), ref)
    assert torch.equal(frames_to_uint8(x.float().cpu()), ((x.float().cpu().clamp(-1.0, 1.0) + 1.0) * 127.5).to(torch.uint8))   # not bf16-on-GPU: torch


def test_decode_picks_reference_tiling(vae):
    m, _ = vae
    meta = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vae_meta.json")))
    for key, (tile, stride) in meta["dec_tiling"].items():
        shape = [int(x) for x in key.split("x")]
        a, b = m.get_dec_optimal_tiling(shape)
        assert list(a) == tile and list(b) == stride


def test_generate_sample_postprocess_matches_reference_uint8(vae):
    m, sd = vae
    z = torch.randn(1, 16, 2, 4, 4, generator=torch.Generator().manual_seed(6))
    dec = m.decode(z.cuda()).sample
    u8 = ((dec.clamp(-1.0, 1.0) + 1.0) * 127.5).to(torch.uint8)           # generation_utils.py:222
    ref = V.postprocess_uint8(_r16(V.decoder_forward(sd, z, CFG, "bf16")))
    diff = (u8.int().cpu() - ref.int()).abs()
    hist = [round((diff == i).float().mean().item(), 5) for i in range(int(diff.max()) + 1)]
    print("uint8 |engine - oracle| histogram (fraction of pixels per grey-level difference):", hist)
    # Two bf16 evaluations of the same ~30-layer decoder (different fp32 summation orders) sit ~1e-2 apart in relative L2,
    # i.e. ~1 grey level of the 255 on this random-weight decoder — DESIGN.md §2 states exactly what is asserted here:
    assert diff.float().mean().item() < 1.0
    assert (diff <= 1).float().mean().item() >= 0.75 and (diff <= 3).float().mean().item() >= 0.99 and int(diff.max()) <= 8, hist


def _r16(x):
    return x.bfloat16()


def test_config5_spatial_tiling_at_real_size():
    """BASELINE config 5 (1280x768, 10 s) is the one shape whose decode tiles SPATIALLY at real size: the reference's policy for
    the (61, 96, 160) latent is 17-frame x 416 x 672 tiles at stride 8 x 352 x 608 (golden dec_tiling table, pinned to the reference).
    Full-width synthetic decoder, 9 latent frames of that 96 x 160 plane -> 3 temporal x 2 x 2 spatial tiles with all three blends.
    Expected = the ORACLE's tiling / blending policy driven by the engine's own tile decode, so the comparison isolates the host
    mirror's tile origins, crops, blend extents and order at real size: identical tile kernels, bit-exact blends -> equality."""
    from kandinsky.models.vae import AutoencoderKLHunyuanVideo
    meta = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "vae_meta.json")))
    tile, stride = meta["dec_tiling"]["1x16x61x96x160"]
    dev = "cuda:0"
    with torch.device("meta"):
        m = AutoencoderKLHunyuanVideo()
    g = torch.Generator(device=dev).manual_seed(0)
    sd = {}
    for k, p in m.state_dict().items():
        if "norm" in k and k.endswith("weight"):
            sd[k] = torch.ones(p.shape, device=dev)
        elif k.endswith("bias"):
            sd[k] = torch.zeros(p.shape, device=dev)
        else:
            sd[k] = (torch.randn(p.shape, device=dev, generator=g) / p[0].numel() ** 0.5).half()
    m.load_state_dict(sd, assign=True)
    a, b = m.get_dec_optimal_tiling([1, 16, 61, 96, 160])
    assert [list(a), list(b)] == [tile, stride]
    z = torch.randn(1, 16, 9, 96, 160, device=dev, generator=g)
    m.apply_tiling(tuple(tile), tuple(stride))
    calls = []

    def decode_tile(t):
        calls.append(tuple(t.shape[2:]))
        return m._decode_tile(t.contiguous()).float()

    out = m._decode(z).sample
    ref = V.tiled_decode(None, z, None, tile, stride, "bf16", decode_tile=decode_tile)
    assert calls == [(5, 52, 84)] * 12, calls                       # 3 temporal x 2 x 2 spatial tiles of 5 x 52 x 84 latents
    assert tuple(out.shape) == tuple(ref.shape) == (1, 3, 33, 768, 1280)
    assert torch.isfinite(out.float()).all() and out.float().std().item() > 1e-3
    assert torch.equal(out.float(), ref.float()), rel(out, ref)


# ------------------------------------------------------------------------------------------ production width
def _fused_stats_case(Cin, Cout, dims, up, use_res, seed):
    from kandinsky import _engine as E
    g = torch.Generator().manual_seed(seed)
    Ts, Hs, Ws = dims
    up_t, up_s = up
    x = bfr(torch.randn(1, Cin, Ts, Hs, Ws, generator=g))
    w = bfr(torch.randn(Cout, Cin, 3, 3, 3, generator=g) * 0.03)
    b = bfr(torch.randn(Cout, generator=g) * 0.1)
    xin = x
    if up_t > 1 or up_s > 1:
        first = torch.nn.functional.interpolate(x[:, :, 0], scale_factor=(up_s, up_s), mode="nearest").unsqueeze(2)
        rest = torch.nn.functional.interpolate(x[:, :, 1:], scale_factor=(up_t, up_s, up_s), mode="nearest")
        xin = torch.cat([first, rest], 2)
    conv = V.causal_conv3d({"c.conv.weight": w, "c.conv.bias": b}, "c", xin, "bf16")            # oracle, bf16 mode (vae.py:125-163)
    M = conv.shape[2] * conv.shape[3] * conv.shape[4]
    want = conv[0].permute(1, 2, 3, 0).reshape(M, Cout)
    resid = bfr(torch.randn(M, Cout, generator=g)) if use_res else None
    if use_res:
        want = bfr(want + resid)                                                             # resnet skip, vae.py:274
    gamma, beta = 1 + 0.2 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g)
    xd = x[0].permute(1, 2, 3, 0).contiguous().cuda().bfloat16()
    wd = w.permute(0, 2, 3, 4, 1).reshape(Cout, 27 * Cin).contiguous().cuda().bfloat16()
    return E, (Ts, Hs, Ws, M), xd, wd, b.cuda(), None if resid is None else resid.cuda().bfloat16(), gamma, beta, want


@pytest.mark.parametrize("Cin,Cout,dims,up,use_res", [(128, 128, (5, 112, 128), (1, 1), True), (256, 128, (4, 64, 72), (1, 2), False),
                                                     (512, 512, (3, 72, 96), (1, 1), True), (256, 256, (3, 50, 68), (2, 2), False),
                                                     (128, 128, (5, 113, 127), (1, 1), False)])
def test_conv_statistics_feed_groupnorm_vs_oracle(Cin, Cout, dims, up, use_res):
    """The decoder's GroupNorm -> SiLU -> conv chain as production runs it (vae.py:230-275, 246-263): the 4-wave conv emits the
    GroupNorm statistics of the outputs it stores (k5_conv3d_bf16_stats), the next norm normalises from them without a statistics
    pass (k5_groupnorm_bf16_quads).  Checked against torch.group_norm (+ SiLU) of the ORACLE's conv output (bf16 mode, residual
    added as the resnet does), and against the two-pass GroupNorm kernel on the very tensor the conv stored.  Shapes: both tile
    widths (Cout 128 / >= 256), folded upsampling, residual epilogue, M not a multiple of 256 / 128 (ragged last statistics block)."""
    G = 32
    E, (Ts, Hs, Ws, M), xd, wd, bd, rd, gamma, beta, want = _fused_stats_case(Cin, Cout, dims, up, use_res, Cin + Cout + dims[1])
    L = E.lib()
    out = torch.empty(M, Cout, dtype=torch.bfloat16, device="cuda")
    qs = torch.full((L.k5_conv3d_stats_size(M, Cout) // 4,), float("nan"), device="cuda")
    st = L.k5_conv3d_bf16_stats(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), Ts, Hs, Ws, Cin, Cout, up[0], up[1], Cout,
                                None if rd is None else rd.data_ptr(), Cout, qs.data_ptr(), E.stream_ptr())
    assert st == 0, (st, E.last_error())                     # really in the 4-wave kernel's range: the statistics exist
    torch.cuda.synchronize()
    assert torch.isfinite(qs).all()
    err = (out.float().cpu() - want).abs()
    tol = 2.0 ** -6 * want.abs().clamp(min=1.0) + (2.0 ** -7 * want.abs() if use_res else 0)
    assert not (err > tol).any(), (float(err.max()), int((err > tol).sum()))
    # the statistics are those of the STORED tensor: fold them on the host and compare with the tensor's own sums
    q = qs.view(-1, Cout // 4, 2).double().sum(0).cpu()                                       # [Cout / 4][sum, sum of squares]
    o64 = out.double().cpu().view(M, Cout // 4, 4)
    assert torch.allclose(q[:, 0], o64.sum((0, 2)), rtol=1e-5, atol=1e-2) and torch.allclose(q[:, 1], (o64 * o64).sum((0, 2)), rtol=1e-5)
    gd, btd = gamma.cuda(), beta.cuda()
    ws = torch.empty(L.k5_groupnorm_workspace_size(M, G), dtype=torch.uint8, device="cuda")
    for silu in (1, 0):
        y = torch.full((M, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
        E.check(L.k5_groupnorm_bf16_quads(out.data_ptr(), gd.data_ptr(), btd.data_ptr(), y.data_ptr(), M, Cout, G, 1e-6, silu, qs.data_ptr(),
                                          ws.data_ptr(), E.stream_ptr()), "k5_groupnorm_bf16_quads")
        y2 = torch.empty_like(y)
        E.check(L.k5_groupnorm_bf16(out.data_ptr(), gd.data_ptr(), btd.data_ptr(), y2.data_ptr(), M, Cout, G, 1e-6, silu, ws.data_ptr(),
                                    E.stream_ptr()), "k5_groupnorm_bf16")
        torch.cuda.synchronize()
        ref = torch.nn.functional.group_norm(want.t()[None], G, gamma, beta, 1e-6)[0].t()        # oracle conv -> GroupNorm
        if silu:
            ref = torch.nn.functional.silu(ref)
        e = (y.float().cpu() - ref).abs()
        # conv outputs differ from the oracle's by a bf16 ulp here and there (2^-8 relative), amplified by gamma / sigma ~ 1.2
        # plus one bf16 ulp of the conv output itself (2^-8 |x| each way) through the norm's gain gamma / sigma (<= 2 here)
        t = 2.0 ** -6 * ref.abs().clamp(min=1.0) + 2.0 ** -6 * want.abs()
        assert not (e > t).any(), (silu, float(e.max()), int((e > t).sum()))
        d = (y.float() - y2.float()).abs()
        assert d.max().item() <= 2.0 ** -7 * max(1.0, y2.float().abs().max().item()) and (d > 0).float().mean().item() < 2e-3, \
            (float(d.max()), float((d > 0).float().mean()))     # same tensor, statistics summed in another order


def test_conv_statistics_outside_the_four_wave_range():
    """Below 5/8 of a round of tiles (or Cin % 128 != 0) the conv has no statistics epilogue: K5_ERR_UNSUPPORTED and nothing launched."""
    from kandinsky import _engine as E
    L = E.lib()
    x = torch.zeros(2 * 8 * 8, 64, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(128, 27 * 64, dtype=torch.bfloat16, device="cuda")
    b = torch.zeros(128, device="cuda")
    out = torch.full((128, 128), 7.0, dtype=torch.bfloat16, device="cuda")
    qs = torch.zeros(L.k5_conv3d_stats_size(128, 128) // 4, device="cuda")
    st = L.k5_conv3d_bf16_stats(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), 2, 8, 8, 64, 128, 1, 1, 128, None, 128, qs.data_ptr(),
                                E.stream_ptr())
    torch.cuda.synchronize()
    assert st == 6 and (out == 7.0).all()


@pytest.fixture(scope="module")
def full_vae():
    """The production decoder (128 / 256 / 512 / 512 channels, 32 groups) with the weights the golden generator used
    (oracle/gen_golden_vae_fullwidth.py: synthetic_decoder_state_dict(full_manifest, 21))."""
    from kandinsky.models.vae import AutoencoderKLHunyuanVideo
    here = os.path.dirname(__file__)
    meta = json.load(open(os.path.join(here, "golden", "vae_meta.json")))
    fm = json.load(open(os.path.join(here, "golden", "vae_fullwidth_meta.json")))
    sd = V.synthetic_decoder_state_dict(meta["full_manifest"], fm["weights_seed"])
    m = AutoencoderKLHunyuanVideo()            # decode-only load: the encoder.* / quant_conv.* parameters stay placeholders
    m.load_state_dict(sd, assign=True)
    return m.to("cuda:0"), fm


def _path_counts(m, reset=True):
    import ctypes as C
    from kandinsky import _engine as E
    c = (C.c_longlong * 8)()
    E.check(E.lib().k5_vae_path_counts(m._engine(torch.device("cuda:0")), c, 1 if reset else 0), "k5_vae_path_counts")
    return list(c)


def test_production_tile_vs_reference_golden(full_vae):
    """The decode bench.py times is 14 of these: one (5, 64, 96) latent tile through the production-width decoder.  Golden: the
    REFERENCE decoder itself (vae.py:589-696, fp32, build container) on the same seeded weights and latent, 32768 sampled outputs +
    whole-tensor sums; plus the same samples from the oracle in bf16-autocast mode (the engine's arithmetic).  The engine must have
    taken the production kernels — 4-wave conv with the fused GroupNorm statistics, GroupNorm from those statistics, the C = 512
    one-kernel mid attention, conv_out3 — which its per-route launch counters prove."""
    from safetensors.torch import load_file
    m, fm = full_vae
    g = load_file(os.path.join(os.path.dirname(__file__), "golden", "vae_fullwidth.safetensors"))
    T, H, W = fm["tile"]
    z = torch.randn(1, 16, T, H, W, generator=torch.Generator().manual_seed(fm["latent_seed"]))
    _path_counts(m)
    out = m._decode_tile(z.cuda())
    torch.cuda.synchronize()
    cnt = _path_counts(m)
    print("VAE kernel routes [tile128, w4, w4+stats, out3, gn quads, gn own pass, attn512, attn gemm]:", cnt)
    assert list(out.shape) == fm["ref"]["out_shape"] == [1, 3, 4 * (T - 1) + 1, 8 * H, 8 * W]
    # conv_in (Cin = 16 -> padded 64) is the only conv outside the 4-wave kernel; 31 convs carry statistics; 2 norms (the first one
    # and the one after the attention) make their own statistics pass
    assert cnt[2] == 31 and cnt[0] == 1 and cnt[1] == 0 and cnt[3] == 1 and cnt[6] == 1 and cnt[7] == 0 and cnt[4] == 28 and cnt[5] == 2, cnt
    flat = out.float().reshape(-1)
    got = flat[g["sample_idx"].cuda()].cpu()
    ref, ob = g["ref.sample_val"], g["oraclebf16.sample_val"]
    r_ref, r_bf = ((got - ref).norm() / ref.norm()).item(), ((got - ob).norm() / ob.norm()).item()
    r_oo = ((ob - ref).norm() / ref.norm()).item()
    print(f"production tile: engine vs reference fp32 {r_ref:.3e}, vs bf16 oracle {r_bf:.3e}; bf16 oracle vs reference {r_oo:.3e}")
    assert r_bf <= 2e-2, r_bf
    assert r_ref <= 4e-2, (r_ref, r_oo)
    s, ss = flat.double().sum().item(), flat.double().pow(2).sum().item()
    assert abs(ss - fm["ref"]["out_sumsq"]) <= 4e-2 * fm["ref"]["out_sumsq"], (ss, fm["ref"]["out_sumsq"])
    assert abs(s - fm["ref"]["out_sum"]) <= 2e-2 * math.sqrt(fm["ref"]["out_sumsq"] * flat.numel()), (s, fm["ref"]["out_sum"])
    u8 = lambda t: ((t.clamp(-1.0, 1.0) + 1.0) * 127.5).to(torch.uint8).int()                 # generation_utils.py:222
    diff = (u8(got) - u8(ref.bfloat16().float())).abs()
    hist = [round((diff == i).float().mean().item(), 5) for i in range(int(diff.max()) + 1)]
    print("uint8 |engine - reference| on the samples:", hist)
    assert diff.float().mean().item() < 1.0 and (diff <= 3).float().mean().item() >= 0.99, hist
    again = m._decode_tile(z.cuda())
    assert torch.equal(out, again)                                                            # deterministic
