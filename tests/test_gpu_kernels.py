"""GPU parity tests (MI355X): every HIP kernel, called through the C ABI, against the CPU oracle
(oracle/k5_oracle.py, bf16-island arithmetic) on the same seeded inputs.  Tolerances are written here:
bf16 outputs agree to `ulps` bf16 ulps (one ulp <= 2^-7 * |x|: fp32 summation order differs between the
GPU and the CPU oracle, which can flip a round-to-nearest tie) plus a small absolute term."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import k5_oracle as O  # noqa: E402

BF = torch.bfloat16


@pytest.fixture(scope="module")
def E():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X (torch.cuda.is_available() is False)")
    from kandinsky import _engine as E
    E.lib()
    return E


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def bfr(x):
    return x.to(BF).float()


def assert_bf16_close(got, ref, ulps=2, atol=1e-3, what=""):
    got, ref = got.float().cpu(), ref.float()
    tol = atol + ulps * 2.0 ** -7 * ref.abs()
    bad = (got - ref).abs() > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} / {bad.numel()} off; max abs err {(got - ref).abs().max():.4g}"


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 1792), (200, 72, 136), (7, 128, 96), (1, 64, 48),
                                   (333, 64, 1792), (1000, 3584, 256), (768, 512, 1792), (600, 300, 192), (512, 256, 128),
                                   (1024, 1792, 7168), (4100, 2048, 192), (5952, 1792, 1792)])
def test_gemm_bias(E, M, N, K):
    a, w, b = bfr(rnd(M, K, seed=1)), bfr(rnd(N, K, seed=2, scale=0.05)), bfr(rnd(N, seed=3, scale=0.1))
    ref = bfr(a @ w.t() + b)
    got = E.gemm(a.cuda().to(BF), w.cuda().to(BF), b.cuda(), E.EPI_BIAS)
    torch.cuda.synchronize()
    assert_bf16_close(got, ref, what=f"gemm {M}x{N}x{K}")


def test_gemm_256_tile_kernel_is_race_free_and_deterministic(E):
    """The 256x256 two-group kernel orders its LDS-DMA traffic with counted vmcnt + barriers only: screen for races by
    repeating a multi-tile problem (odd and even K-tile counts) and demanding bit-identical results every time, equal to
    the 128x128 kernel's (same fp32 accumulation order along K)."""
    import os
    for (M, N, K) in ((4096, 2048, 1792), (3072, 3072, 448), (8192, 1024, 128)):   # >= 128 tiles of 256x256: the persistent kernel's range
        a, w = bfr(rnd(M, K, seed=11)).cuda().to(BF), bfr(rnd(N, K, seed=12, scale=0.05)).cuda().to(BF)
        b = bfr(rnd(N, seed=13)).cuda()
        first = E.gemm(a, w, b, E.EPI_BIAS).clone()
        for _ in range(25):
            assert torch.equal(E.gemm(a, w, b, E.EPI_BIAS), first)
        ref = bfr(a.float().cpu() @ w.float().cpu().t() + b.cpu())
        assert_bf16_close(first, ref, what=f"k8 gemm {M}x{N}x{K}")


@pytest.mark.parametrize("epi", ["bias", "gate"])
def test_gemm_tail_round_as_128_quadrants(E, epi):
    """More 256x256 tiles than CUs with a small remainder (264 = 256 + 8): the persistent kernel takes the whole round, the
    8 leftover tiles are computed as 32 quadrants by the 128x128 kernel; incl. a ragged edge (M, N not multiples of 256)."""
    M, N, K = 33 * 256 - 40, 8 * 256 - 24, 128
    a, w = bfr(rnd(M, K, seed=21)), bfr(rnd(N, K, seed=22, scale=0.05))
    b = bfr(rnd(N, seed=23, scale=0.1))
    if epi == "bias":
        ref = bfr(a @ w.t() + b)
        got = E.gemm(a.cuda().to(BF), w.cuda().to(BF), b.cuda(), E.EPI_BIAS)
    else:
        resid, gate = bfr(rnd(M, N, seed=24)), rnd(N, seed=25)
        ref = bfr(resid + gate * bfr(a @ w.t() + b))
        r = resid.cuda().to(BF)
        got = E.gemm(a.cuda().to(BF), w.cuda().to(BF), b.cuda(), E.EPI_GATE, resid=r, gate=gate.cuda(), out=r)
    assert_bf16_close(got, ref, ulps=3, what=f"tail-split gemm {epi}")


def test_gemm_four_wave_kernel_is_race_free_and_handles_edges(E):
    """>= 512 tiles of 256x256 and a non-GELU epilogue: the 4-wave kernel (one wave per SIMD, hand-ordered MFMA / ds_read /
    buffer_load-to-LDS stream, K-tile stream running on across output tiles).  Its LDS traffic is ordered by counted waits and
    barriers only: repeat and demand bit-identical results; ragged M and N edges rely on the buffer range check."""
    M, N, K = 24 * 256 - 40, 23 * 256 - 24, 384          # 552 tiles, 6 K-tiles, ragged in both directions
    a, w = bfr(rnd(M, K, seed=71)), bfr(rnd(N, K, seed=72, scale=0.05))
    b = bfr(rnd(N, seed=73, scale=0.1))
    ad, wd = a.cuda().to(BF), w.cuda().to(BF)
    first = E.gemm(ad, wd, b.cuda(), E.EPI_BIAS).clone()
    for _ in range(10):
        assert torch.equal(E.gemm(ad, wd, b.cuda(), E.EPI_BIAS), first)
    acc = a @ w.t()
    assert_bf16_close(first, bfr(acc + b), what="4-wave gemm bias")
    assert_bf16_close(E.gemm(ad, wd, None, E.EPI_BIAS), bfr(acc), what="4-wave gemm no bias")
    resid, gate = bfr(rnd(M, N, seed=74)), rnd(N, seed=75)
    r = resid.cuda().to(BF)
    got = E.gemm(ad, wd, b.cuda(), E.EPI_GATE, resid=r, gate=gate.cuda(), out=r)   # in place, as the engine calls it
    inner = bfr(acc + b)                                                            # a 1-ulp flip of the INNER bf16 rounding (fp32
    ref = bfr(resid + gate * inner)                                                 # summation order) is worth |gate| ulp(inner) outside
    err = (got.float().cpu() - ref).abs()
    tol = 1e-3 + 3 * 2.0 ** -7 * ref.abs() + gate.abs() * 2.0 ** -7 * inner.abs()
    assert not (err > tol).any(), f"4-wave gemm gate: {int((err > tol).sum())} off, max abs err {err.max():.4g}"
    assert (err > 3 * 2.0 ** -7 * ref.abs() + 1e-3).float().mean().item() < 1e-5   # ... and such flips are rare
    bm = bfr(rnd(M, seed=76))
    ld = (N + 7) // 8 * 8 + 8                                                     # padded leading dimension (the V^T layout)
    out = torch.zeros(M, ld, dtype=BF, device="cuda")
    E.gemm(ad, wd, bm.cuda(), E.EPI_BIAS_M, out=out)
    assert_bf16_close(out[:, :N], bfr(acc + bm[:, None]), what="4-wave gemm bias_m")
    assert torch.count_nonzero(out[:, N:]) == 0


@pytest.mark.parametrize("M,N,K", [(5952, 1792, 1792), (3328, 1792, 384), (3328 - 40, 2048 - 24, 256), (1792, 5952, 256), (700, 1000, 512), (4000, 2304, 512), (9000, 1792, 7168), (40000, 128, 256)])
def test_gemm_four_wave_token_tile_heights(E, M, N, K):
    """Round 5: the four-wave kernel with 256- / 192- / 128-row token tiles (8 / 6 / 4 MFMA row tiles per wave; the dispatch picks by
    tile quantisation: 8-GPU shards, BASELINE config 1).  Every height sums a K column in the same order as the 128 x 128 kernel: the
    results must be BIT-identical to it for every epilogue, repeatable (the LDS traffic is ordered by counted waits only), and right.
    The 128-row form runs THREE LDS stages whose index rotates at run time across output tiles (K-tiles per tile mod 3 = 1, 0, 1, 1, 2, 2, 1 in these
    cases; the 329-, 288- and 497-tile cases give a workgroup two output tiles, so the rotation carries over a tile boundary)."""
    a, w = bfr(rnd(M, K, seed=81)), bfr(rnd(N, K, seed=82, scale=0.05))
    ad, wd = a.cuda().to(BF), w.cuda().to(BF)
    acc = a @ w.t()
    b, bm = bfr(rnd(N, seed=83, scale=0.1)), bfr(rnd(M, seed=84))
    resid, gate = bfr(rnd(M, N, seed=85)), rnd(N, seed=86)
    ld = (N + 7) // 8 * 8 + 8

    def run(kernel, tile):
        out = {}
        out["bias"] = E.gemm(ad, wd, b.cuda(), E.EPI_BIAS, kernel=kernel, token_tile=tile)
        out["nobias"] = E.gemm(ad, wd, None, E.EPI_BIAS, kernel=kernel, token_tile=tile)
        out["gelu"] = E.gemm(ad, wd, None, E.EPI_GELU, kernel=kernel, token_tile=tile)
        r = resid.cuda().to(BF)
        out["gate"] = E.gemm(ad, wd, b.cuda(), E.EPI_GATE, resid=r, gate=gate.cuda(), out=r, kernel=kernel, token_tile=tile)
        r2 = resid.cuda().to(BF)      # the feed-forward's gated residual has NO bias (nn.py:352-361): its own instantiation of the epilogue
        out["gate_nobias"] = E.gemm(ad, wd, None, E.EPI_GATE, resid=r2, gate=gate.cuda(), out=r2, kernel=kernel, token_tile=tile)
        o = torch.zeros(M, ld, dtype=BF, device="cuda")
        E.gemm(ad, wd, bm.cuda(), E.EPI_BIAS_M, out=o, kernel=kernel, token_tile=tile)
        out["bias_m"] = o
        return out
    base = run(2, 0)                                                     # the 128 x 128 kernel
    assert_bf16_close(base["bias"], bfr(acc + b), what="128x128 bias")
    assert_bf16_close(base["gelu"], bfr(torch.nn.functional.gelu(bfr(acc))), what="128x128 gelu")
    assert_bf16_close(base["bias_m"][:, :N], bfr(acc + bm[:, None]), what="128x128 bias_m")
    inner = bfr(acc)
    ref = bfr(resid + gate * inner)
    err = (base["gate_nobias"].float().cpu() - ref).abs()
    assert not (err > 1e-3 + 3 * 2.0 ** -7 * ref.abs() + gate.abs() * 2.0 ** -7 * inner.abs()).any(), f"128x128 gate without bias: max abs err {err.max():.4g}"
    for tile in (256, 192, 128):
        for rep in range(3):
            got = run(4, tile)
            for k in base:
                assert torch.equal(got[k], base[k]), f"{M}x{N}x{K} token tile {tile} {k} (run {rep}): {int((got[k] != base[k]).sum())} values differ from the 128x128 kernel"
        assert torch.count_nonzero(got["bias_m"][:, N:]) == 0


@pytest.mark.parametrize("M,N,K,epi", [(47616, 1792, 1792, "gate"), (47616, 1792, 7168, "gate_nobias"), (20000, 3584, 1792, "bias"), (1792, 47616, 1792, "bias_m"),
                                       (33333, 2048 + 8, 3584, "gelu"), (17000, 4096, 256, "bias")])
def test_gemm_split_k_tail_vs_whole_tiles_and_fp32(E, M, N, K, epi):
    """Round 6: the ragged last round of a four-wave launch is cut along K into two aligned slices per tile on two workgroups of one XCD (kernel id 24 =
    wherever a tile can be cut; the default policy takes it where at most half of the CUs would be busy) — the helper hands its raw fp32 accumulators over
    through the XCD's L2, the owner adds them and runs the epilogue.  K-ORDER CONTRACT: a split tile's sum is (second half) + (first half) instead of one
    chain, so the last fp32 bit of some sums moves and a bf16 rounding flips here and there; everything else of the launch is the whole-tile schedule.
    Demanded: repeatable bit for bit (the flags are left clean), at most 1e-4 of the outputs differ from the whole-tile schedule (kernel id 4), and against
    an fp32 evaluation with the engine's rounding points the split schedule is no worse than the whole-tile one (max within 25 %, mean within 2 %)."""
    a, w = bfr(rnd(M, K, seed=91)), bfr(rnd(N, K, seed=92, scale=0.05))
    ad, wd = a.cuda().to(BF), w.cuda().to(BF)
    b = bfr(rnd(M if epi == "bias_m" else N, seed=93, scale=0.1)).cuda()
    resid, gate = bfr(rnd(M, N, seed=94)).cuda().to(BF), rnd(N, seed=95).cuda()
    code = {"bias": E.EPI_BIAS, "bias_m": E.EPI_BIAS_M, "gelu": E.EPI_GELU, "gate": E.EPI_GATE, "gate_nobias": E.EPI_GATE}[epi]

    def run(kernel):
        r = resid.clone() if code == E.EPI_GATE else None
        return E.gemm(ad, wd, None if epi in ("gelu", "gate_nobias") else b, code, resid=r, gate=gate if code == E.EPI_GATE else None, kernel=kernel)
    whole, split = run(4), run(24)
    for _ in range(3):
        assert torch.equal(run(24), split), "the split-K tail is not repeatable"
    frac = (whole != split).float().mean().item()
    rows = torch.randperm(M, generator=torch.Generator().manual_seed(1))[:1024].cuda()
    y = ad[rows].float() @ wd.float().t()
    if epi in ("bias", "gate"):
        y = y + b[None, :]
    if epi == "bias_m":
        y = y + b[rows][:, None]
    y = y.to(BF).float()
    if epi == "gelu":
        y = torch.nn.functional.gelu(y)
    if code == E.EPI_GATE:
        y = resid[rows].float() + gate[None, :] * y
    e_w, e_s = (whole[rows].float() - y).abs(), (split[rows].float() - y).abs()
    print(f"{M}x{N}x{K} {epi}: outputs that differ from the whole-tile schedule {frac:.2e}; |error| vs fp32 max {e_w.max():.3e} / {e_s.max():.3e}, mean {e_w.mean():.3e} / {e_s.mean():.3e}")
    assert frac < 1e-4
    assert e_s.max().item() <= 1.25 * e_w.max().item() + 1e-6 and e_s.mean().item() <= 1.02 * e_w.mean().item() + 1e-9


def test_gemm_is_transpose_correct(E):
    """A = I with an asymmetric W catches any row/col swap in the MFMA C/D mapping (guide rule 16)."""
    n = 256
    a = torch.eye(n)
    w = bfr(rnd(192, n, seed=5))
    got = E.gemm(a.cuda().to(BF), w.cuda().to(BF), None, E.EPI_BIAS)
    assert torch.equal(got.float().cpu(), w.t().contiguous())


def test_gemm_bias_m_emits_transposed_value(E):
    """V^T = W_v X^T + b_v[:,None]  (the layout the attention kernel consumes)."""
    D, S = 128, 300
    x, wv, bv = bfr(rnd(S, D, seed=1)), bfr(rnd(D, D, seed=2, scale=0.1)), bfr(rnd(D, seed=3))
    ref = bfr(x @ wv.t() + bv).t()
    ld = (S + 7) // 8 * 8
    out = torch.zeros(D, ld, dtype=BF, device="cuda")
    E.gemm(wv.cuda().to(BF), x.cuda().to(BF), bv.cuda(), E.EPI_BIAS_M, out=out)
    assert_bf16_close(out[:, :S], ref, what="V^T gemm")
    assert torch.count_nonzero(out[:, S:]) == 0


def test_gemm_gelu_and_gate_epilogues(E):
    M, D, FF = 300, 128, 256
    x, w1, w2 = bfr(rnd(M, D, seed=1)), bfr(rnd(FF, D, seed=2, scale=0.1)), bfr(rnd(D, FF, seed=3, scale=0.1))
    h_ref = bfr(torch.nn.functional.gelu(bfr(x @ w1.t())))
    h = E.gemm(x.cuda().to(BF), w1.cuda().to(BF), None, E.EPI_GELU)
    assert_bf16_close(h, h_ref, what="gelu epilogue")
    resid, gate = bfr(rnd(M, D, seed=4)), rnd(D, seed=5)
    ref = bfr(resid + gate * bfr(h_ref @ w2.t()))
    r = resid.cuda().to(BF)
    got = E.gemm(h_ref.cuda().to(BF), w2.cuda().to(BF), None, E.EPI_GATE, resid=r, gate=gate.cuda(), out=r)  # in place
    assert_bf16_close(got, ref, ulps=3, what="gate epilogue (in place)")


def test_gemm_gelu_epilogue_persistent_kernel(E):
    """GELU epilogue at a size the 256x256 persistent kernel takes (>= 128 tiles), ragged in M and N."""
    M, D, FF = 4000, 256, 2040
    x, w1 = bfr(rnd(M, D, seed=51)), bfr(rnd(FF, D, seed=52, scale=0.1))
    ref = bfr(torch.nn.functional.gelu(bfr(x @ w1.t())))
    got = E.gemm(x.cuda().to(BF), w1.cuda().to(BF), None, E.EPI_GELU)
    assert_bf16_close(got, ref, what="gelu epilogue (k8)")


@pytest.mark.parametrize("S,K,hw", [(6144, 512, 2048), (5000, 512, 1000), (4608, 512, 1536), (700, 192, 100), (6144, 512, 0)])
def test_gemm_f32out_frame_causal_scores(E, S, K, hw):
    """k5_gemm_bf16_f32out (fp32 attention scores of the VAE mid block, vae.py:341-362 + causal mask vae.py:110-122): every
    entry a causal softmax of frame size hw reads — column < (row // hw + 1) * hw — against an fp32 matmul of the same bf16
    operands; 256x256 (128x128 on the small shapes) tiles wholly beyond the limit must have been skipped: the NaN the buffer was
    filled with is still there.  (6144, 2048) and (5000, 1000) take the 4-wave persistent kernel (>= 256 kept tiles), the others
    the 128x128 one; hw = 1000 puts frame boundaries inside tiles; hw = 0 = no mask."""
    q = rnd(S, K, seed=1).cuda().to(BF)
    k = rnd(S, K, seed=2).cuda().to(BF)
    ld = (S + 7) // 8 * 8
    out = torch.full((S, ld), float("nan"), device="cuda")
    alpha = 1.0 / math.sqrt(K)
    E.check(E.lib().k5_gemm_bf16_f32out(q.data_ptr(), k.data_ptr(), out.data_ptr(), S, S, K, K, K, ld, alpha, hw, E.stream_ptr()),
            "k5_gemm_bf16_f32out")
    torch.cuda.synchronize()
    ref = (q.float() @ k.float().t()) * alpha
    rows = torch.arange(S, device="cuda")[:, None]
    cols = torch.arange(S, device="cuda")[None, :]
    need = (cols < ((rows // hw + 1) * hw if hw else S)).expand(S, S)
    got = out[:, :S]
    assert torch.isfinite(got[need]).all()
    assert (got[need] - ref[need]).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    if hw:
        tile = 256 if S >= 5000 else 128
        skipped = cols // tile * tile >= ((torch.clamp((rows // tile + 1) * tile, max=S) - 1) // hw + 1) * hw
        assert skipped.any() and torch.isnan(got[skipped]).all()      # those tiles were never computed


def test_gemm_alignment_error_is_loud(E):
    a, w = torch.zeros(8, 12, dtype=BF, device="cuda"), torch.zeros(8, 12, dtype=BF, device="cuda")
    with pytest.raises(RuntimeError, match="status 2"):
        E.gemm(a, w)


# ------------------------------------------------------------------------------------------ attention
def attn_ref(q, k, v):
    """oracle sdpa on bf16-valued inputs (fp32 softmax), (S,H,64) layout."""
    return O.sdpa(q, k, v, "bf16")


@pytest.mark.parametrize("Sq,Sk,H", [(72, 72, 2), (256, 256, 1), (300, 77, 3), (513, 7, 2), (1000, 1000, 2),
                                     (64, 640, 1), (1, 1, 1)])
def test_attention_matches_oracle(E, Sq, Sk, H):
    q, k, v = bfr(rnd(Sq, H, 64, seed=1)), bfr(rnd(Sk, H, 64, seed=2)), bfr(rnd(Sk, H, 64, seed=3))
    ref = attn_ref(q, k, v)
    ld = (Sk + 7) // 8 * 8
    vt = torch.zeros(H * 64, ld, dtype=BF, device="cuda")
    vt[:, :Sk] = v.reshape(Sk, H * 64).t().to(BF)
    got = E.attention(q.reshape(Sq, -1).cuda().to(BF), k.reshape(Sk, -1).cuda().to(BF), vt, H, kv_len=Sk)
    torch.cuda.synchronize()
    # P is rounded to bf16 before P.V (as in flash-attn): allow 1e-2 absolute on O(1) outputs
    assert_bf16_close(got, ref, ulps=4, atol=1e-2, what=f"attention {Sq}x{Sk}x{H}")


def test_attention_strided_qk_views(E):
    """Q and K are column slices of the fused [S][2D] projection buffer (ldq = ldk = 2D)."""
    S, H = 200, 2
    qk = bfr(rnd(S, 2 * H * 64, seed=7))
    v = bfr(rnd(S, H, 64, seed=8))
    ref = attn_ref(qk[:, :H * 64].reshape(S, H, 64), qk[:, H * 64:].reshape(S, H, 64), v)
    d = qk.cuda().to(BF)
    vt = v.reshape(S, -1).t().contiguous().cuda().to(BF)
    got = E.attention(d[:, :H * 64], d[:, H * 64:], vt, H)
    assert_bf16_close(got, ref, ulps=4, atol=1e-2, what="strided attention")


def test_attention_online_softmax_rescale_branch(E):
    """Force the running-max rescale late in the key sequence: one key with a huge score at tile 9
    (guide rule 26: the rare data-dependent branch needs its own test)."""
    S, H = 1024, 1
    q, k, v = bfr(rnd(S, H, 64, seed=1)), bfr(rnd(S, H, 64, seed=2)), bfr(rnd(S, H, 64, seed=3))
    k[600, 0] = q[5, 0] * 4.0  # spikes q5.k600 far above everything seen before
    k[900, 0] = q[700, 0] * 6.0
    ref = attn_ref(q, k, v)
    vt = v.reshape(S, -1).t().contiguous().cuda().to(BF)
    got = E.attention(q.reshape(S, -1).cuda().to(BF), k.reshape(S, -1).cuda().to(BF), vt, H)
    assert_bf16_close(got, ref, ulps=4, atol=1e-2, what="rescale branch")


def test_attention_full_size_properties(E):
    """BASELINE config-2 sequence length (N = 47 616 tokens) on 2 heads: size-independent properties
    (constant V -> constant O; key permutation invariance) + sampled rows against the oracle."""
    N, H = 47616, 2
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(N, H * 64, device="cuda", generator=g).to(BF)
    k = torch.randn(N, H * 64, device="cuda", generator=g).to(BF)
    v = torch.randn(N, H * 64, device="cuda", generator=g).to(BF)
    vt = v.t().contiguous()
    o = E.attention(q, k, vt, H)
    # (1) sampled query rows vs the CPU oracle
    rows = torch.tensor([0, 1, 31, 32, 255, 256, 4097, 23808, 47615 - 64, 47615])
    ref = attn_ref(q[rows].float().cpu().reshape(-1, H, 64), k.float().cpu().reshape(N, H, 64),
                   v.float().cpu().reshape(N, H, 64))
    assert_bf16_close(o[rows], ref, ulps=4, atol=5e-3, what="full-size sampled rows")
    # (2) softmax rows sum to one: V = const  ->  O = const
    vc = torch.full_like(vt, 0.75)
    oc = E.attention(q, k, vc, H)
    assert (oc.float() - 0.75).abs().max().item() <= 2 ** -8
    # (3) permuting the keys (and values) does not change the result beyond accumulation order
    perm = torch.randperm(N, device="cuda", generator=g)
    op = E.attention(q, k[perm].contiguous(), vt[:, perm].contiguous(), H)
    assert (op.float() - o.float()).abs().max().item() < 2e-2
    assert not torch.isnan(o.float()).any()


# ------------------------------------------------------------------------------------------ elementwise / norms
@pytest.mark.parametrize("rows,D", [(72, 128), (301, 1792), (5, 2048), (9, 64)])
def test_ln_modulate(E, rows, D):
    x = bfr(rnd(rows, D, seed=1, scale=2.0) + 0.3)
    sc, sh = rnd(1, D, seed=2, scale=0.3), rnd(1, D, seed=3, scale=0.3)
    ref = O.scale_shift_norm(x, sc, sh, "bf16")
    got = E.ln_modulate(x.cuda().to(BF), sc.cuda().reshape(-1), sh.cuda().reshape(-1))
    assert_bf16_close(got, ref, ulps=1, atol=1e-4, what="ln_modulate")


@pytest.mark.parametrize("rows,D", [(7, 512), (256, 1792), (1, 512), (129, 1792), (33, 192)])
def test_ln_affine_text_embedding_norm(E, rows, D):
    """k5_ln_affine_bf16 = TextEmbeddings.norm (nn.py:64-72: nn.LayerNorm with elementwise affine on the bf16 in_layer output, fp32 statistics under
    autocast, `.type_as(x)` back to bf16) — both outputs (bf16 rows for the text stream; the bf16-rounded fp32 copy the pooled embedding adds to
    the time embedding, dit.py:134) against torch's own LayerNorm on the same bf16-valued rows (VERDICT r4 missing #4: until round 5 this entry was
    only covered through whole-forward parity)."""
    x = bfr(rnd(rows, D, seed=1, scale=2.5) + 0.7)
    w, b = rnd(D, seed=2, scale=0.4) + 1.0, rnd(D, seed=3, scale=0.3)
    ref = bfr(torch.nn.functional.layer_norm(x, (D,), w, b, eps=1e-5))
    xd, wd, bd = x.cuda().to(BF), w.cuda(), b.cuda()        # (kept alive: the calls below take raw pointers)
    o16 = torch.empty(rows, D, dtype=BF, device="cuda")
    o32 = torch.empty(rows, D, dtype=torch.float32, device="cuda")
    E.check(E.lib().k5_ln_affine_bf16(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), o16.data_ptr(), o32.data_ptr(), rows, D, E.stream_ptr()), "k5_ln_affine_bf16")
    torch.cuda.synchronize()
    assert_bf16_close(o16, ref, ulps=1, atol=1e-4, what="ln_affine bf16 out")
    assert torch.equal(o32.cpu(), o16.float().cpu()), "the fp32 output is the bf16-rounded value"
    only16, only32 = torch.empty_like(o16), torch.empty_like(o32)
    E.check(E.lib().k5_ln_affine_bf16(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), only16.data_ptr(), None, rows, D, E.stream_ptr()), "k5_ln_affine_bf16")
    E.check(E.lib().k5_ln_affine_bf16(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, only32.data_ptr(), rows, D, E.stream_ptr()), "k5_ln_affine_bf16")
    torch.cuda.synchronize()
    assert torch.equal(only16, o16) and torch.equal(only32, o32)


def test_rmsnorm_rope_fused_qk(E):
    S, H = 150, 3
    qk = bfr(rnd(S, 2 * H * 64, seed=1, scale=3.0))
    wq, wk = rnd(64, seed=2) * 0.2 + 1, rnd(64, seed=3) * 0.2 + 1
    ang = rnd(S, 32, seed=4, scale=3.0)
    cos, sin = torch.cos(ang), torch.sin(ang)
    q = O.apply_rotary(O.rms_norm_heads(qk[:, :H * 64].reshape(S, H, 64), wq, "bf16"), cos, sin, "bf16")
    k = O.apply_rotary(O.rms_norm_heads(qk[:, H * 64:].reshape(S, H, 64), wk, "bf16"), cos, sin, "bf16")
    ref = torch.cat([q.reshape(S, -1), k.reshape(S, -1)], dim=1)
    d = qk.cuda().to(BF)
    E.rmsnorm_rope_(d, torch.cat([wq, wk]).cuda(), cos.cuda().contiguous(), sin.cuda().contiguous(), heads=2 * H,
                    heads_per_weight=H, rope_heads=2 * H)
    assert_bf16_close(d, ref, ulps=2, atol=1e-4, what="rmsnorm+rope")  # two chained bf16 roundings
    # no-rope variant (cross-attention q / k, reference nn.py:343-349)
    d2 = qk[:, :H * 64].contiguous().cuda().to(BF)
    E.rmsnorm_rope_(d2, wq.cuda())
    assert_bf16_close(d2, O.rms_norm_heads(qk[:, :H * 64].reshape(S, H, 64), wq, "bf16").reshape(S, -1), ulps=1,
                      atol=1e-4, what="rmsnorm only")


def test_gate_sum(E):
    x, y, g = bfr(rnd(77, 256, seed=1)), bfr(rnd(77, 256, seed=2)), rnd(256, seed=3)
    got = E.gate_sum(x.cuda().to(BF), y.cuda().to(BF), g.cuda())
    assert_bf16_close(got, O.gate_sum(x, y, g, "bf16"), ulps=1, atol=1e-5, what="gate_sum")


def test_gemv_f32_modulation(E, tiny_sd):
    temb = rnd(1, 64, seed=1)
    ref = O.modulation(tiny_sd, "visual_transformer_blocks.0.visual_modulation", temb)[0]
    got = E.gemv_f32(temb.cuda().reshape(-1), tiny_sd["visual_transformer_blocks.0.visual_modulation.out_layer.weight"].cuda(),
                     tiny_sd["visual_transformer_blocks.0.visual_modulation.out_layer.bias"].cuda(), silu_in=True)
    torch.testing.assert_close(got.cpu(), ref, atol=2e-5, rtol=2e-5)


def test_time_features_and_rope_tables(E):
    import ctypes as C
    out = torch.empty(1792, device="cuda")
    E.check(E.lib().k5_time_features_f32(731.25, out.data_ptr(), 1792, E.stream_ptr()))
    a = torch.outer(torch.tensor([731.25]), O.get_freqs(896))[0]
    ref = torch.cat([torch.cos(a), torch.sin(a)])
    torch.testing.assert_close(out.cpu(), ref, atol=5e-4, rtol=0)  # angles up to ~731 rad: fp32 argument ulp 6e-5
    T, H, W = 3, 4, 6
    pos = torch.cat([torch.arange(T), torch.arange(H), torch.arange(W)]).int().cuda()
    cos, sin = torch.empty(T * H * W, 32, device="cuda"), torch.empty(T * H * W, 32, device="cuda")
    E.check(E.lib().k5_rope_table_f32(cos.data_ptr(), sin.data_ptr(), pos.data_ptr(), pos[T:].data_ptr(),
                                      pos[T + H:].data_ptr(), T, H, W, 8, 12, 12, 1.0, 2.0, 2.0, None, E.stream_ptr()))
    args = O.rope_3d_args((T, H, W), [torch.arange(T), torch.arange(H), torch.arange(W)], (16, 24, 24), (1.0, 2.0, 2.0))
    torch.testing.assert_close(cos.cpu(), torch.cos(args).reshape(-1, 32), atol=2e-6, rtol=0)
    torch.testing.assert_close(sin.cpu(), torch.sin(args).reshape(-1, 32), atol=2e-6, rtol=0)


def test_patchify_unpatchify_orders(E):
    T, H, W, C = 2, 4, 6, 16
    x = rnd(T, H, W, C, seed=1)
    full = torch.cat([x, torch.zeros(T, H, W, C + 1)], dim=-1)
    ref = bfr(O.patchify(full, (1, 2, 2))).reshape(-1, 4 * 33)
    out = torch.empty(T * (H // 2) * (W // 2), 136, dtype=BF, device="cuda")
    xd = x.cuda()
    E.check(E.lib().k5_patchify_bf16(xd.data_ptr(), out.data_ptr(), T, H, W, C, 33, 136, None, E.stream_ptr()))
    assert torch.equal(out[:, :132].float().cpu(), ref) and torch.count_nonzero(out[:, 132:]) == 0
    y = bfr(rnd(T * 2 * 3, 64, seed=2))
    ref_u = O.unpatchify(y.reshape(T, 2, 3, 64), (1, 2, 2))
    yo = torch.empty(T, 4, 6, 16, dtype=BF, device="cuda")
    yd = y.cuda().to(BF)
    E.check(E.lib().k5_unpatchify_bf16(yd.data_ptr(), yo.data_ptr(), T, 2, 3, 16, 64, None, E.stream_ptr()))
    assert torch.equal(yo.float().cpu(), ref_u)


def test_cfg_euler(E):
    img, c, u = rnd(1000, seed=1), bfr(rnd(1000, seed=2)), bfr(rnd(1000, seed=3))
    w, dt = 5.0, -0.0625
    ref = img + bfr(dt * bfr(u + bfr(w * bfr(c - u))))
    d = img.cuda()
    E.cfg_euler_(d, c.cuda().to(BF), u.cuda().to(BF), w, dt)
    assert torch.equal(d.cpu(), ref)
    d = img.cuda()
    E.cfg_euler_(d, c.cuda().to(BF), None, 1.0, dt)
    assert torch.equal(d.cpu(), img + bfr(dt * c))


def test_attention_bounded_scores_equals_online_max(E):
    """RMS-normalised q,k (|q|=|k|=8 -> |q.k| <= 64): the fixed-offset kernel is the same softmax."""
    S, H = 700, 2
    def rmsn(x):
        return bfr(x / x.pow(2).mean(-1, keepdim=True).sqrt())
    q, k, v = rmsn(rnd(S, H, 64, seed=1)), rmsn(rnd(S, H, 64, seed=2)), bfr(rnd(S, H, 64, seed=3))
    ref = attn_ref(q, k, v)
    vt = torch.zeros(H * 64, 704, dtype=BF, device="cuda")  # leading dimension must be a multiple of 8
    vt[:, :S] = v.reshape(S, -1).t().to(BF)
    qd, kd = q.reshape(S, -1).cuda().to(BF), k.reshape(S, -1).cuda().to(BF)
    with pytest.raises(RuntimeError, match="status 2"):  # unaligned V^T rows are refused loudly
        E.attention(qd, kd, v.reshape(S, -1).t().contiguous().cuda().to(BF), H, score_bound=64 * 1.05)
    a = E.attention(qd, kd, vt, H, kv_len=S, score_bound=64 * 1.05)
    b = E.attention(qd, kd, vt, H, kv_len=S)
    assert_bf16_close(a, ref, ulps=4, atol=1e-2, what="bounded attention")
    assert (a.float() - b.float()).abs().max().item() <= 2 ** -7
    # a bound too large for a fixed offset silently falls back to the online-max kernel (still correct)
    c = E.attention(qd, kd, vt, H, kv_len=S, score_bound=1e4)
    assert torch.equal(c, b)


@pytest.mark.parametrize("Sq,Sk,split,bound", [(300, 640, 3, None), (300, 640, 3, 64 * 1.05), (257, 1000, 9, None), (64, 200, 1, 64 * 1.05)])
def test_attention_split_key_passes_equal_single_pass(E, Sq, Sk, split, bound):
    """Sequence-parallel overlap building block: pass 1 over key tiles [split, split+n) leaving the fp32 state, pass 2 over
    all the other tiles resuming it == one pass over all keys (up to fp32 summation order)."""
    H = 2
    def rmsn(x):
        return bfr(x / x.pow(2).mean(-1, keepdim=True).sqrt())
    q, k, v = rmsn(rnd(Sq, H, 64, seed=1)), rmsn(rnd(Sk, H, 64, seed=2)), bfr(rnd(Sk, H, 64, seed=3))
    ld = (Sk + 7) // 8 * 8
    vt = torch.zeros(H * 64, ld, dtype=BF, device="cuda")
    vt[:, :Sk] = v.reshape(Sk, H * 64).t().to(BF)
    qd, kd = q.reshape(Sq, -1).cuda().to(BF), k.reshape(Sk, -1).cuda().to(BF)
    one = E.attention(qd, kd, vt, H, kv_len=Sk, score_bound=bound)
    total = (Sk + 63) // 64
    n1 = max(1, total // 3)
    state = torch.zeros(E.lib().k5_attention_state_size(H, Sq), dtype=torch.uint8, device="cuda")
    out = torch.full((Sq, H * 64), float("nan"), dtype=BF, device="cuda")
    sb = 0.0 if bound is None else bound
    args = (qd.data_ptr(), kd.data_ptr(), vt.data_ptr(), out.data_ptr(), H, Sq, Sk, qd.stride(0), kd.stride(0), vt.stride(0),
            out.stride(0), sb)
    E.check(E.lib().k5_attention_bf16_range(*args, split, n1, 0x7fffffff, 0, state.data_ptr(), 2, E.stream_ptr()))
    assert torch.isnan(out.float()).all()                       # pass 1 writes only the state
    E.check(E.lib().k5_attention_bf16_range(*args, 0, total - n1, split, n1, state.data_ptr(), 1, E.stream_ptr()))
    assert_bf16_close(out, one.float().cpu(), ulps=2, atol=2e-3, what="two-pass attention")
    assert_bf16_close(out, attn_ref(q, k, v), ulps=4, atol=1e-2, what="two-pass attention vs oracle")


@pytest.mark.parametrize("Sq,Sk,H,bound", [(47616 // 8, 47616, 28, 64 * 1.05), (5952, 8192, 28, None), (2600, 1500, 9, None),
                                           (300, 640, 2, 64 * 1.05)])
def test_attention_balanced_tail_split_equals_single_launch(E, Sq, Sk, H, bound):
    """Jobs of the last, partially filled round are split along the keys and merged: same result as one launch up to
    fp32 summation order (an 8-GPU shard's shape: 672 jobs on 512 slots -> 160 jobs split 3 ways)."""
    g = torch.Generator(device="cuda").manual_seed(5)
    def rmsn(x):
        return (x / x.pow(2).mean(-1, keepdim=True).sqrt())
    q = rmsn(torch.randn(Sq, H, 64, device="cuda", generator=g)).reshape(Sq, -1).to(BF)
    k = rmsn(torch.randn(Sk, H, 64, device="cuda", generator=g)).reshape(Sk, -1).to(BF)
    ld = (Sk + 7) // 8 * 8
    vt = torch.zeros(H * 64, ld, dtype=BF, device="cuda")
    vt[:, :Sk] = torch.randn(H * 64, Sk, device="cuda", generator=g).to(BF)
    one = E.attention(q, k, vt, H, kv_len=Sk, score_bound=bound)
    L = E.lib()
    ws = torch.empty(L.k5_attention_balance_size(H, Sq), dtype=torch.uint8, device="cuda")
    out = torch.full((Sq, H * 64), float("nan"), dtype=BF, device="cuda")
    E.check(L.k5_attention_bf16_balanced(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), H, Sq, Sk, q.stride(0),
                                         k.stride(0), vt.stride(0), out.stride(0), 0.0 if bound is None else bound,
                                         ws.data_ptr(), E.stream_ptr()))
    assert not torch.isnan(out.float()).any()
    assert_bf16_close(out, one.float().cpu(), ulps=2, atol=2e-3, what="balanced attention")


# ------------------------------------------------------------------------------------------ fp8 (e4m3) GEMM, opt-in path
F8 = torch.float8_e4m3fn


def _to_f8_bytes(x):
    return x.to(F8).view(torch.uint8)


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (512, 768, 1792), (600, 300, 384), (1024, 7168, 1792),
                                   (4096, 4096, 512), (4100, 4352, 1792), (9000, 1792, 7168)])   # the last three: >= 256 tiles -> the four-wave kernel (round 4)
def test_gemm_fp8_matches_fp8_reference(E, M, N, K):
    """W8A8 e4m3 GEMM on v_mfma_scale_f32_16x16x128_f8f6f4: operands already in fp8 -> the only difference to an fp32
    matmul of the de-quantised operands is the accumulation order."""
    a8 = _to_f8_bytes(rnd(M, K, seed=31) * 2.0)
    w8 = _to_f8_bytes(rnd(N, K, seed=32))
    ws = (rnd(N, seed=33).abs() * 0.01 + 0.001)
    ref = bfr((a8.view(F8).float() @ w8.view(F8).float().t()) * ws)
    out = torch.empty(M, N, dtype=BF, device="cuda")
    L = E.lib()
    ad, wd, sd = a8.cuda(), w8.cuda(), ws.cuda()
    E.check(L.k5_gemm_fp8(ad.data_ptr(), wd.data_ptr(), sd.data_ptr(), out.data_ptr(), M, N, K, K, K, N, E.EPI_BIAS, None, 0, None, E.stream_ptr()))
    assert_bf16_close(out, ref, ulps=2, atol=1e-3, what=f"fp8 gemm {M}x{N}x{K}")
    for _ in range(10):   # race screen of the DMA / barrier schedule
        o2 = torch.empty_like(out)
        E.check(L.k5_gemm_fp8(ad.data_ptr(), wd.data_ptr(), sd.data_ptr(), o2.data_ptr(), M, N, K, K, K, N, E.EPI_BIAS, None, 0, None, E.stream_ptr()))
        assert torch.equal(o2, out)


@pytest.mark.parametrize("M,D,FF", [(512, 256, 512), (9472, 1792, 7168)])   # the second: 1036 and 259 tiles -> both GEMMs on the four-wave kernel (round 4)
def test_gemm_fp8_epilogues_and_row_quantisation(E, M, D, FF):
    L = E.lib()
    x = bfr(rnd(M, D, seed=41))
    w1, w2 = bfr(rnd(FF, D, seed=42, scale=0.05)), bfr(rnd(D, FF, seed=43, scale=0.05))
    # per-channel weight quantisation and static-scale activation quantisation done by the library
    def quant(t, per_row):
        src = t.cuda().to(BF)
        out = torch.empty(t.shape, dtype=torch.uint8, device="cuda")
        sc = torch.empty(t.shape[0], dtype=torch.float32, device="cuda") if per_row else None
        E.check(L.k5_quant_rows_fp8(src.data_ptr(), out.data_ptr(), sc.data_ptr() if per_row else None, t.shape[0], t.shape[1],
                                    t.shape[1], t.shape[1], E.stream_ptr()))
        return out, sc
    x8, _ = quant(x, False)
    w18, s1 = quant(w1, True)
    w28, s2 = quant(w2, True)
    # reference of the quantisers themselves
    assert torch.equal(x8.cpu(), _to_f8_bytes(x.clamp(-448, 448)))
    s1_ref = w1.abs().amax(1) / 448
    assert torch.allclose(s1.cpu(), s1_ref, rtol=1e-6)
    assert torch.equal(w18.cpu(), _to_f8_bytes(w1 / s1.cpu()[:, None]))
    # FF1: fp8 out = e4m3(GELU(bf16(acc * s)))
    h8 = torch.empty(M, FF, dtype=torch.uint8, device="cuda")
    E.check(L.k5_gemm_fp8(x8.data_ptr(), w18.data_ptr(), s1.data_ptr(), h8.data_ptr(), M, FF, D, D, D, FF, E.EPI_GELU, None, 0, None, E.stream_ptr()))
    pre = bfr((x8.cpu().view(F8).float() @ w18.cpu().view(F8).float().t()) * s1.cpu())
    h_ref = torch.nn.functional.gelu(pre)
    got = h8.cpu().view(F8).float()
    assert ((got - h_ref.to(F8).float()).abs() <= 0.13 * h_ref.abs() + 2e-3).all()       # at most one e4m3 step (bf16 tie flips)
    # FF2: gated residual, in place
    resid, gate = bfr(rnd(M, D, seed=44)), rnd(D, seed=45)
    r = resid.cuda().to(BF)
    E.check(L.k5_gemm_fp8(h8.data_ptr(), w28.data_ptr(), s2.data_ptr(), r.data_ptr(), M, D, FF, FF, FF, D, E.EPI_GATE, r.data_ptr(), D,
                          gate.cuda().data_ptr(), E.stream_ptr()))
    inner = bfr((got @ w28.cpu().view(F8).float().t()) * s2.cpu())
    ref = bfr(resid + gate * inner)
    # bf16 tie flips of the inner rounding (fp32 summation order over K differs from the host matmul's), times the gate: one bf16 ulp of the
    # INNER value scaled by |gate| — at K = 7168 the inner values reach ~30, their ulp 0.25 (round 4: the large shape)
    err = (r.float().cpu() - ref).abs()
    tol = 4e-3 + 4 * 2.0 ** -7 * ref.abs() + gate.abs()[None, :] * 2.0 ** -7 * inner.abs()
    assert (err <= tol).all(), f"fp8 ff2 gate: {int((err > tol).sum())} / {err.numel()} off; max abs err {err.max():.4g}"


def test_attention_prescaled_keys(E):
    """k5_attention_bf16_prescaled: keys pre-multiplied by log2(e)/8 and rounded once.  Exactly the softmax (base 2) of those
    keys — checked against the oracle fed the same rounded keys — and within bf16 noise of the unscaled formulation."""
    H, Sq, Sk = 3, 700, 1088          # whole 64-key tiles: the pre-scaled entry has no ragged-tile path
    def rmsn(x):
        return bfr(x / x.pow(2).mean(-1, keepdim=True).sqrt())
    q, k, v = rmsn(rnd(Sq, H, 64, seed=61)), rmsn(rnd(Sk, H, 64, seed=62)), bfr(rnd(Sk, H, 64, seed=63))
    kc = bfr(k * torch.tensor(O.SOFTMAX_C, dtype=torch.float32))
    ld = (Sk + 7) // 8 * 8
    vt = torch.zeros(H * 64, ld, dtype=BF, device="cuda")
    vt[:, :Sk] = v.reshape(Sk, H * 64).t().to(BF)
    qd, kd, kcd = q.reshape(Sq, -1).cuda().to(BF), k.reshape(Sk, -1).cuda().to(BF), kc.reshape(Sk, -1).cuda().to(BF)
    out = torch.empty(Sq, H * 64, dtype=BF, device="cuda")
    L = E.lib()
    E.check(L.k5_attention_bf16_prescaled(qd.data_ptr(), kcd.data_ptr(), vt.data_ptr(), out.data_ptr(), H, Sq, Sk, qd.stride(0),
                                          kcd.stride(0), vt.stride(0), out.stride(0), 64 * 1.05, E.stream_ptr()))
    assert_bf16_close(out, O.sdpa(q, kc, v, "bf16", None, base2=True), ulps=4, atol=1e-2, what="prescaled attention vs base-2 oracle")
    plain = E.attention(qd, kd, vt, H, kv_len=Sk, score_bound=64 * 1.05)
    assert (out.float() - plain.float()).abs().max().item() <= 3e-2        # re-rounded keys: bf16-level differences only
    # without a usable bound the same entry runs the online-max form (tests/test_gpu_softmax_variants.py): same softmax
    out2 = torch.empty_like(out)
    E.check(L.k5_attention_bf16_prescaled(qd.data_ptr(), kcd.data_ptr(), vt.data_ptr(), out2.data_ptr(), H, Sq, Sk, qd.stride(0),
                                          kcd.stride(0), vt.stride(0), out2.stride(0), 0.0, E.stream_ptr()))
    assert (out2.float() - out.float()).abs().max().item() <= 2 ** -6
    with pytest.raises(RuntimeError):                                       # ... and whole key tiles
        E.check(L.k5_attention_bf16_prescaled(qd.data_ptr(), kcd.data_ptr(), vt.data_ptr(), out.data_ptr(), H, Sq, Sk - 8, qd.stride(0),
                                              kcd.stride(0), vt.stride(0), out.stride(0), 64 * 1.05, E.stream_ptr()))


def test_attention_64_row_waves_give_the_same_bits():
    """K5_ATTN_WAVE_ROWS=64 (round 4, opt-in): the dense fixed-offset launch in 64-row waves — four waves per 256-query workgroup, every K / V^T
    fragment read feeding four MFMAs instead of two — is the same arithmetic per query row in the same order: its output must equal the
    default launch's BIT FOR BIT (ragged query count, balanced tail split, per-row offsets).  The switch is read once per process, hence the
    two child processes."""
    import hashlib, os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import hashlib, sys, torch
        sys.path.insert(0, %r)
        from kandinsky import _engine as E
        H, N = 5, 9 * 256 + 192
        g = torch.Generator().manual_seed(23)
        def rms(x): return x / x.pow(2).mean(-1, keepdim=True).sqrt()
        q = (rms(torch.randn(N, H, 64, generator=g)) * 2.5).reshape(N, H * 64).bfloat16().cuda()
        k = (rms(torch.randn(N, H, 64, generator=g)) * 2.5 * 0.18033688011112042).reshape(N, H * 64).bfloat16().cuda()
        vt = torch.randn(H * 64, N, generator=g).bfloat16().cuda()
        o = torch.empty(N, H * 64, dtype=torch.bfloat16, device="cuda")
        E.check(E.lib().k5_attention_bf16_prescaled(q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(), H, N, N, q.stride(0), k.stride(0),
                                                    vt.stride(0), o.stride(0), 64 * 2.5 * 2.5 * 1.01, E.stream_ptr()))
        torch.cuda.synchronize()
        assert torch.isfinite(o.float()).all()
        print("SHA", hashlib.sha256(o.cpu().view(torch.int16).numpy().tobytes()).hexdigest())
    """ % os.path.join(root, "kandinsky-5_amd"))
    shas = []
    for rows in ("32", "64", "164", "132"):      # 164 (round 5): 64-row waves, one per SIMD, software-pipelined over half tiles
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, K5_ATTN_WAVE_ROWS=rows), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        shas.append([l for l in out.stdout.splitlines() if l.startswith("SHA")][-1])
    assert shas[0] == shas[1] == shas[2] == shas[3], shas
