"""GPU parity of the two softmax forms of the visual self-attention (pre-scaled keys) and of the device-side, data-derived
choice between them — VERDICT r1 item 1: "make the fast softmax path survive real weights, and prove it".

  fixed offset  : exp2(q.k') with the constant offset 0 — valid when |q|max |k'|max <= 90 for the head (Cauchy-Schwarz)
  online max    : the lazy running offset (attn_fwd_kernel<BOUNDED=false>), any magnitudes
Both against oracle.sdpa (fp32 softmax) on the same bf16 inputs; the flags against a host recomputation of the bound."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import k5_oracle as O  # noqa: E402

BF = torch.bfloat16
C = torch.tensor(O.SOFTMAX_C, dtype=torch.float32)


@pytest.fixture(scope="module")
def E():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X")
    from kandinsky import _engine as E
    E.lib()
    return E


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def bfr(x):
    return x.to(BF).float()


def close(got, ref, ulps=4, atol=1e-2, what=""):
    got, ref = got.float().cpu(), ref.float()
    bad = (got - ref).abs() > atol + ulps * 2.0 ** -7 * ref.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())} / {bad.numel()} off; max abs err {(got - ref).abs().max():.4g}"


def vt_of(v):
    Sk, H = v.shape[0], v.shape[1]
    return v.reshape(Sk, H * 64).t().contiguous().cuda().to(BF)


def run_auto(E, q, kc, vt, H, flags=None, variant=0, balanced=False):
    Sq, Sk = q.shape[0], kc.shape[0]
    out = torch.full((Sq, H * 64), float("nan"), dtype=BF, device="cuda")
    L = E.lib()
    ws = torch.empty(L.k5_attention_balance_size(H, Sq), dtype=torch.uint8, device="cuda") if balanced else None
    E.check(L.k5_attention_bf16_prescaled_auto(q.data_ptr(), kc.data_ptr(), vt.data_ptr(), out.data_ptr(), H, Sq, Sk, q.stride(0),
                                               kc.stride(0), vt.stride(0), out.stride(0), None if flags is None else flags.data_ptr(),
                                               variant, None if ws is None else ws.data_ptr(), E.stream_ptr()),
            "k5_attention_bf16_prescaled_auto")
    torch.cuda.synchronize()
    return out


# ------------------------------------------------------------------------------------------ statistics of the norm kernel
@pytest.mark.parametrize("rows,H", [(300, 4), (1000, 56), (64, 2)])
def test_rmsnorm_rope_statistics_are_the_max_row_norms(E, rows, H):
    """k5_rmsnorm_rope_stats_bf16: stats[h] = max over rows of |x_h|^2 of the bf16 values it wrote (scaled heads included),
    accumulated on top of what the buffer held."""
    x = bfr(rnd(rows, H * 64, seed=1, scale=3.0))
    w = rnd(64, seed=2).abs() + 0.5
    cos, sin = torch.cos(rnd(rows, 32, seed=3)), torch.sin(rnd(rows, 32, seed=3))
    xd = x.cuda().to(BF)
    wd, cd, sd = w.cuda(), cos.cuda(), sin.cuda()      # named: a temporary's memory may be reused before the kernel runs
    stats = torch.zeros(H, device="cuda")
    stats[0] = 1e9                                     # an existing larger value must survive (max, not overwrite)
    E.check(E.lib().k5_rmsnorm_rope_stats_bf16(xd.data_ptr(), wd.data_ptr(), cd.data_ptr(), sd.data_ptr(), rows, H,
                                               xd.stride(0), H, H, float(O.SOFTMAX_C), H // 2, stats.data_ptr(), E.stream_ptr()))
    torch.cuda.synchronize()
    y = xd.float().reshape(rows, H, 64)
    n2 = (y * y).sum(-1).amax(0)
    n2[0] = 1e9
    assert torch.allclose(stats, n2, rtol=1e-5), (stats, n2)
    # the op itself: same values as the stats-less entry (oracle parity of that entry: test_gpu_kernels.py), scaled heads scaled
    x2 = x.cuda().to(BF)
    E.rmsnorm_rope_(x2, wd, cd, sd, heads=H)
    ref = x2.float().reshape(rows, H, 64).clone()
    assert torch.equal(xd.float().reshape(rows, H, 64)[:, :H // 2], ref[:, :H // 2])
    assert (xd.float().reshape(rows, H, 64)[:, H // 2:] - bfr(ref[:, H // 2:].cpu() * C).cuda()).abs().max() <= 2 ** -7 * ref.abs().max() * O.SOFTMAX_C


# ------------------------------------------------------------------------------------------ online-max form
@pytest.mark.parametrize("Sq,Sk,H,gain", [(700, 1088, 3, 1.0), (300, 640, 2, 6.0), (257, 2048, 2, 12.0), (64, 64, 1, 3.0)])
def test_prescaled_online_max_matches_oracle(E, Sq, Sk, H, gain):
    """variant 1 (online max everywhere) at growing score magnitudes: gain 12 puts |q.k'| up to ~ 12^2 * 64 * 0.18 = 1650 in
    the exp2 domain — far outside any fixed-offset window."""
    def rmsn(x):
        return bfr(gain * x / x.pow(2).mean(-1, keepdim=True).sqrt())
    q, k, v = rmsn(rnd(Sq, H, 64, seed=61)), rmsn(rnd(Sk, H, 64, seed=62)), bfr(rnd(Sk, H, 64, seed=63))
    kc = bfr(k * C)
    out = run_auto(E, q.reshape(Sq, -1).cuda().to(BF), kc.reshape(Sk, -1).cuda().to(BF), vt_of(v), H, None, 1)
    close(out, O.sdpa(q, kc, v, "bf16", None, base2=True), what=f"online max, gain {gain}")


def test_prescaled_online_max_rescale_branch(E):
    """Guide rule 26: force the rescale late in the key sequence (spikes at tiles 9 and 14, one of them > 2^60 above the
    running offset), a first tile whose scores are all hugely NEGATIVE for some rows (the first tile must SET the offset,
    not clamp it at 0), and a row whose late keys are far below the offset (they must flush to 0, not disturb)."""
    S, H = 1024, 2
    q, k, v = bfr(rnd(S, H, 64, seed=1)), bfr(rnd(S, H, 64, seed=2)), bfr(rnd(S, H, 64, seed=3))
    k[600, 0] = q[5, 0] * 40.0        # q5.k600 ~ 40 |q5|^2 ~ 2500 raw -> ~450 in the exp2 domain
    k[900, 1] = q[700, 1] * 6.0
    k[:64, 0] = -q[33, 0] * 30.0      # row 33, head 0: first tile ~ -350 (exp2 domain): offset must start there
    ref = O.sdpa(q, bfr(k * C), v, "bf16", None, base2=True)
    out = run_auto(E, q.reshape(S, -1).cuda().to(BF), bfr(k * C).reshape(S, -1).cuda().to(BF), vt_of(v), H, None, 1)
    close(out, ref, what="online rescale branch")
    assert not torch.isnan(out.float()).any()


def test_prescaled_online_two_pass_and_balanced_merge(E):
    """The online form through the resumable state (sequence-parallel two-pass schedule) and through the balanced launch
    (tail jobs split along the keys, merged with exp2(m_s - max m) weights): same result as one plain launch."""
    H, Sq, Sk = 28, 47616 // 8, 8192
    g = torch.Generator(device="cuda").manual_seed(5)
    def rmsn(x, gain):
        return gain * x / x.pow(2).mean(-1, keepdim=True).sqrt()
    q = rmsn(torch.randn(Sq, H, 64, device="cuda", generator=g), 4.0).reshape(Sq, -1).to(BF)
    kc = (rmsn(torch.randn(Sk, H, 64, device="cuda", generator=g), 4.0) * O.SOFTMAX_C).reshape(Sk, -1).to(BF)
    vt = torch.randn(H * 64, Sk, device="cuda", generator=g).to(BF)
    one = run_auto(E, q, kc, vt, H, None, 1)
    bal = run_auto(E, q, kc, vt, H, None, 1, balanced=True)
    # split parts exponentiate against their own offsets: the bf16 roundings of P differ from the single launch's
    close(bal, one.cpu(), ulps=4, atol=1e-2, what="balanced online")
    rows = torch.tensor([0, 77, 255, 256, 3000, Sq - 1])
    ref = O.sdpa(q[rows].float().cpu().reshape(-1, H, 64), kc.float().cpu().reshape(Sk, H, 64), vt.t().float().cpu().reshape(Sk, H, 64),
                 "bf16", None, base2=True)
    close(one[rows], ref, what="online sampled rows")


# ------------------------------------------------------------------------------------------ per-head choice on the device
def test_per_head_flags_from_the_data_and_mixed_launch(E):
    """Heads 0 and 2 small (|q||k'| ~ 12: fixed offset), head 1 large (~ 420: online max), head 3 just above the window.
    k5_attention_flags must say exactly that (recomputed here from the tensors), reset the statistics, and the mixed launch
    must match the oracle on every head."""
    H, S = 4, 1152
    gains = torch.tensor([1.0, 6.0, 1.0, 2.9])
    def rmsn(x):
        return bfr(gains[None, :, None] * x / x.pow(2).mean(-1, keepdim=True).sqrt())
    q, k, v = rmsn(rnd(S, H, 64, seed=11)), rmsn(rnd(S, H, 64, seed=12)), bfr(rnd(S, H, 64, seed=13))
    kc = bfr(k * C)
    qstat = (q * q).sum(-1).amax(0).cuda().contiguous()
    kstat = torch.zeros(2, H, device="cuda")            # two partial maxima per head (two sequence-parallel ranks)
    kstat[0] = (kc[: S // 2] ** 2).sum(-1).amax(0).cuda()
    kstat[1] = (kc[S // 2:] ** 2).sum(-1).amax(0).cuda()
    bound = qstat.sqrt() * kstat.amax(0).sqrt()
    flags = torch.full((H,), -1, dtype=torch.int32, device="cuda")
    E.check(E.lib().k5_attention_flags(qstat.data_ptr(), kstat.data_ptr(), 2, H, H, 0, flags.data_ptr(), E.stream_ptr()))
    torch.cuda.synchronize()
    assert flags.tolist() == [int(b * 1.002 <= 90.0) for b in bound.tolist()] == [1, 0, 1, 0], (flags, bound)
    assert float(qstat.abs().max()) == 0.0 and float(kstat.abs().max()) == 0.0          # consumed
    qd, kd, vt = q.reshape(S, -1).cuda().to(BF), kc.reshape(S, -1).cuda().to(BF), vt_of(v)
    ref = O.sdpa(q, kc, v, "bf16", None, base2=True)
    for balanced in (False, True):
        out = run_auto(E, qd, kd, vt, H, flags, 0, balanced)
        close(out, ref, what=f"mixed launch (balanced={balanced})")
    # forcing: every head online
    qstat.fill_(1.0); kstat.fill_(1.0)
    E.check(E.lib().k5_attention_flags(qstat.data_ptr(), kstat.data_ptr(), 2, H, H, 1, flags.data_ptr(), E.stream_ptr()))
    assert flags.tolist() == [0, 0, 0, 0]
    # NaN statistics (a diverged activation) must not pick the fixed offset
    qstat.fill_(float("nan")); kstat.fill_(1.0)
    E.check(E.lib().k5_attention_flags(qstat.data_ptr(), kstat.data_ptr(), 2, H, H, 0, flags.data_ptr(), E.stream_ptr()))
    assert flags.tolist() == [0, 0, 0, 0]


def test_fixed_offset_window_edge(E):
    """The fixed-offset form right at its limit: |q||k'| = 89 on every pair's bound, with pairs that reach both ends
    (a key parallel and a key anti-parallel to a query): p spans 2^-89 .. 2^+89 without overflow or a flushed row."""
    H, S = 1, 256
    q = bfr(rnd(S, H, 64, seed=21))
    q = bfr(q / q.norm(dim=-1, keepdim=True) * 8.0)
    k = bfr(rnd(S, H, 64, seed=22))
    k = k / k.norm(dim=-1, keepdim=True) * (89.0 / 8.0)
    k[10, 0] = q[3, 0] / 8.0 * (89.0 / 8.0)      # parallel to q3: score +89
    k[11, 0] = -q[4, 0] / 8.0 * (89.0 / 8.0)     # anti-parallel to q4: score -89; q4's other keys sit near 0
    kc = bfr(k)                                  # already in the exp2 domain
    v = bfr(rnd(S, H, 64, seed=23))
    flags = torch.ones(H, dtype=torch.int32, device="cuda")
    out = run_auto(E, q.reshape(S, -1).cuda().to(BF), kc.reshape(S, -1).cuda().to(BF), vt_of(v), H, flags, 0)
    close(out, O.sdpa(q, kc, v, "bf16", None, base2=True), what="window edge")


def test_config2_size_both_forms_sampled_rows_vs_oracle(E):
    """BASELINE config-2 attention shape (47 616 tokens x 28 heads), the launch the bench times (balanced, per-head flags):
    RMS-normalised q / k like the engine's, heads 0..13 at gain 1 (fixed offset) and 14..27 at gain 3 (online max).
    Sampled query rows of every head against the oracle; V = const -> O = const on ALL rows (softmax sums to one)."""
    N, H = 47616, 28
    g = torch.Generator(device="cuda").manual_seed(0)
    gains = torch.cat([torch.ones(14), torch.full((14,), 3.0)]).cuda()
    def rmsn(x):
        return gains[None, :, None] * x / x.pow(2).mean(-1, keepdim=True).sqrt()
    q = rmsn(torch.randn(N, H, 64, device="cuda", generator=g)).reshape(N, -1).to(BF)
    kc = (rmsn(torch.randn(N, H, 64, device="cuda", generator=g)) * O.SOFTMAX_C).reshape(N, -1).to(BF)
    v = torch.randn(N, H * 64, device="cuda", generator=g).to(BF)
    vt = v.t().contiguous()
    qf, kf = q.float().reshape(N, H, 64), kc.float().reshape(N, H, 64)
    qstat, kstat = (qf * qf).sum(-1).amax(0).contiguous(), (kf * kf).sum(-1).amax(0).contiguous()
    flags = torch.zeros(H, dtype=torch.int32, device="cuda")
    E.check(E.lib().k5_attention_flags(qstat.data_ptr(), kstat.data_ptr(), 1, H, H, 0, flags.data_ptr(), E.stream_ptr()))
    assert flags.tolist() == [1] * 14 + [0] * 14, flags
    out = run_auto(E, q, kc, vt, H, flags, 0, balanced=True)
    rows = torch.tensor([0, 1, 31, 255, 256, 4097, 23808, 40000, 47104, 47615 - 64, 47615])   # incl. rows of the split tail jobs
    ref = O.sdpa(qf[rows].cpu(), kf.cpu(), v.float().cpu().reshape(N, H, 64), "bf16", None, base2=True)
    close(out[rows], ref, ulps=4, atol=5e-3, what="config-2 sampled rows")
    oc = run_auto(E, q, kc, torch.full_like(vt, 0.75), H, flags, 0, balanced=True)
    assert (oc.float() - 0.75).abs().max().item() <= 2 ** -8


def test_config5_size_both_forms_sampled_rows_vs_oracle(E):
    """BASELINE config-5 sequence length (1280x768, 10 s: 234 240 tokens = 3660 blocks), two heads — head 0 on the fixed offset,
    head 1 (gain 3) on the online max — through the balanced launch: sampled query rows against the oracle, and V = const ->
    O = const on every one of the 234 240 rows (every workgroup's softmax sums to one over all 3660 key blocks)."""
    N, H = 234240, 2
    g = torch.Generator(device="cuda").manual_seed(5)
    gains = torch.tensor([1.0, 3.0]).cuda()
    def rmsn(x):
        return gains[None, :, None] * x / x.pow(2).mean(-1, keepdim=True).sqrt()
    q = rmsn(torch.randn(N, H, 64, device="cuda", generator=g)).reshape(N, -1).to(BF)
    kc = (rmsn(torch.randn(N, H, 64, device="cuda", generator=g)) * O.SOFTMAX_C).reshape(N, -1).to(BF)
    v = torch.randn(N, H * 64, device="cuda", generator=g).to(BF)
    vt = v.t().contiguous()
    qf, kf = q.float().reshape(N, H, 64), kc.float().reshape(N, H, 64)
    qstat, kstat = (qf * qf).sum(-1).amax(0).contiguous(), (kf * kf).sum(-1).amax(0).contiguous()
    flags = torch.zeros(H, dtype=torch.int32, device="cuda")
    E.check(E.lib().k5_attention_flags(qstat.data_ptr(), kstat.data_ptr(), 1, H, H, 0, flags.data_ptr(), E.stream_ptr()))
    assert flags.tolist() == [1, 0], flags
    out = run_auto(E, q, kc, vt, H, flags, 0, balanced=True)
    rows = torch.tensor([0, 63, 64, 4097, 117120, 200000, N - 65, N - 1])
    ref = O.sdpa(qf[rows].cpu(), kf.cpu(), v.float().cpu().reshape(N, H, 64), "bf16", None, base2=True)
    close(out[rows], ref, ulps=4, atol=5e-3, what="config-5 sampled rows")
    oc = run_auto(E, q, kc, torch.full_like(vt, 0.75), H, flags, 0, balanced=True)
    assert (oc.float() - 0.75).abs().max().item() <= 2 ** -8


# ------------------------------------------------------------------------------------------ per-row offsets of the fixed-offset form
def run_rows(E, q, kc, vt, H, flags, kmax, balanced=True):
    Sq, Sk = q.shape[0], kc.shape[0]
    out = torch.full((Sq, H * 64), float("nan"), dtype=BF, device="cuda")
    L = E.lib()
    ws = torch.empty(L.k5_attention_balance_size(H, Sq), dtype=torch.uint8, device="cuda") if balanced else None
    E.check(L.k5_attention_bf16_prescaled_rows(q.data_ptr(), kc.data_ptr(), vt.data_ptr(), out.data_ptr(), H, Sq, Sk, q.stride(0),
                                               kc.stride(0), vt.stride(0), out.stride(0), flags.data_ptr(), kmax.data_ptr(),
                                               None if ws is None else ws.data_ptr(), E.stream_ptr()), "k5_attention_bf16_prescaled_rows")
    torch.cuda.synchronize()
    return out


def flags_rows(E, qf, kf, H):
    qstat, kstat = (qf * qf).sum(-1).amax(0).contiguous().cuda(), (kf * kf).sum(-1).amax(0).contiguous().cuda()
    flags = torch.zeros(H, dtype=torch.int32, device="cuda")
    kmax = torch.zeros(H, device="cuda")
    E.check(E.lib().k5_attention_flags_rows(qstat.data_ptr(), kstat.data_ptr(), 1, H, H, 0, flags.data_ptr(), kmax.data_ptr(), E.stream_ptr()))
    torch.cuda.synchronize()
    return flags, kmax


@pytest.mark.parametrize("Sq,Sk,H", [(700 - 700 % 64 + 64, 1088, 4), (33280, 2048, 4)])
def test_row_offsets_keep_large_norm_heads_on_the_fixed_form(E, Sq, Sk, H):
    """Gains 1, 2, 3, 3.7 on heads 0..3: bounds 11.5 g^2 = 11.5 / 46 / 104 / 158 — all within 190, so k5_attention_flags_rows keeps
    every head on the fixed-offset kernel, the last two on non-zero per-row offsets; parity with the oracle, flags untouched
    afterwards (no row underflowed).  (33 280 x 4 heads = 520 jobs: whole rounds AND split tail jobs.)"""
    g = torch.Generator().manual_seed(Sq)
    gains = torch.tensor([1.0, 2.0, 3.0, 3.7])
    def rmsn(x):
        return gains[None, :, None] * x / x.pow(2).mean(-1, keepdim=True).sqrt()
    q = bfr(rmsn(torch.randn(Sq, H, 64, generator=g)))
    k = bfr(rmsn(torch.randn(Sk, H, 64, generator=g)) * O.SOFTMAX_C)
    v = bfr(torch.randn(Sk, H, 64, generator=g))
    flags, kmax = flags_rows(E, q, k, H)
    assert flags.tolist() == [1, 1, 1, 1], flags
    assert (kmax.cpu() * q.pow(2).sum(-1).amax(0).sqrt() > torch.tensor([0.0, 0.0, 90.0, 90.0])).all()    # heads 2, 3 really need an offset
    out = run_rows(E, q.reshape(Sq, -1).cuda().to(BF), k.reshape(Sk, -1).cuda().to(BF), vt_of(v), H, flags, kmax)
    assert flags.tolist() == [1, 1, 1, 1], flags
    rows = torch.arange(Sq) if Sq < 2000 else torch.tensor([0, 5, 255, 256, 4097, 20000, 32767, 32768, 33000, Sq - 1])
    close(out[rows], O.sdpa(q[rows], k, v, "bf16", None, base2=True), ulps=4, atol=5e-3, what="per-row offsets")


def test_row_offsets_late_fallback_when_a_row_underflows(E):
    """Head 0: every key points AWAY from every query (scores ~ -|q||k'| = -140 with a little noise), so the row maxima sit ~280
    below the Cauchy-Schwarz bound the per-row offset (bound - 90 = 50) was built from: exp2(s - 50) underflows for every key, the
    workgroups see l < 2^-100, set head_flags[0] = 0, and the online-max launch of the same call recomputes the head.  Head 1 is an
    ordinary gain-3 head (bound 104) that stays on the fixed form.  Both must match the oracle."""
    Sq, Sk, H = 768, 1024, 2
    g = torch.Generator().manual_seed(9)
    u = torch.randn(64, generator=g); u = u / u.norm()
    q = torch.empty(Sq, H, 64); k = torch.empty(Sk, H, 64)
    q[:, 0] = 28.0 * u + 0.5 * torch.randn(Sq, 64, generator=g)
    k[:, 0] = -5.0 * u + 0.1 * torch.randn(Sk, 64, generator=g)          # exp2-domain keys: |q||k'| ~ 140
    def rmsn(x, gain):
        return gain * x / x.pow(2).mean(-1, keepdim=True).sqrt()
    q[:, 1] = rmsn(torch.randn(Sq, 64, generator=g), 3.0)
    k[:, 1] = rmsn(torch.randn(Sk, 64, generator=g), 3.0) * O.SOFTMAX_C
    q, k = bfr(q), bfr(k)
    v = bfr(torch.randn(Sk, H, 64, generator=g))
    flags, kmax = flags_rows(E, q, k, H)
    assert flags.tolist() == [1, 1], flags                                 # both bounds are <= 190: the fixed form is tried first
    s0 = (q[:, 0] @ k[:, 0].t())
    assert s0.max().item() < -100 and (kmax[0].item() * q[:, 0].norm(dim=-1).min().item()) > 130
    ref = O.sdpa(q, k, v, "bf16", None, base2=True)
    # with a workspace the fallback is per JOB (round 3): the underflowing (head, 256-query) jobs are redone by the online launch of the
    # same call and the head's flag is left alone
    out = run_rows(E, q.reshape(Sq, -1).cuda().to(BF), k.reshape(Sk, -1).cuda().to(BF), vt_of(v), H, flags, kmax)
    assert flags.tolist() == [1, 1], flags
    close(out, ref, ulps=4, atol=5e-3, what="late fallback, per job")
    # without one (no room for job flags): the whole head, as in round 2
    out = run_rows(E, q.reshape(Sq, -1).cuda().to(BF), k.reshape(Sk, -1).cuda().to(BF), vt_of(v), H, flags, kmax, balanced=False)
    assert flags.tolist() == [0, 1], flags                                 # head 0 fell back late, head 1 did not
    close(out, ref, ulps=4, atol=5e-3, what="late fallback, per head")


def test_row_offsets_fallback_is_per_job(E):
    """What the per-job flags are for: ONE query block of a head underflows (its queries point away from every key: scores ~ -140
    against an offset of +50) while the head's other 129 blocks are ordinary rows on non-zero offsets.  33 280 rows x 4 heads = 520
    jobs = whole-round jobs and split tail jobs; the bad block sits once in a whole-round job (block 3 of head 1) and once in a tail job
    (the last block of head 3).  Only those jobs are redone by the online launch — the head flags stay on the fixed form — and every
    row matches the oracle."""
    Sq, Sk, H = 33280, 2048, 4
    g = torch.Generator().manual_seed(33)
    def rmsn(x, gain):
        return gain * x / x.pow(2).mean(-1, keepdim=True).sqrt()
    q = rmsn(torch.randn(Sq, H, 64, generator=g), 3.3)
    k = rmsn(torch.randn(Sk, H, 64, generator=g), 3.3) * O.SOFTMAX_C
    for h, blk in ((1, 3), (3, Sq // 256 - 1)):
        u = k[:, h].mean(0); u = u / u.norm()
        kk = k[:, h]
        k[:, h] = kk + (4.0 - kk @ u)[:, None] * u[None, :]                   # every key of the head gets the same component 4 along u ...
        q[256 * blk:256 * blk + 256, h] = -26.0 * u + 0.3 * torch.randn(256, 64, generator=g)   # ... and the bad block's queries point against it
    q, k = bfr(q), bfr(k)
    v = bfr(torch.randn(Sk, H, 64, generator=g))
    flags, kmax = flags_rows(E, q, k, H)
    assert flags.tolist() == [1, 1, 1, 1], flags
    s_bad = q[256 * 3:256 * 3 + 256, 1] @ k[:, 1].t()
    off_bad = q[256 * 3:256 * 3 + 256, 1].norm(dim=-1) * kmax[1].item() - 90
    assert (s_bad.max(-1).values < off_bad - 110).all(), (s_bad.max().item(), off_bad.min().item())     # really below 2^-100 after the offset
    out = run_rows(E, q.reshape(Sq, -1).cuda().to(BF), k.reshape(Sk, -1).cuda().to(BF), vt_of(v), H, flags, kmax)
    assert flags.tolist() == [1, 1, 1, 1], flags
    rows = torch.cat([torch.arange(0, Sq, 1237), torch.arange(256 * 3, 256 * 4, 17), torch.arange(Sq - 256, Sq, 13), torch.arange(Sq - 2048, Sq - 256, 211)])
    close(out[rows], O.sdpa(q[rows], k, v, "bf16", None, base2=True), ulps=4, atol=5e-3, what="per-job fallback")


@pytest.mark.parametrize("bad_half", ["first", "both"])
def test_row_offsets_late_fallback_across_passes(E, bad_half):
    """The sequence-parallel schedule's form of the same thing (k5_attention_bf16_prescaled_rows_pass): pass A over the first half of
    the keys leaves the fp32 state, pass B resumes it over the second half and normalises.  Head 0's keys of the first half point
    away from every query (its rows underflow in pass A: head_flags[0] = 2, 'late'); in "first" its second-half keys are ordinary,
    in "both" they point away too.  Either way every later fixed-offset launch skips the head, pass A's and B's states of that head
    are ignored, and the ONLINE launch of the last pass recomputes it from scratch over all keys.  Head 1 (gain 3) stays fixed."""
    Sq, Sk, H = 512, 2048, 2
    g = torch.Generator().manual_seed(21)
    u = torch.randn(64, generator=g); u = u / u.norm()
    q = torch.empty(Sq, H, 64); k = torch.empty(Sk, H, 64)
    q[:, 0] = 28.0 * u + 0.5 * torch.randn(Sq, 64, generator=g)
    k[:, 0] = -5.0 * u + 0.1 * torch.randn(Sk, 64, generator=g)
    if bad_half == "first":
        k[Sk // 2:, 0] = 0.5 * torch.randn(Sk // 2, 64, generator=g)     # ordinary keys: the row's true maximum lives in pass B
    def rmsn(x, gain):
        return gain * x / x.pow(2).mean(-1, keepdim=True).sqrt()
    q[:, 1] = rmsn(torch.randn(Sq, 64, generator=g), 3.0)
    k[:, 1] = rmsn(torch.randn(Sk, 64, generator=g), 3.0) * O.SOFTMAX_C
    q, k = bfr(q), bfr(k)
    v = bfr(torch.randn(Sk, H, 64, generator=g))
    flags, kmax = flags_rows(E, q, k, H)
    assert flags.tolist() == [1, 1], flags
    qd, kd, vt = q.reshape(Sq, -1).cuda().to(BF), k.reshape(Sk, -1).cuda().to(BF), vt_of(v)
    out = torch.full((Sq, H * 64), float("nan"), dtype=BF, device="cuda")
    L = E.lib()
    state = torch.zeros(L.k5_attention_state_size(H, Sq) // 4, device="cuda")
    ws = torch.empty(L.k5_attention_balance_size(H, Sq), dtype=torch.uint8, device="cuda")
    T = Sk // 64
    def run_pass(t0, cnt, fl, late):
        E.check(L.k5_attention_bf16_prescaled_rows_pass(qd.data_ptr(), kd.data_ptr(), vt.data_ptr(), out.data_ptr(), H, Sq, Sk, qd.stride(0),
                                                        kd.stride(0), vt.stride(0), out.stride(0), flags.data_ptr(), kmax.data_ptr(), t0, cnt,
                                                        state.data_ptr(), fl, late, ws.data_ptr(), E.stream_ptr()), "rows_pass")
        torch.cuda.synchronize()
    run_pass(0, T // 2, 2, 1)
    assert flags.tolist() == [1, 1], flags            # per-job flags (in the workspace): head 0's jobs went late in pass A, its head flag stays
    run_pass(T // 2, T - T // 2, 1, 2)
    assert flags.tolist() == [1, 1], flags
    close(out, O.sdpa(q, k, v, "bf16", None, base2=True), ulps=4, atol=5e-3, what=f"late fallback across passes ({bad_half})")


# ------------------------------------------------------------------------------------------ norm_qk + RoPE of the queries fused into the Q load
def run_qnorm(E, qraw, w, cos, sin, kc, vt, H, flags, kmax, passes=None):
    """k5_attention_bf16_prescaled_qnorm_pass: one launch over all keys, or (passes = [(tile0, cnt, state_flags, late_pass), ...]) a
    multi-pass schedule with the fp32 state between the passes."""
    Sq, Sk = qraw.shape[0], kc.shape[0]
    out = torch.full((Sq, H * 64), float("nan"), dtype=BF, device="cuda")
    L = E.lib()
    ws = torch.empty(L.k5_attention_balance_size(H, Sq), dtype=torch.uint8, device="cuda")
    state = torch.zeros(L.k5_attention_state_size(H, Sq) // 4, device="cuda") if passes else None
    for (t0, cnt, fl, late) in (passes or [(0, -1, 0, 0)]):
        E.check(L.k5_attention_bf16_prescaled_qnorm_pass(qraw.data_ptr(), kc.data_ptr(), vt.data_ptr(), out.data_ptr(), H, Sq, Sk, qraw.stride(0),
                                                         kc.stride(0), vt.stride(0), out.stride(0), w.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                                         None if flags is None else flags.data_ptr(), None if kmax is None else kmax.data_ptr(),
                                                         t0, cnt, None if state is None else state.data_ptr(), fl, late, ws.data_ptr(),
                                                         E.stream_ptr()), "k5_attention_bf16_prescaled_qnorm_pass")
        torch.cuda.synchronize()
    return out


def qnorm_case(Sq, Sk, H, head_gain, seed, row_scale=None):
    """raw query projection (any scale: the norm removes it), RMSNorm weights w (one set for all heads, as in the model) x per-head
    gain folded into the raw keys instead (so heads differ in their bound), rotary table, pre-scaled normed keys, values"""
    g = torch.Generator().manual_seed(seed)
    qraw = bfr(torch.randn(Sq, H, 64, generator=g) * 2.5)
    if row_scale is not None:
        qraw = bfr(qraw * row_scale[:, None, None])
    w = torch.randn(64, generator=g).abs() * 0.3 + 0.85
    ang = torch.randn(Sq, 32, generator=g) * 3.0
    cos, sin = torch.cos(ang), torch.sin(ang)
    k = torch.randn(Sk, H, 64, generator=g)
    k = bfr(head_gain[None, :, None] * k / k.pow(2).mean(-1, keepdim=True).sqrt() * O.SOFTMAX_C)
    v = bfr(torch.randn(Sk, H, 64, generator=g))
    return qraw, w, cos, sin, k, v


def normed_queries(E, qraw, w, cos, sin, H):
    """the standalone kernel's result on the same raw queries (its oracle parity: test_gpu_kernels.py)"""
    qd = qraw.reshape(qraw.shape[0], -1).cuda().to(BF).clone()
    wd, cd, sd = w.cuda(), cos.cuda(), sin.cuda()
    E.rmsnorm_rope_(qd, wd, cd, sd, heads=H)
    torch.cuda.synchronize()
    return qd


def kflags(E, k, H):
    """k5_attention_flags_rows with a zero query statistic: every head starts on the fixed form, kmax = max|k'_h| with margin"""
    kstat = (k * k).sum(-1).amax(0).contiguous().cuda()
    qstat = torch.zeros(H, device="cuda")
    flags = torch.zeros(H, dtype=torch.int32, device="cuda")
    kmax = torch.zeros(H, device="cuda")
    E.check(E.lib().k5_attention_flags_rows(qstat.data_ptr(), kstat.data_ptr(), 1, H, H, 0, flags.data_ptr(), kmax.data_ptr(), E.stream_ptr()))
    torch.cuda.synchronize()
    return flags, kmax


@pytest.mark.parametrize("Sq,Sk,H", [(768, 1088, 4), (33280, 2048, 4)])
def test_fused_query_norm_matches_the_standalone_norm(E, Sq, Sk, H):
    """Heads with key gains 1, 5, 10, 24 (|q| ~ 9-10.5, |k'| = 1.44 gain: bounds ~ 15 / 76 / 150 / 360): the first three stay on the fixed
    form (the third on non-zero per-row offsets), and the kernel's own decision sends the last one (beyond the limit of 190) to the
    online form.  Reference: the attention on the queries the standalone norm kernel wrote (same arithmetic: equal up to the rounding
    of a few fp32 sums), and the oracle."""
    qraw, w, cos, sin, k, v = qnorm_case(Sq, Sk, H, torch.tensor([1.0, 5.0, 10.0, 24.0]), Sq)
    qn = normed_queries(E, qraw, w, cos, sin, H)
    qnf = qn.float().cpu().reshape(Sq, H, 64)
    bound = qnf.norm(dim=-1).amax(0) * k.norm(dim=-1).amax(0)
    assert bound[2] > 95 and bound[2] < 175 and bound[3] > 310 and bound[1] < 88, bound
    wd, cd, sd = w.cuda(), cos.cuda(), sin.cuda()
    kd, vt = k.reshape(Sk, -1).cuda().to(BF), vt_of(v)
    flags, kmax = kflags(E, k, H)
    assert flags.tolist() == [1, 1, 1, 1]
    out = run_qnorm(E, qraw.reshape(Sq, -1).cuda().to(BF), wd, cd, sd, kd, vt, H, flags, kmax)
    assert flags.tolist() == [1, 1, 1, 0], flags                       # decided by the fixed-offset workgroups themselves
    f2, kmax2 = flags_rows(E, qnf, k, H)
    assert f2.tolist() == [1, 1, 1, 0]                                   # ... as the statistics-based rule decides on the normed queries
    ref_k = run_rows(E, qn, kd, vt, H, f2, kmax2)
    # BIT-IDENTICAL since round 6: the fused norm restates the standalone kernel's arithmetic operation for operation — the order of the sum of squares
    # (8-dimension chunks, then a tree) and WHICH product of a rotation is fused into the add (`__fmul_rn` / `__fadd_rn` are plain * and + to hipcc, so
    # -ffp-contract=fast had picked one per kernel: tools/probes/qn_arith_probe.hip).  Before: 1e-4 of the (row, head) pairs differed in one query element.
    assert torch.equal(out, ref_k), ((out.float() - ref_k.float()).abs().max().item(), (out != ref_k).float().mean().item())
    rows = torch.arange(Sq) if Sq < 2000 else torch.tensor([0, 5, 255, 256, 4097, 20000, 32767, 32768, 33000, Sq - 1])
    close(out[rows], O.sdpa(qnf[rows], k, v, "bf16", None, base2=True), ulps=4, atol=5e-3, what="fused query norm")
    # forced online max (no flags): the same queries through the other form
    out_on = run_qnorm(E, qraw.reshape(Sq, -1).cuda().to(BF), wd, cd, sd, kd, vt, H, None, None)
    close(out_on[rows], O.sdpa(qnf[rows], k, v, "bf16", None, base2=True), ulps=4, atol=5e-3, what="fused query norm, online")


def test_fused_query_norm_one_large_row_flips_its_head(E):
    """The decision is per HEAD although it is taken per workgroup: w has one large channel and a single query row (in the last
    256-row block) has its energy in that channel, so only that row's bound exceeds 190 — its wave flips the flag at the end of its
    tile loop, after most other workgroups of the head have already finished in the fixed form; the online launch then recomputes
    the whole head."""
    Sq, Sk, H = 2048, 1024, 2
    qraw, w, cos, sin, k, v = qnorm_case(Sq, Sk, H, torch.tensor([7.0, 7.0]), 5)
    w[7] = 3.0
    qraw[Sq - 3, 0] = 0.01 * qraw[Sq - 3, 0]
    qraw[Sq - 3, 0, 7] = 30.0                                           # after the norm: ~ 8 * 3 = 24 in channel 7, |q| |k'| ~ 24 * 10 = 240
    qraw = bfr(qraw)
    qn = normed_queries(E, qraw, w, cos, sin, H)
    qnf = qn.float().cpu().reshape(Sq, H, 64)
    b = qnf.norm(dim=-1) * k.norm(dim=-1).amax(0)[None]
    assert (b[:, 0] > 190).sum() == 1 and b[Sq - 3, 0] > 200 and b[:, 1].max() < 185, (b[:, 0].topk(3), b[:, 1].max())
    wd, cd, sd = w.cuda(), cos.cuda(), sin.cuda()
    kd, vt = k.reshape(Sk, -1).cuda().to(BF), vt_of(v)
    flags, kmax = kflags(E, k, H)
    out = run_qnorm(E, qraw.reshape(Sq, -1).cuda().to(BF), wd, cd, sd, kd, vt, H, flags, kmax)
    assert flags.tolist() == [0, 1], flags
    close(out, O.sdpa(qnf, k, v, "bf16", None, base2=True), ulps=4, atol=5e-3, what="one large row")


def test_fused_query_norm_across_passes(E):
    """The sequence-parallel schedule: pass A (first half of the keys, state out, late_pass 1), pass B (resume, normalise, late_pass 2).
    Head 2 leaves the fixed form in pass A by the kernel's own decision — flag 0 (online from pass A on) or, when a row of another
    wave underflowed on its over-large offset first, flag 2 (late: recomputed from scratch by the online launch of pass B); both are
    the online form on all keys."""
    Sq, Sk, H = 1024, 2048, 3
    qraw, w, cos, sin, k, v = qnorm_case(Sq, Sk, H, torch.tensor([2.0, 10.0, 24.0]), 77)
    qn = normed_queries(E, qraw, w, cos, sin, H)
    qnf = qn.float().cpu().reshape(Sq, H, 64)
    wd, cd, sd = w.cuda(), cos.cuda(), sin.cuda()
    kd, vt = k.reshape(Sk, -1).cuda().to(BF), vt_of(v)
    flags, kmax = kflags(E, k, H)
    T = Sk // 64
    out = run_qnorm(E, qraw.reshape(Sq, -1).cuda().to(BF), wd, cd, sd, kd, vt, H, flags, kmax,
                    passes=[(0, T // 2, 2, 1), (T // 2, T - T // 2, 1, 2)])
    assert flags.tolist() == [1, 1, 2], flags       # a flip in a multi-pass schedule is always the LATE flag (recompute from scratch)
    close(out, O.sdpa(qnf, k, v, "bf16", None, base2=True), ulps=4, atol=5e-3, what="fused query norm across passes")


def test_fused_query_norm_flip_from_a_tail_job_across_passes(E):
    """ADVICE r2 (attn_fwd.hip QN flip): 4 heads x 130 query blocks = 520 jobs = one full round of the 512 resident workgroups + 8
    TAIL jobs (head 3, the last 8 query blocks), which the balanced launcher runs after BOTH forms' full-round launches.  Head 3's
    rows sit on non-zero per-row offsets (bound ~ 110) and exactly one row — in the very last query block, i.e. in a tail job — is over
    the limit of 190: its flip lands after online(full) has skipped the head, while the full jobs of the head left fixed-form state.
    The flip must be the LATE flag (2) so that pass B's online launch recomputes the head from scratch instead of resuming that
    state as offset 0."""
    Sq, Sk, H = 33280, 2048, 4
    qraw, w, cos, sin, k, v = qnorm_case(Sq, Sk, H, torch.tensor([2.0, 2.0, 2.0, 7.5]), 91)
    w[7] = 3.0
    qraw[Sq - 3, 3] = 0.01 * qraw[Sq - 3, 3]
    qraw[Sq - 3, 3, 7] = 30.0
    qraw = bfr(qraw)
    qn = normed_queries(E, qraw, w, cos, sin, H)
    qnf = qn.float().cpu().reshape(Sq, H, 64)
    b = qnf.norm(dim=-1) * k.norm(dim=-1).amax(0)[None]
    assert (b[:, 3] > 190).sum() == 1 and b[Sq - 3, 3] > 200 and b[:, :3].max() < 88 and b[:, 3].median() > 92, (b[:, 3].topk(3), b[:, 3].median())
    wd, cd, sd = w.cuda(), cos.cuda(), sin.cuda()
    kd, vt = k.reshape(Sk, -1).cuda().to(BF), vt_of(v)
    flags, kmax = kflags(E, k, H)
    T = Sk // 64
    out = run_qnorm(E, qraw.reshape(Sq, -1).cuda().to(BF), wd, cd, sd, kd, vt, H, flags, kmax,
                    passes=[(0, T // 2, 2, 1), (T // 2, T - T // 2, 1, 2)])
    assert flags.tolist() == [1, 1, 1, 2], flags
    rows = torch.cat([torch.arange(0, Sq - 2048, 997), torch.arange(Sq - 2048, Sq, 61), torch.tensor([Sq - 3, Sq - 1])])
    close(out[rows], O.sdpa(qnf[rows], k, v, "bf16", None, base2=True), ulps=4, atol=5e-3, what="flip from a tail job across passes")


# ------------------------------------------------------------------------------------------ centred per-row offsets (round 3)
def flags_rows_centred(E, qf, kf, H):
    """centre = the mean key of each head (a convex combination), radii around it; k5_attention_flags_rows_centred"""
    c = kf.mean(0)                                                               # (H, 64)
    r2 = ((kf - c[None]) ** 2).sum(-1).amax(0).contiguous().cuda()
    qstat, kstat = (qf * qf).sum(-1).amax(0).contiguous().cuda(), (kf * kf).sum(-1).amax(0).contiguous().cuda()
    flags = torch.zeros(H, dtype=torch.int32, device="cuda")
    kmax, krad = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
    E.check(E.lib().k5_attention_flags_rows_centred(qstat.data_ptr(), kstat.data_ptr(), 1, H, H, 0, flags.data_ptr(), kmax.data_ptr(), r2.data_ptr(),
                                                    krad.data_ptr(), E.stream_ptr()), "k5_attention_flags_rows_centred")
    torch.cuda.synchronize()
    assert (r2 == 0).all()                                                        # consumed
    return flags, kmax, krad, c.contiguous().cuda()


def run_rows_centred(E, q, kc, vt, H, flags, kmax, centre, krad, balanced=True):
    Sq, Sk = q.shape[0], kc.shape[0]
    out = torch.full((Sq, H * 64), float("nan"), dtype=BF, device="cuda")
    L = E.lib()
    ws = torch.empty(L.k5_attention_balance_size(H, Sq), dtype=torch.uint8, device="cuda") if balanced else None
    E.check(L.k5_attention_bf16_prescaled_rows_centred(q.data_ptr(), kc.data_ptr(), vt.data_ptr(), out.data_ptr(), H, Sq, Sk, q.stride(0), kc.stride(0),
                                                       vt.stride(0), out.stride(0), flags.data_ptr(), kmax.data_ptr(), centre.data_ptr(), krad.data_ptr(),
                                                       None if ws is None else ws.data_ptr(), E.stream_ptr()), "k5_attention_bf16_prescaled_rows_centred")
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("Sq,Sk", [(768, 1088 - 1088 % 64), (33280, 2048)])
def test_centred_offsets_on_keys_with_a_common_component(E, Sq, Sk):
    """Keys that share a large mean direction (what the keys of a layer look like when the tokens' activations are correlated): every score
    of a row sits near q.c — far below the plain Cauchy-Schwarz bound |q| max|k'| when q.c < 0, so the plain offset underflows whole rows
    (head 0: all scores ~ -315 under a plain bound of 330: beyond the plain window, online form), and far ABOVE 90 when q.c > 0
    (head 1: scores ~ +315: an offset is indispensable).  Head 2: no common component, plain bound 150, radius ~ the norm (centring buys
    nothing and costs nothing).  Head 3: small norms (plain bound < 90): offset 0 as ever.  With the centred bound every head has
    |q| R <= 190: the fixed form on every head, NO fallback possible — run without a workspace, where a single underflowing row would flip
    its head's flag — and oracle parity."""
    H = 4
    g = torch.Generator().manual_seed(Sq + 1)
    def unit(x):
        return x / x.norm(dim=-1, keepdim=True)
    u = unit(torch.randn(H, 64, generator=g))
    q = torch.empty(Sq, H, 64); k = torch.empty(Sk, H, 64)
    k[:, 0] = 7.0 * u[0] + 1.6 * unit(torch.randn(Sk, 64, generator=g));  q[:, 0] = -45.0 * u[0] + 10.0 * unit(torch.randn(Sq, 64, generator=g))
    k[:, 1] = 7.0 * u[1] + 1.6 * unit(torch.randn(Sk, 64, generator=g));  q[:, 1] = +45.0 * u[1] + 10.0 * unit(torch.randn(Sq, 64, generator=g))
    k[:, 2] = 5.0 * unit(torch.randn(Sk, 64, generator=g));               q[:, 2] = 30.0 * unit(torch.randn(Sq, 64, generator=g))
    k[:, 3] = 1.5 * unit(torch.randn(Sk, 64, generator=g));               q[:, 3] = 10.0 * unit(torch.randn(Sq, 64, generator=g))
    q, k = bfr(q), bfr(k)
    v = bfr(torch.randn(Sk, H, 64, generator=g))
    s0, s1 = q[:64, 0] @ k[:, 0].t(), q[:64, 1] @ k[:, 1].t()
    assert s0.max().item() < -150 and s1.min().item() > 150
    f_plain, kmax_plain = flags_rows(E, q, k, H)
    assert f_plain.tolist() == [0, 0, 1, 1], f_plain                       # plain bounds ~ 330 / 330 / 150 / 15: heads 0, 1 beyond the window of 190
    flags, kmax, krad, centre = flags_rows_centred(E, q, k, H)
    nqR = q.norm(dim=-1).amax(0) * krad.cpu()
    assert flags.tolist() == [1, 1, 1, 1] and (nqR[:2] < 190).all() and nqR[2] < 190, (flags, nqR)
    qd, kd, vt = q.reshape(Sq, -1).cuda().to(BF), k.reshape(Sk, -1).cuda().to(BF), vt_of(v)
    rows = torch.arange(Sq) if Sq < 2000 else torch.tensor([0, 5, 255, 256, 4097, 20000, 32767, 32768, 33000, Sq - 1])
    ref = O.sdpa(q[rows], k, v, "bf16", None, base2=True)
    for balanced in (False, True):
        out = run_rows_centred(E, qd, kd, vt, H, flags, kmax, centre, krad, balanced=balanced)
        assert flags.tolist() == [1, 1, 1, 1], (balanced, flags)           # nothing underflowed: no head was sent to the online form late
        close(out[rows], ref, ulps=4, atol=5e-3, what=f"centred offsets (balanced={balanced})")
    # the plain form of the same call: heads 0 and 1 on the online max — the same numbers
    out_p = run_rows(E, qd, kd, vt, H, f_plain, kmax_plain)
    close(out_p[rows], ref, ulps=4, atol=5e-3, what="plain offsets / online form")


def test_key_centre_and_radius_from_the_norm_pass(E):
    """k5_rmsnorm_rope_centre_bf16: the centres are the mean of the kernel's own k' over its strided sample of rows (convex combination of
    actual keys), the appended statistics the squared radii max |k' - c|^2 over ALL rows, the first H entries the squared norms as before."""
    rows, Hq, D = 3000, 3, 64
    H = 2 * Hq                                                                 # a q | k projection: heads 0..2 queries, 3..5 keys (scaled, centred)
    g = torch.Generator().manual_seed(3)
    base = torch.randn(1, H, D, generator=g) * 6.0
    x = bfr(base + torch.randn(rows, H, D, generator=g))
    w = torch.randn(2, 64, generator=g).abs() * 0.3 + 0.85
    ang = torch.randn(rows, 32, generator=g) * 0.05
    cos, sin = torch.cos(ang), torch.sin(ang)
    xd = x.reshape(rows, -1).cuda().to(BF).clone()
    stats = torch.zeros(H + Hq, device="cuda")
    centre = torch.full((Hq, 64), float("nan"), device="cuda")
    wd, cd, sd = w.cuda(), cos.cuda(), sin.cuda()                               # (keep them alive: a temporary's memory is reused at once)
    E.check(E.lib().k5_rmsnorm_rope_centre_bf16(xd.data_ptr(), wd.data_ptr(), cd.data_ptr(), sd.data_ptr(), rows, H, xd.stride(0),
                                                Hq, H, O.SOFTMAX_C, Hq, stats.data_ptr(), centre.data_ptr(), E.stream_ptr()), "k5_rmsnorm_rope_centre_bf16")
    torch.cuda.synchronize()
    kp = xd.float().cpu().reshape(rows, H, D)[:, Hq:]                           # the keys the kernel wrote
    ns = 256                                                                   # K5_CENTRE_SAMPLE (small_ops.hip)
    idx = (torch.arange(ns) * rows) // ns
    want_c = kp[idx].mean(0)
    assert (centre.cpu() - want_c).abs().max().item() <= 2e-3 * want_c.abs().max().item() + 1e-4
    r2 = ((kp - centre.cpu()[None]) ** 2).sum(-1).amax(0)
    n2 = (xd.float().cpu().reshape(rows, H, D) ** 2).sum(-1).amax(0)
    assert torch.allclose(stats[:H].cpu(), n2, rtol=1e-5) and torch.allclose(stats[H:].cpu(), r2, rtol=1e-4), (stats, n2, r2)
    assert (r2 < 0.5 * n2[Hq:]).all()                                          # these keys do share a common component


# ------------------------------------------------------------------------------------------ anchored offsets beyond the window (round 3)
def flags_rows_anchored(E, qf, kf, H, prefer=None):
    c = kf.mean(0)
    r2 = ((kf - c[None]) ** 2).sum(-1).amax(0).contiguous().cuda()
    qstat, kstat = (qf * qf).sum(-1).amax(0).contiguous().cuda(), (kf * kf).sum(-1).amax(0).contiguous().cuda()
    flags = torch.zeros(H, dtype=torch.int32, device="cuda")
    kmax, krad = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
    E.check(E.lib().k5_attention_flags_rows_anchored(qstat.data_ptr(), kstat.data_ptr(), 1, H, H, 0, flags.data_ptr(), kmax.data_ptr(), r2.data_ptr(),
                                                     krad.data_ptr(), None if prefer is None else prefer.data_ptr(), E.stream_ptr()),
            "k5_attention_flags_rows_anchored")
    torch.cuda.synchronize()
    return flags, kmax, krad, c.contiguous().cuda()


def run_rows_anchored(E, q, kc, vt, H, flags, kmax, centre, krad, balanced=True, key0=0):
    Sq, Sk = q.shape[0], kc.shape[0]
    L = E.lib()
    anchor = torch.full((H, Sq), float("nan"), device="cuda")
    E.check(L.k5_attention_row_anchor(q.data_ptr(), kc.data_ptr(), H, Sq, Sk, q.stride(0), kc.stride(0), key0, Sk, kmax.data_ptr(), anchor.data_ptr(),
                                      E.stream_ptr()), "k5_attention_row_anchor")
    out = torch.full((Sq, H * 64), float("nan"), dtype=BF, device="cuda")
    ws = torch.empty(L.k5_attention_balance_size(H, Sq), dtype=torch.uint8, device="cuda") if balanced else None
    E.check(L.k5_attention_bf16_prescaled_rows_anchored(q.data_ptr(), kc.data_ptr(), vt.data_ptr(), out.data_ptr(), H, Sq, Sk, q.stride(0), kc.stride(0),
                                                        vt.stride(0), out.stride(0), flags.data_ptr(), kmax.data_ptr(), centre.data_ptr(), krad.data_ptr(),
                                                        anchor.data_ptr(), None if ws is None else ws.data_ptr(), E.stream_ptr()),
            "k5_attention_bf16_prescaled_rows_anchored")
    torch.cuda.synchronize()
    return out, anchor


def _anchor_offsets(q, k, key0=0):
    """host restatement of attn_row_anchor_kernel: sample = own 64-token block (4 tiles of 16) + 28 strided 16-key tiles; returns the
    sample maximum and ceil(max + min(60, spread (max - mean))) + 20, spread = 1.4 (sqrt(2 ln Sk) - c) / c, c = sqrt(2 ln 512)"""
    import math
    c = math.sqrt(2 * math.log(512))
    spread = 1.4 * (math.sqrt(2 * math.log(max(k.shape[0], 512))) - c) / c
    Sq, Sk = q.shape[0], k.shape[0]
    nt = Sk // 16
    strided = [((i - 4) * nt) // 28 for i in range(4, 32)]
    out, mean = torch.empty(Sq), torch.empty(Sq)
    for b in range((Sq + 63) // 64):
        own = [4 * min(max((64 * b + key0) // 64, 0), nt // 4 - 1) + i for i in range(4)]
        idx = torch.cat([torch.arange(16 * t, 16 * t + 16) for t in own + strided])
        rows = slice(64 * b, min(64 * b + 64, Sq))
        sc = q[rows] @ k[idx].t()
        out[rows], mean[rows] = sc.amax(-1), sc.mean(-1)
    return out, torch.ceil(out + (spread * (out - mean)).clamp(0.0, 60.0)) + 20.0


@pytest.mark.parametrize("Sq,Sk", [(768, 1024), (33280, 2048)])
def test_anchored_offsets_beyond_the_window(E, Sq, Sk):
    """Heads whose Cauchy-Schwarz bound (plain AND centred) lies far beyond 190 used to take the online form.  With anchored offsets they
    keep the fixed form: head 0 — large norms, random directions (scores ~ N(0, 75^2), bound 600); head 1 — a large common component AND a
    large radius (centred bound ~ 400); head 2 — inside the centred window (untouched: its kmax stays positive); head 3 — small norms
    (offset 0).  The anchors are the host restatement's; with a workspace no HEAD leaves the fixed form (a job may: the maximum of
    iid scores over all keys occasionally lies beyond the exact range — rows counted below) and everything matches the oracle."""
    H = 4
    g = torch.Generator().manual_seed(Sq + 7)
    def unit(x):
        return x / x.norm(dim=-1, keepdim=True)
    u = unit(torch.randn(H, 64, generator=g))
    q = torch.empty(Sq, H, 64); k = torch.empty(Sk, H, 64)
    k[:, 0] = 10.0 * unit(torch.randn(Sk, 64, generator=g));              q[:, 0] = 60.0 * unit(torch.randn(Sq, 64, generator=g))
    k[:, 1] = 6.0 * u[1] + 8.0 * unit(torch.randn(Sk, 64, generator=g));  q[:, 1] = 30.0 * u[1] + 40.0 * unit(torch.randn(Sq, 64, generator=g))
    k[:, 2] = 5.0 * unit(torch.randn(Sk, 64, generator=g));               q[:, 2] = 30.0 * unit(torch.randn(Sq, 64, generator=g))
    k[:, 3] = 1.5 * unit(torch.randn(Sk, 64, generator=g));               q[:, 3] = 10.0 * unit(torch.randn(Sq, 64, generator=g))
    q, k = bfr(q), bfr(k)
    v = bfr(torch.randn(Sk, H, 64, generator=g))
    f_c, _, krad_c, _ = flags_rows_centred(E, q, k, H)
    assert f_c.tolist() == [0, 0, 1, 1], f_c                                # without anchoring: heads 0, 1 on the online form
    flags, kmax, krad, centre = flags_rows_anchored(E, q, k, H)
    assert flags.tolist() == [1, 1, 1, 1] and (kmax[:2] < 0).all() and (kmax[2:] > 0).all(), (flags, kmax)   # marked = negative, |entry| = max|k'| (still a bound)
    assert torch.allclose(-kmax[:2].cpu(), k.norm(dim=-1).amax(0)[:2] * 1.002, rtol=1e-3), kmax
    qd, kd, vt = q.reshape(Sq, -1).cuda().to(BF), k.reshape(Sk, -1).cuda().to(BF), vt_of(v)
    rows = torch.arange(Sq) if Sq < 2000 else torch.tensor([0, 5, 255, 256, 4097, 20000, 32767, 32768, 33000, Sq - 1])
    ref = O.sdpa(q[rows], k, v, "bf16", None, base2=True)
    out, anchor = run_rows_anchored(E, qd, kd, vt, H, flags, kmax, centre, krad, balanced=True)
    assert flags.tolist() == [1, 1, 1, 1], flags
    close(out[rows], ref, ulps=4, atol=5e-3, what="anchored offsets")
    beyond = 0
    for h in (0, 1):
        smax, want = _anchor_offsets(q[:, h], k[:, h])
        got = anchor[h].cpu()
        assert ((got - want).abs() <= 1.0).all() and ((got - want) != 0).float().mean() < 0.01, (h, (got - want).abs().max())   # ceil at an MFMA-rounding edge
        true_max = torch.cat([(q[i:i + 4096, h] @ k[:, h].t()).amax(-1) for i in range(0, Sq, 4096)])
        assert (true_max >= smax - 1e-3).all() and (true_max - got > -81.0).all()   # the row's own maximum term is >= 2^-81: no underflow
        beyond += int((true_max - got >= 112.0).sum())                      # rows whose sum passes 2^112: their jobs fell back
    assert beyond <= Sq // 1000, beyond
    assert torch.isnan(anchor[2:]).all()                                    # rows of unmarked heads are not touched
    if beyond == 0:   # then nothing may have fallen back at all: without a workspace a single overflowing row would flip its head
        flags2, kmax2, krad2, centre2 = flags_rows_anchored(E, q, k, H)
        out2, _ = run_rows_anchored(E, qd, kd, vt, H, flags2, kmax2, centre2, krad2, balanced=False)
        assert flags2.tolist() == [1, 1, 1, 1], flags2
        close(out2[rows], ref, ulps=4, atol=5e-3, what="anchored offsets (no workspace)")


def test_marked_heads_attended_without_anchors_fall_back_safely(E):
    """k5_attention_flags_rows_anchored marks a head beyond the window by a NEGATIVE kmax entry; a caller of the public ABI that
    then attends through k5_attention_bf16_prescaled_rows / _rows_centred (no row anchors) used to run such a head on offset 0 with the
    underflow / overflow check skipped — exp2 unguarded at bounds of 400-600.  The entry now keeps its magnitude, the kernel uses |entry| as
    the plain per-row bound (exp2 arguments <= 90 whatever the data) and rows that underflow send their job (workspace) or head (none) to
    the online form: the result is the oracle's either way."""
    H, Sq, Sk = 4, 768, 1024
    g = torch.Generator().manual_seed(Sq + 7)
    def unit(x):
        return x / x.norm(dim=-1, keepdim=True)
    u = unit(torch.randn(H, 64, generator=g))
    q = torch.empty(Sq, H, 64); k = torch.empty(Sk, H, 64)
    k[:, 0] = 10.0 * unit(torch.randn(Sk, 64, generator=g));              q[:, 0] = 60.0 * unit(torch.randn(Sq, 64, generator=g))
    k[:, 1] = 6.0 * u[1] + 8.0 * unit(torch.randn(Sk, 64, generator=g));  q[:, 1] = 30.0 * u[1] + 40.0 * unit(torch.randn(Sq, 64, generator=g))
    k[:, 2] = 5.0 * unit(torch.randn(Sk, 64, generator=g));               q[:, 2] = 30.0 * unit(torch.randn(Sq, 64, generator=g))
    k[:, 3] = 1.5 * unit(torch.randn(Sk, 64, generator=g));               q[:, 3] = 10.0 * unit(torch.randn(Sq, 64, generator=g))
    q, k = bfr(q), bfr(k)
    v = bfr(torch.randn(Sk, H, 64, generator=g))
    qd, kd, vt = q.reshape(Sq, -1).cuda().to(BF), k.reshape(Sk, -1).cuda().to(BF), vt_of(v)
    ref = O.sdpa(q, k, v, "bf16", None, base2=True)
    for balanced in (True, False):
        flags, kmax, krad, centre = flags_rows_anchored(E, q, k, H)
        assert flags.tolist() == [1, 1, 1, 1] and (kmax[:2] < 0).all()
        out = run_rows_centred(E, qd, kd, vt, H, flags, kmax, centre, krad, balanced=balanced)       # marked heads, NO anchors
        assert torch.isfinite(out.float()).all()
        close(out, ref, ulps=4, atol=5e-3, what=f"marked heads without anchors (balanced={balanced})")
        flags, kmax, _, _ = flags_rows_anchored(E, q, k, H)
        out_p = run_rows(E, qd, kd, vt, H, flags, kmax)                                             # the plain per-row-offset entry point
        close(out_p, ref, ulps=4, atol=5e-3, what="marked heads through k5_attention_bf16_prescaled_rows")


def test_anchored_offsets_overflow_falls_back_per_job(E):
    """A key OUTSIDE every row's sample (index 20: not in a strided tile, not in the rows' own block) that a few query rows hit with a score
    ~ 500 above everything else: their anchors sit ~ 500 too low, exp2 overflows, the row sum is inf — the job (head 0, query block 1)
    falls back to the online form, every other job of the head stays on the fixed form, the head's flag stays 1, the output is the oracle's."""
    H, Sq, Sk = 2, 768, 1024
    g = torch.Generator().manual_seed(11)
    def unit(x):
        return x / x.norm(dim=-1, keepdim=True)
    d = unit(torch.randn(64, generator=g))
    k = 4.0 * unit(torch.randn(Sk, H, 64, generator=g))
    k[:, 0] -= (k[:, 0] @ d)[:, None] * d                                   # head 0: no key has a component along d ...
    k[20, 0] = 12.0 * d                                                     # ... except key 20
    q = 30.0 * unit(torch.randn(Sq, H, 64, generator=g))
    q[300:311, 0] += 45.0 * d                                               # rows 300..310 (query block 1) see key 20 at ~ +540
    q, k = bfr(q), bfr(k)
    v = bfr(torch.randn(Sk, H, 64, generator=g))
    s = q[300:311, 0] @ k[:, 0].t()
    assert (s[:, 20] - s[:, torch.arange(Sk) != 20].amax(-1) > 300).all()
    flags, kmax, krad, centre = flags_rows_anchored(E, q, k, H)
    assert flags.tolist() == [1, 1] and kmax[0].item() < 0, (flags, kmax)
    qd, kd, vt = q.reshape(Sq, -1).cuda().to(BF), k.reshape(Sk, -1).cuda().to(BF), vt_of(v)
    out, anchor = run_rows_anchored(E, qd, kd, vt, H, flags, kmax, centre, krad, balanced=True)
    assert flags.tolist() == [1, 1], flags                                  # a job fell back, not the head
    assert ((q[300:311, 0] @ k[:, 0].t()).amax(-1) - anchor[0, 300:311].cpu() > 300).all()   # the anchors really were too low
    ref = O.sdpa(q, k, v, "bf16", None, base2=True)
    close(out, ref, ulps=4, atol=5e-3, what="anchored offsets with one overflowing job")
    # without a workspace there are no job flags: the head itself is sent to the online form late (flag 0), same numbers
    flags2, kmax2, krad2, centre2 = flags_rows_anchored(E, q, k, H)
    out2, _ = run_rows_anchored(E, qd, kd, vt, H, flags2, kmax2, centre2, krad2, balanced=False)
    assert flags2.tolist() == [0, 1], flags2
    close(out2, ref, ulps=4, atol=5e-3, what="anchored offsets, head-level fallback")
    # a head whose jobs kept falling back is not anchored again
    prefer = torch.tensor([1, 0], dtype=torch.int32, device="cuda")
    flags3, kmax3, _, _ = flags_rows_anchored(E, q, k, H, prefer=prefer)
    assert flags3.tolist() == [0, 1] and kmax3[0].item() > 0, (flags3, kmax3)


def test_config2_size_anchored_offsets_sampled_rows_vs_oracle(E):
    """BASELINE config-2 attention shape (47 616 tokens x 28 heads) at QK-norm gains beyond the Cauchy-Schwarz window — RMS-normalised q / k
    like the engine's, heads 0..13 at gain 6 (bound 415), 14..27 at gain 7 (565): what `bench.py --qk-gain 6 / 7` runs.  Every head is
    marked for anchored offsets, stays on the fixed form through the balanced launch (a job may fall back: counted through the
    constant-V check — an output row is exactly the constant either way) and sampled rows match the oracle."""
    N, H = 47616, 28
    g = torch.Generator(device="cuda").manual_seed(0)
    gains = torch.cat([torch.full((14,), 6.0), torch.full((14,), 7.0)]).cuda()
    def rmsn(x):
        return gains[None, :, None] * x / x.pow(2).mean(-1, keepdim=True).sqrt()
    q = rmsn(torch.randn(N, H, 64, device="cuda", generator=g)).reshape(N, -1).to(BF)
    kc = (rmsn(torch.randn(N, H, 64, device="cuda", generator=g)) * O.SOFTMAX_C).reshape(N, -1).to(BF)
    v = torch.randn(N, H * 64, device="cuda", generator=g).to(BF)
    vt = v.t().contiguous()
    qf, kf = q.float().reshape(N, H, 64), kc.float().reshape(N, H, 64)
    flags, kmax, krad, centre = flags_rows_anchored(E, qf.cpu(), kf.cpu(), H)
    assert flags.tolist() == [1] * H and (kmax < 0).all(), (flags, kmax)
    out, anchor = run_rows_anchored(E, q, kc, vt, H, flags, kmax, centre, krad, balanced=True)
    assert flags.tolist() == [1] * H, flags                                     # no head left the fixed form
    rows = torch.tensor([0, 1, 31, 255, 256, 4097, 23808, 40000, 47104, 47615 - 64, 47615])   # incl. rows of the split tail jobs
    ref = O.sdpa(qf[rows].cpu(), kf.cpu(), v.float().cpu().reshape(N, H, 64), "bf16", None, base2=True)
    close(out[rows], ref, ulps=4, atol=5e-3, what="config-2 sampled rows, anchored offsets")
    smax = torch.stack([(qf[rows, h] @ kf[:, h].t()).amax(-1) for h in range(H)]).cpu()   # the rows' true maxima: inside the exact range
    assert (smax - anchor[:, rows].cpu() < 112).all() and (smax - anchor[:, rows].cpu() > -81).all()
    oc, _ = run_rows_anchored(E, q, kc, torch.full_like(vt, 0.75), H, flags, kmax, centre, krad, balanced=True)
    assert (oc.float() - 0.75).abs().max().item() <= 2 ** -8
