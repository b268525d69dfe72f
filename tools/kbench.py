"""Kernel microbenchmarks through the C ABI at BASELINE config-2 shapes (N = 47 616 tokens, D = 1792).

    python tools/kbench.py [attn] [gemm] [elem]

Prints TFLOP/s (MFMA kernels) or GB/s (HBM kernels) per kernel from HIP-event timing on the launch stream.
Random (not zero) operands — zero-filled data inflates clocks (guide rule 25)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch  # noqa: E402
from kandinsky import _engine as E  # noqa: E402

BF = torch.bfloat16
N, D, H, FF = 47616, 1792, 28, 7168


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def rnd(*s):
    return torch.randn(*s, device="cuda").to(BF)


def bench_attn():
    for (sq, sk, h) in ((N, N, H), (N, 256, H), (3328, 3328, H), (N, 4096, H), (N, 16384, H)):
        q, k = rnd(sq, h * 64), rnd(sk, h * 64)
        vt = rnd(h * 64, (sk + 7) // 8 * 8)
        o = torch.empty(sq, h * 64, dtype=BF, device="cuda")
        ms = timeit(lambda: E.attention(q, k, vt, h, out=o), iters=3, warm=1)
        fl = 4.0 * sq * sk * 64 * h
        print(f"attention Sq={sq} Sk={sk} H={h}: {ms:9.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s", flush=True)


def bench_gemm():
    for (m, n, k, epi, name) in ((N, 2 * D, D, E.EPI_BIAS, "qk"), (D, N, D, E.EPI_BIAS_M, "v^T"), (N, D, D, E.EPI_GATE, "out+gate"),
                                 (N, FF, D, E.EPI_GELU, "ff1+gelu"), (N, D, FF, E.EPI_GATE, "ff2+gate"),
                                 (4096, 4096, 4096, E.EPI_BIAS, "4096^3")):
        a, w = rnd(m, k), rnd(n, k) * 0.05
        bias = torch.randn(m if epi == E.EPI_BIAS_M else n, device="cuda")
        out = torch.empty(m, n, dtype=BF, device="cuda")
        resid = rnd(m, n) if epi == E.EPI_GATE else None
        gate = torch.randn(n, device="cuda") if epi == E.EPI_GATE else None
        ms = timeit(lambda: E.gemm(a, w, bias, epi, resid=resid, gate=gate, out=out))
        print(f"gemm {name:9s} M={m} N={n} K={k}: {ms:8.3f} ms  {2.0 * m * n * k / ms / 1e9:8.1f} TFLOP/s", flush=True)


def bench_gemm_sp():
    """the same projections on an 8-GPU / 4-GPU / 2-GPU token shard"""
    for P in (8, 4, 2):
        n = N // P
        for (m, nn, k, epi, name) in ((n, D, D, E.EPI_BIAS, f"k/q P={P}"), (D, n, D, E.EPI_BIAS_M, f"v^T P={P}"),
                                      (n, FF, D, E.EPI_GELU, f"ff1 P={P}"), (n, D, FF, E.EPI_GATE, f"ff2 P={P}")):
            a, w = rnd(m, k), rnd(nn, k) * 0.05
            bias = torch.randn(m if epi == E.EPI_BIAS_M else nn, device="cuda")
            out = torch.empty(m, nn, dtype=BF, device="cuda")
            resid = rnd(m, nn) if epi == E.EPI_GATE else None
            gate = torch.randn(nn, device="cuda") if epi == E.EPI_GATE else None
            ms = timeit(lambda: E.gemm(a, w, bias, epi, resid=resid, gate=gate, out=out), iters=10)
            print(f"gemm {name:10s} M={m} N={nn} K={k}: {ms:8.3f} ms  {2.0 * m * nn * k / ms / 1e9:8.1f} TFLOP/s", flush=True)


def bench_gemm_fp8():
    L = E.lib()
    for (m, n, k, epi, name) in ((N, FF, D, E.EPI_GELU, "ff1+gelu fp8"), (N, D, FF, E.EPI_GATE, "ff2+gate fp8"), (4096, 4096, 4096, E.EPI_BIAS, "4096^3 fp8")):
        a = torch.randint(0, 255, (m, k), dtype=torch.uint8, device="cuda") & 0x77    # finite e4m3 patterns of both signs
        a |= torch.randint(0, 2, (m, k), dtype=torch.uint8, device="cuda") << 7
        w = torch.randint(0, 255, (n, k), dtype=torch.uint8, device="cuda") & 0x77
        sc = torch.full((n,), 1e-3, device="cuda")
        out = torch.empty(m, n, dtype=torch.uint8 if epi == E.EPI_GELU else BF, device="cuda")
        resid = rnd(m, n) if epi == E.EPI_GATE else None
        gate = torch.randn(n, device="cuda") if epi == E.EPI_GATE else None
        def run():
            E.check(L.k5_gemm_fp8(a.data_ptr(), w.data_ptr(), sc.data_ptr(), out.data_ptr(), m, n, k, k, k, n, epi,
                                  resid.data_ptr() if resid is not None else None, n, gate.data_ptr() if gate is not None else None, E.stream_ptr()))
        ms = timeit(run)
        print(f"gemm {name:13s} M={m} N={n} K={k}: {ms:8.3f} ms  {2.0 * m * n * k / ms / 1e9:8.1f} TFLOP/s", flush=True)


def bench_elem():
    x = rnd(N, D)
    sc, sh = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
    ms = timeit(lambda: E.ln_modulate(x, sc, sh))
    print(f"ln_modulate [{N}x{D}]: {ms:7.3f} ms  {2 * N * D * 2 / ms / 1e6:8.1f} GB/s", flush=True)
    qk = rnd(N, 2 * D)
    w = torch.ones(128, device="cuda")
    cos, sin = torch.randn(N, 32, device="cuda"), torch.randn(N, 32, device="cuda")
    ms = timeit(lambda: E.rmsnorm_rope_(qk, w, cos, sin, heads=2 * H, heads_per_weight=H, rope_heads=2 * H))
    print(f"rmsnorm_rope [{N}x{2 * D}]: {ms:7.3f} ms  {2 * N * 2 * D * 2 / ms / 1e6:8.1f} GB/s", flush=True)
    wm = torch.randn(32 * 9 * D, 512, device="cuda")
    t = torch.randn(512, device="cuda")
    ms = timeit(lambda: E.gemv_f32(t, wm, None, silu_in=True))
    print(f"gemv modulation [{wm.shape[0]}x512] fp32: {ms:7.3f} ms  {wm.numel() * 4 / ms / 1e6:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["attn", "gemm", "elem"]
    print(torch.cuda.get_device_name(0), flush=True)
    if "attn" in which:
        bench_attn()
    if "gemm_fp8" in which:
        bench_gemm_fp8()
    if "gemm_sp" in which:
        bench_gemm_sp()
    if "gemm" in which:
        bench_gemm()
    if "elem" in which:
        bench_elem()
