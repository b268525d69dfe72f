"""Which shader clock do the hot kernels run at?  A one-wave monitor kernel (tools/probes/clock_monitor.hip) samples the shader-cycle counter
against the 100-MHz wall clock on a side stream while the library's kernels run on the main stream: idle, the dense attention launch of the
bench (47 616 tokens, 28 heads, random RMS-normalised q / k), the five GEMM shapes of a block, the same on ZERO operands.

    python tools/clock_under_load.py          (builds tools/probes/libclockmon.so on first use)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch  # noqa: E402
from kandinsky import _engine as E  # noqa: E402

SO = os.path.join(ROOT, "tools", "probes", "libclockmon.so")
if not os.path.exists(SO):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", SO, os.path.join(ROOT, "tools", "probes", "clock_monitor.hip")])
mon = ctypes.CDLL(SO)
mon.clock_monitor_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
BF = torch.bfloat16
side = torch.cuda.Stream()


def measure(name, work, ms_budget=60.0, flop=None):
    """run `work()` repeatedly for ~ms_budget under the monitor; report the clock over the middle 80 % of the samples"""
    period = 2000                       # 20 us
    n = int(ms_budget * 1e3 / 20)
    buf = torch.zeros(2 * n, dtype=torch.int64, device="cuda")
    work(); work()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        mon.clock_monitor_launch(buf.data_ptr(), n, period, side.cuda_stream)
    a.record()
    reps = 0
    import time
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms_budget * 0.9:
        work()
        reps += 1
        if reps % 4 == 0:
            torch.cuda.current_stream().synchronize()
    b.record()
    torch.cuda.synchronize()
    s = buf.cpu().reshape(n, 2).double()
    lo, hi = n // 10, n - n // 10
    ghz = (s[hi, 1] - s[lo, 1]) / ((s[hi, 0] - s[lo, 0]) * 10.0)
    seg = (s[lo + 50:hi:50, 1] - s[lo:hi - 50:50, 1]) / ((s[lo + 50:hi:50, 0] - s[lo:hi - 50:50, 0]) * 10.0)     # per-millisecond clocks
    ms = a.elapsed_time(b) / max(reps, 1)
    extra = f"; {ms:8.3f} ms per call" + (f" = {flop / ms / 1e9:6.0f} TFLOP/s = {flop / ms / 1e9 / (ghz / 2.4 * 2500) :.3f} of the MFMA peak AT THIS CLOCK ({ghz / 2.4 * 2500:.0f})" if flop else "") if reps else ""
    print(f"{name:58s} shader clock {ghz:.3f} GHz (per-ms min {seg.min():.3f} max {seg.max():.3f}){extra}", flush=True)
    return float(ghz)


def gemm_case(M, N, K, epi, zero=False):
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.zeros(M, K, dtype=BF, device="cuda") if zero else torch.randn(M, K, device="cuda", generator=g).to(BF)
    w = torch.zeros(N, K, dtype=BF, device="cuda") if zero else (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(BF)
    out = torch.zeros(M, N, dtype=BF, device="cuda")
    bias, gate = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
    kw = {"bias": dict(bias=bias, epilogue=E.EPI_BIAS), "gelu": dict(bias=None, epilogue=E.EPI_GELU),
          "gate": dict(bias=bias, epilogue=E.EPI_GATE, resid=out, gate=gate)}[epi]
    return lambda: E.gemm(a, w, out=out, **kw)


def attn_case(zero=False):
    N, H = 47616, 28
    def rmsn(x):
        x = x.float().reshape(N, H, 64)
        return (x / x.pow(2).mean(-1, keepdim=True).sqrt()).reshape(N, H * 64).to(BF)
    if zero:
        q = k = torch.zeros(N, H * 64, dtype=BF, device="cuda")
        vt = torch.zeros(H * 64, N, dtype=BF, device="cuda")
    else:
        q, k = rmsn(torch.randn(N, H * 64, device="cuda")), rmsn(torch.randn(N, H * 64, device="cuda"))
        vt = torch.randn(H * 64, N, device="cuda").to(BF)
    kc = (k.float() * 0.18033688011112042).to(BF)
    o = torch.empty(N, H * 64, dtype=BF, device="cuda")
    return lambda: E.check(E.lib().k5_attention_bf16_prescaled(q.data_ptr(), kc.data_ptr(), vt.data_ptr(), o.data_ptr(), H, N, N, q.stride(0), kc.stride(0),
                                                                vt.stride(0), o.stride(0), 64 * 1.05, E.stream_ptr()))


def main():
    N = 47616
    if "--attention" in sys.argv:     # the attention launch only (A/B of kernel variants through the environment, e.g. K5_ATTN_WAVE_ROWS=64)
        measure(f"dense attention, random data, K5_ATTN_WAVE_ROWS={os.environ.get('K5_ATTN_WAVE_ROWS', 'default')}", attn_case(), 200.0, 4.0 * N * N * 64 * 28)
        return
    if "--blaslt" in sys.argv:        # the vendor library's GEMM (torch.matmul -> hipBLASLt) on the model's shapes under the same monitor: is it granted a higher clock?
        for name, (M, Nn, K) in {"q|k": (N, 3584, 1792), "FF1": (N, 7168, 1792), "FF2": (N, 1792, 7168), "4096 x 4096 x 32768": (4096, 4096, 32768)}.items():
            g = torch.Generator(device="cuda").manual_seed(0)
            a = torch.randn(M, K, device="cuda", generator=g).to(BF)
            w = (torch.randn(Nn, K, device="cuda", generator=g) * 0.05).to(BF)
            out = torch.empty(M, Nn, dtype=BF, device="cuda")
            measure(f"hipBLASLt (torch.matmul) {name} {M}x{Nn}x{K}, random data", lambda: torch.matmul(a, w.t(), out=out), 60.0, 2.0 * M * Nn * K)
            bias = torch.randn(Nn, device="cuda")
            measure(f"libk5 w4 (bias epilogue)  {name} {M}x{Nn}x{K}, random data", lambda: E.gemm(a, w, out=out, bias=bias, epilogue=E.EPI_BIAS), 60.0, 2.0 * M * Nn * K)
        return
    measure("idle (nothing on the main stream)", lambda: None, 20.0)
    measure("dense attention, 47 616 tokens x 28 heads, random data", attn_case(), 150.0, 4.0 * N * N * 64 * 28)
    measure("dense attention, zero operands", attn_case(True), 150.0, 4.0 * N * N * 64 * 28)
    for name, (M, Nn, K, epi) in {"q|k (bias)": (N, 3584, 1792, "bias"), "out + gate": (N, 1792, 1792, "gate"), "FF1 + GELU": (N, 7168, 1792, "gelu"),
                                  "FF2 + gate": (N, 1792, 7168, "gate"), "4096 x 4096 x 32768 (bias)": (4096, 4096, 32768, "bias")}.items():
        measure(f"GEMM {name} {M}x{Nn}x{K}, random data", gemm_case(M, Nn, K, epi), 60.0, 2.0 * M * Nn * K)
    measure("GEMM FF2 + gate, zero operands", gemm_case(N, 1792, 7168, "gate", True), 60.0, 2.0 * N * 1792 * 7168)


if __name__ == "__main__":
    main()
