"""W8A8 e4m3 GEMM on the model's feed-forward (and projection) shapes through the C ABI: TFLOP/s per shape and epilogue.
K5_GEMM_FP8_V=8 forces the 8-wave kernel of rounds 1-3, 4 the four-wave one (default: four-wave from 256 tiles up)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch
from kandinsky import _engine as E
L = E.lib()
N_TOK = 47616
SHAPES = [("FF1 + GELU (fp8 out)", N_TOK, 7168, 1792, E.EPI_GELU), ("FF2 + gate", N_TOK, 1792, 7168, E.EPI_GATE),
          ("q|k shape, plain", N_TOK, 3584, 1792, E.EPI_BIAS), ("out shape, gate", N_TOK, 1792, 1792, E.EPI_GATE), ("4096 x 4096 x 32768", 4096, 4096, 32768, E.EPI_BIAS)]
for name, M, N, K, epi in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    a8 = (torch.randn(M, K, device="cuda", generator=g)).to(torch.float8_e4m3fn).view(torch.uint8)
    w8 = (torch.randn(N, K, device="cuda", generator=g)).to(torch.float8_e4m3fn).view(torch.uint8)
    ws = torch.rand(N, device="cuda") * 0.01 + 0.001
    out = torch.zeros(M, N, dtype=torch.uint8 if epi == E.EPI_GELU else torch.bfloat16, device="cuda")
    gate = torch.randn(N, device="cuda")
    args = (a8.data_ptr(), w8.data_ptr(), ws.data_ptr(), out.data_ptr(), M, N, K, K, K, N, epi, out.data_ptr() if epi == E.EPI_GATE else None,
            N if epi == E.EPI_GATE else 0, gate.data_ptr() if epi == E.EPI_GATE else None, E.stream_ptr())
    for _ in range(3):
        E.check(L.k5_gemm_fp8(*args))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        E.check(L.k5_gemm_fp8(*args))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"K5_GEMM_FP8_V={os.environ.get('K5_GEMM_FP8_V', '-')} {name:24s} {M}x{N}x{K}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.0f} TFLOP/s")
