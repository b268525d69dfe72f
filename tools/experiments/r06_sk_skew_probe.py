"""Is the stream-K region slow because of the K skew between workgroups?  N = 2048 (8 n-tiles), K = 1792 / 7168, M swept so that the tiles go from
exactly 2 rounds (no split tile, no skew) through small remainders to 3 rounds; whole-tile schedule (kernel 4) against the stream-K walk forced on (34)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "kandinsky-5_amd"))
import torch
from kandinsky import _engine as E
BF = torch.bfloat16
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for K in (1792, 7168):
    for M in (16384, 16640, 17408, 18432, 20480, 22528, 24320, 24576):
        N = 2048
        a, w = torch.randn(M, K, device="cuda").to(BF), (torch.randn(N, K, device="cuda") * 0.05).to(BF)
        bias = torch.randn(N, device="cuda").to(BF).float()
        out = torch.empty(M, N, dtype=BF, device="cuda")
        t4 = timeit(lambda: E.gemm(a, w, bias, E.EPI_BIAS, out=out, kernel=4))
        t34 = timeit(lambda: E.gemm(a, w, bias, E.EPI_BIAS, out=out, kernel=34))
        tiles = (M + 255) // 256 * 8
        print(f"K={K} M={M} tiles {tiles} = {tiles / 256:.3f} rounds: whole tiles {t4:7.1f} us ({t4 / (tiles / 256):6.1f} per round of work), stream-K walk {t34:7.1f} us ({t34 / (tiles / 256):6.1f})", flush=True)
