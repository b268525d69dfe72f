"""Why does the split-K tail win 6.6 % per block in a 20-iteration microbenchmark and nothing through the engine?  The same launches, timed over 20
iterations (a few ms: boost clock) and over ~1 s of back-to-back launches (the power-limited regime the sampler runs in), whole tiles (kernel 4)
against the split tail (24), interleaved."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "kandinsky-5_amd"))
import torch
from kandinsky import _engine as E
BF = torch.bfloat16
D, FF = 1792, 7168
def timeit(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
M = 47616
for name, m, n, k, epi in (("out+gate", M, D, D, E.EPI_GATE), ("q|k", M, 2 * D, D, E.EPI_BIAS), ("FF2+gate", M, D, FF, E.EPI_GATE)):
    a, w = torch.randn(m, k, device="cuda").to(BF), (torch.randn(n, k, device="cuda") * 0.05).to(BF)
    bias = torch.randn(n, device="cuda").to(BF).float()
    resid = torch.randn(m, n, device="cuda").to(BF) if epi == E.EPI_GATE else None
    gate = torch.randn(n, device="cuda") if epi == E.EPI_GATE else None
    out = torch.empty(m, n, dtype=BF, device="cuda")
    for iters in (20, 3000 if k == D else 1000):
        for rep in range(2):
            t = {kern: timeit(lambda: E.gemm(a, w, bias, epi, resid=resid, gate=gate, out=out, kernel=kern), iters) for kern in (4, 24)}
            print(f"{name:9s} {iters:5d} launches back to back: whole tiles {t[4]:7.1f} us, split tail {t[24]:7.1f} us ({t[24] / t[4] - 1:+.1%})", flush=True)
