"""would a weight-shared ("batched") cond + uncond pass pay?  BASELINE config 3 says "cond+uncond batched"; the reference
runs the two forwards one after the other (generation_utils.py:53-73).  The two branches differ from the first AdaLN on (the pooled text
embedding enters the time embedding), so batching = stacking 2 N rows through the linear layers; the attention stays per branch.  What the
stacking could buy is the weight traffic and the ragged last rounds: measured here as one 2N-row GEMM against two N-row GEMMs per shape.

    python tools/batched_cfg_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch
from kandinsky import _engine as E
N, D, FF = 47616, 1792, 7168
SH = [("q|k", 2 * D, D, E.EPI_BIAS), ("out+gate", D, D, E.EPI_GATE), ("FF1+GELU", FF, D, E.EPI_GELU), ("FF2+gate", D, FF, E.EPI_GATE)]
tot1 = tot2 = 0.0
for name, n, k, epi in SH:
    g = torch.Generator(device="cuda").manual_seed(0)
    a2 = torch.randn(2 * N, k, device="cuda", generator=g).bfloat16()
    w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).bfloat16()
    bias, gate = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
    out2 = torch.zeros(2 * N, n, dtype=torch.bfloat16, device="cuda")
    kw = dict(bias=None if epi == E.EPI_GELU else bias, epilogue=epi)
    def run(a, o):
        if epi == E.EPI_GATE:
            E.gemm(a, w, out=o, resid=o, gate=gate, **kw)
        else:
            E.gemm(a, w, out=o, **kw)
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10
    best1 = best2 = 1e9
    for rep in range(3):    # interleaved
        best2 = min(best2, t(lambda: (run(a2[:N], out2[:N]), run(a2[N:], out2[N:]))))
        best1 = min(best1, t(lambda: run(a2, out2)))
    tot1 += best1; tot2 += best2
    print(f"{name:9s} two N-row launches {best2 * 1e3:8.1f} us   one 2N-row launch {best1 * 1e3:8.1f} us   saving {100 * (1 - best1 / best2):5.2f} %")
print(f"sum of the four shapes: {tot2 * 1e3:.1f} -> {tot1 * 1e3:.1f} us per block and CFG step = {32 * (tot2 - tot1):.2f} ms per step of ~1060 ({100 * 32 * (tot2 - tot1) / 1060:.2f} %)")
