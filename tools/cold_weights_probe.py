"""How much of a small-N GEMM's time is its weights coming from HBM instead of the 256-MB Infinity Cache?  One visual block's seven projections at a
token count, each timed alone (HIP events) (a) warm: the same weights again and again, (b) cold: 1 GB of other data streamed through the cache before
every call (what the engine sees: 4 GB of weights per forward, every GEMM meets its weights for the first time since the last step), (c) cold + the
weights read once by a trivial kernel just before the call (what a prefetch on a side stream would leave behind).
    python tools/cold_weights_probe.py [tokens]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch
from kandinsky import _engine as E
BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 3328
D, FF = 1792, 7168
shapes = [("q|k", M, 2 * D, D, E.EPI_BIAS), ("V^T", D, M, D, E.EPI_BIAS_M), ("out+gate", M, D, D, E.EPI_GATE), ("xq", M, D, D, E.EPI_BIAS),
          ("FF1+GELU", M, FF, D, E.EPI_GELU), ("FF2+gate", M, D, FF, E.EPI_GATE)]
junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
tot = {"warm": 0.0, "cold": 0.0, "cold+touch": 0.0}
for name, m, n, k, epi in shapes:
    a = torch.randn(m, k, device="cuda").to(BF); w = (torch.randn(n, k, device="cuda") * 0.05).to(BF)
    bias = torch.randn(m if epi == E.EPI_BIAS_M else n, device="cuda").to(BF).float()
    gate = torch.randn(n, device="cuda") if epi == E.EPI_GATE else None
    out = torch.randn(m, n, device="cuda").to(BF)
    wt = w if epi != E.EPI_BIAS_M else a          # the operand that is a WEIGHT in the engine (V^T: the weight is the A operand)
    call = lambda: E.gemm(a, w, bias, epi, resid=out if epi == E.EPI_GATE else None, gate=gate, out=out)
    res = {}
    for mode in ("warm", "cold", "cold+touch"):
        ts = []
        for it in range(12):
            if mode != "warm":
                junk.add_(1)                      # 1 GB read + 1 GB written: nothing of this GEMM is left in the 256-MB cache
            if mode == "cold+touch":
                wt.view(torch.int16).sum()        # read the weights once (a prefetch would do this on another stream, under the previous kernel)
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); call(); t.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(t) * 1e3)
        res[mode] = sorted(ts)[len(ts) // 2]
        tot[mode] += res[mode] * (2 if name in ("out+gate",) else 1)
    print(f"M={m:6d} {name:9s}: warm {res['warm']:7.1f}  cold {res['cold']:7.1f}  cold + weights touched {res['cold+touch']:7.1f} us", flush=True)
print(f"block (out+gate counted twice for the cross out projection): warm {tot['warm']:.0f}  cold {tot['cold']:.0f}  cold + touched {tot['cold+touch']:.0f} us")
