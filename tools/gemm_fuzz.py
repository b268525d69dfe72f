"""Random-shape sweep of the GEMM entry against torch (fp32 matmul on the GPU, bf16-rounded) — every epilogue, ragged M / N,
K-tile counts from the 4-wave kernel's minimum up.  python tools/gemm_fuzz.py [n_cases] [seed]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch
from kandinsky import _engine as E
BF = torch.bfloat16
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.manual_seed(0)
bad = 0
for case in range(n_cases):
    tiles = rng.choice([130, 200, 256, 300, 520, 700])
    tm = rng.randint(2, max(2, tiles // 2)); tn = max(1, tiles // tm)
    M = tm * 256 - rng.choice([0, 0, 8, 40, 100, 255])
    N = tn * 256 - rng.choice([0, 0, 8, 24, 64, 248])
    K = 128 * rng.choice([2, 3, 4, 6, 14]) if rng.random() < 0.8 else 64 * rng.choice([3, 5, 9])
    epi = rng.choice(["bias", "nobias", "gelu", "gate", "bias_m"])
    a = (torch.randn(M, K, device="cuda")).to(BF); w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    acc = a.float() @ w.float().t()
    rnd = lambda x: x.to(BF).float()
    if epi == "bias":
        b = rnd(torch.randn(N, device="cuda") * 0.1); ref = rnd(acc + b); got = E.gemm(a, w, b, E.EPI_BIAS)
    elif epi == "nobias":
        ref = rnd(acc); got = E.gemm(a, w, None, E.EPI_BIAS)
    elif epi == "gelu":
        ref = rnd(torch.nn.functional.gelu(rnd(acc))); got = E.gemm(a, w, None, E.EPI_GELU)
    elif epi == "gate":
        resid = rnd(torch.randn(M, N, device="cuda")); gate = torch.randn(N, device="cuda")
        inner = rnd(acc); ref = rnd(resid + gate * inner); r = resid.to(BF)
        got = E.gemm(a, w, None, E.EPI_GATE, resid=r, gate=gate, out=r)
    else:
        bm = rnd(torch.randn(M, device="cuda")); ref = rnd(acc + bm[:, None]); got = E.gemm(a, w, bm, E.EPI_BIAS_M)
    err = (got.float() - ref).abs()
    tol = 2e-3 + 3 * 2.0 ** -7 * ref.abs() + (gate.abs() * 2.0 ** -7 * inner.abs() if epi == "gate" else 0)
    nbad = int((err > tol).sum())
    print(f"case {case:2d} M={M} N={N} K={K} {epi:7s} tiles={((M+255)//256)*((N+255)//256):4d}: max err {err.max().item():.4g} off {nbad}", flush=True)
    bad += nbad > 0
print("FAILED" if bad else "all ok", bad)
sys.exit(1 if bad else 0)
