"""Round 6: the aligned split-K cut on SUB-ROUND launches of the four-wave GEMM (fewer tiles than CUs: BASELINE config 1's 3328 tokens, 8-GPU token
shards) — whole tiles (kernel id 4) against the cut (id 24) at every token-tile height, ~0.3 s of back-to-back launches per variant, orders alternating.

    python tools/experiments/r06_sub_round_split.py [--tokens 3328,5952]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch  # noqa: E402
from kandinsky import _engine as E  # noqa: E402

BF = torch.bfloat16
D, FF = 1792, 7168


def timeit(fn, seconds=0.25):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record(); torch.cuda.synchronize()
    iters = max(20, int(seconds * 1e3 / (a.elapsed_time(b) / 20)))
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", default="3328,5952")
    args = ap.parse_args()
    torch.manual_seed(0)
    for M in [int(t) for t in args.tokens.split(",")]:
        shapes = [("q|k", M, 2 * D, D, E.EPI_BIAS), ("V^T", D, M, D, E.EPI_BIAS_M), ("out+gate", M, D, D, E.EPI_GATE), ("cross q", M, D, D, E.EPI_BIAS),
                  ("FF1+GELU", M, FF, D, E.EPI_GELU), ("FF2+gate", M, D, FF, E.EPI_GATE)]
        for name, m, n, k, epi in shapes:
            a, w = torch.randn(m, k, device="cuda").to(BF), (torch.randn(n, k, device="cuda") * 0.05).to(BF)
            bias = torch.randn(m if epi == E.EPI_BIAS_M else n, device="cuda").to(BF).float()
            resid = torch.randn(m, n, device="cuda").to(BF) if epi == E.EPI_GATE else None
            gate = torch.randn(n, device="cuda") if epi == E.EPI_GATE else None
            out = torch.empty(m, n, dtype=BF, device="cuda")
            ref = None
            res = {}
            for rep in range(2):
                order = [(kern, mt) for mt in (4, 6, 8) for kern in (4, 24)]
                if rep:
                    order.reverse()
                for kern, mt in order:
                    fn = lambda: E.gemm(a, w, bias, epi, resid=resid, gate=gate, out=out, kernel=kern, token_tile=32 * mt)   # noqa: E731
                    fn()
                    if epi != E.EPI_GATE:
                        if ref is None:
                            ref = out.clone()
                        d = (out != ref).float().mean().item()
                        assert d < 2e-3, (name, kern, mt, d)
                    res.setdefault((kern, mt), []).append(timeit(fn))
            default = timeit(lambda: E.gemm(a, w, bias, epi, resid=resid, gate=gate, out=out))
            tiles = {mt: ((m + 32 * mt - 1) // (32 * mt)) * ((n + 255) // 256) for mt in (4, 6, 8)}
            line = "  ".join(f"MT{mt} ({tiles[mt]:3d} tiles) whole {min(res[(4, mt)]):6.1f} cut {min(res[(24, mt)]):6.1f}" for mt in (4, 6, 8))
            print(f"M={m:5d} N={n:5d} K={k:5d} {name:9s}: default {default:6.1f} us | {line}", flush=True)


if __name__ == "__main__":
    main()
