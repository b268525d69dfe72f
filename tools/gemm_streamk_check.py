"""Split-K tail of the four-wave GEMM (kernel id 24 = wherever a tile can be cut, 14 = by policy) against the whole-tile schedule (id 4) and an fp32
reference: every epilogue, the model's shapes + ragged ones; fraction of outputs that differ, errors against fp32, microseconds.

    python tools/gemm_streamk_check.py [--iters 20]

CAUTION about the times: each shape's buffers are new, kernel 4 is timed first and pays the warm-up — 20 iterations in that order showed the split
tail 6.6 % ahead per block (profiles/r06_split_tail_slices.log); a second of back-to-back launches in both orders shows 0 +- 0.6 %
(tools/experiments/r06_split_tail_sustained.py, profiles/r06_split_tail_sustained.log).  Use --iters 1000 or that script for timing; this one is the
correctness check."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch  # noqa: E402
from kandinsky import _engine as E  # noqa: E402

BF = torch.bfloat16
D, FF = 1792, 7168


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--tokens", default="47616,23808,11904,93696")
    args = ap.parse_args()
    torch.manual_seed(0)
    bad = 0
    for M in [int(t) for t in args.tokens.split(",")]:
        shapes = [("q|k", M, 2 * D, D, E.EPI_BIAS), ("V^T", D, M, D, E.EPI_BIAS_M), ("out+gate", M, D, D, E.EPI_GATE),
                  ("FF1+GELU", M, FF, D, E.EPI_GELU), ("FF2+gate", M, D, FF, E.EPI_GATE)]
        if M == 47616:
            shapes += [("ragged", 40000 + 8, 1792 + 256 + 8, 1792, E.EPI_BIAS), ("ragged gate", 33333, 2048, 3584, E.EPI_GATE)]
        tot = {4: 0.0, 24: 0.0}
        for name, m, n, k, epi in shapes:
            a, w = torch.randn(m, k, device="cuda").to(BF), (torch.randn(n, k, device="cuda") * 0.05).to(BF)
            bias = torch.randn(m if epi == E.EPI_BIAS_M else n, device="cuda").to(BF).float()
            resid0 = torch.randn(m, n, device="cuda").to(BF) if epi == E.EPI_GATE else None
            gate = torch.randn(n, device="cuda") if epi == E.EPI_GATE else None
            outs, us = {}, {}
            for kern in (4, 24):
                out = torch.empty(m, n, dtype=BF, device="cuda")
                resid = resid0.clone() if resid0 is not None else None
                E.gemm(a, w, bias, epi, resid=resid, gate=gate, out=out, kernel=kern)
                outs[kern] = out.clone()
                # repeat once more: the flags must have been left clean
                resid = resid0.clone() if resid0 is not None else None
                E.gemm(a, w, bias, epi, resid=resid, gate=gate, out=out, kernel=kern)
                assert torch.equal(out, outs[kern]), f"{name} kernel {kern}: not repeatable"
                us[kern] = timeit(lambda: E.gemm(a, w, bias, epi, resid=resid, gate=gate, out=out, kernel=kern), args.iters)
                tot[kern] += us[kern] if len(name) <= 9 and "ragged" not in name else 0.0
            # both schedules against an fp32 evaluation with the engine's rounding points (bf16(acc + bias) -> GELU / gated residual -> bf16): the split
            # sum may flip the bf16 rounding of a pre-epilogue value by one ulp here and there — it must not be WORSE than the whole-tile schedule
            rows = torch.randperm(m, device="cuda")[:2048]
            y = a[rows].float() @ w.float().t()
            y = (y + (bias[rows][:, None] if epi == E.EPI_BIAS_M else bias[None, :])).to(BF).float()
            if epi == E.EPI_GELU:
                y = torch.nn.functional.gelu(y)
            if epi == E.EPI_GATE:
                y = resid0[rows].float() + gate[None, :] * y
            e4, e24 = (outs[4][rows].float() - y).abs(), (outs[24][rows].float() - y).abs()
            frac = (outs[4] != outs[24]).float().mean().item()
            ok = frac < 2e-3 and e24.max().item() <= 1.25 * e4.max().item() + 1e-6 and e24.mean().item() <= 1.02 * e4.mean().item() + 1e-9
            bad += not ok
            print(f"M={m:6d} N={n:5d} K={k:5d} {name:12s}: whole tiles {us[4]:8.1f} us  split tail {us[24]:8.1f} us ({us[24] / us[4] - 1:+.1%})  "
                  f"outputs that differ {frac:.2e}; |error| vs fp32 max {e4.max().item():.3e} / {e24.max().item():.3e}, mean {e4.mean().item():.3e} / {e24.mean().item():.3e}  {'ok' if ok else 'MISMATCH'}", flush=True)
        print(f"  block of five at {M} rows: {tot[4]:.1f} -> {tot[24]:.1f} us ({tot[24] / tot[4] - 1:+.1%})", flush=True)
    print("FAILED" if bad else "all ok")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
