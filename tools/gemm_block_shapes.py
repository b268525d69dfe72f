"""The seven projections of one visual block with their real epilogues, at the token counts of one GPU and of 2/4/8-GPU token shards and
of BASELINE config 1, microseconds per call and per block, for each kernel the dispatch can pick:

    python tools/gemm_block_shapes.py [--tokens 47616,11904,5952,3328] [--kernels auto,4,8] [--iters 20]

(K5_GEMM_V1 forces a kernel: 4 = four-wave persistent, 8 = eight-wave ping-pong, 2 = 128x128, 5 = quadrant kernel; K5_GEMM_SK=0/1 the
stream-K schedule.)  One subprocess per kernel setting because the switches are read once per process.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D, FF = 1792, 7168


def shapes(M):
    # (name, M, N, K, epilogue)
    return [("q|k", M, 2 * D, D, "bias"), ("V^T", D, M, D, "bias_m"), ("out+gate", M, D, D, "gate"), ("xq", M, D, D, "bias"),
            ("xout+gate", M, D, D, "gate"), ("FF1+GELU", M, FF, D, "gelu"), ("FF2+gate", M, D, FF, "gate")]


def run(tokens, iters):
    sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
    import torch
    from kandinsky import _engine as E
    BF = torch.bfloat16
    epi = {"bias": E.EPI_BIAS, "bias_m": E.EPI_BIAS_M, "gelu": E.EPI_GELU, "gate": E.EPI_GATE}
    out_rows = {}
    for M in tokens:
        row = {}
        for name, m, n, k, e in shapes(M):
            a = torch.randn(m, k, device="cuda").to(BF)
            w = (torch.randn(n, k, device="cuda") * 0.05).to(BF)
            bias = torch.randn(m if e == "bias_m" else n, device="cuda").to(BF).float()
            resid = torch.randn(m, n, device="cuda").to(BF) if e == "gate" else None
            gate = torch.randn(n, device="cuda") if e == "gate" else None
            out = resid.clone() if e == "gate" else torch.empty(m, n, dtype=BF, device="cuda")
            call = lambda: E.gemm(a, w, bias, epi[e], resid=out if e == "gate" else None, gate=gate, out=out)
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                call()
            t.record()
            torch.cuda.synchronize()
            row[name] = s.elapsed_time(t) / iters * 1e3
        out_rows[M] = row
    print(json.dumps(out_rows))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", default="47616,23808,11904,5952,3328")
    ap.add_argument("--kernels", default="auto,4,8")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--env", default="", help="extra VAR=VALUE pairs for every run, comma separated")
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    toks = [int(v) for v in a.tokens.split(",")]
    if a.child:
        run(toks, a.iters)
        sys.exit(0)
    extra = dict(kv.split("=") for kv in a.env.split(",") if kv)
    names = [s[0] for s in shapes(1)]
    print("# us per call; block = sum of the seven; TF = TFLOP/s of the block")
    for kern in a.kernels.split(","):
        env = dict(os.environ, **extra)
        if kern != "auto":
            env["K5_GEMM_V1"] = kern
        r = subprocess.run([sys.executable, __file__, "--child", "--tokens", a.tokens, "--iters", str(a.iters)], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(kern, "FAILED", r.stderr[-400:])
            continue
        rows = json.loads(r.stdout.strip().splitlines()[-1])
        for M in toks:
            row = rows[str(M)]
            tot = sum(row.values())
            flop = sum(2.0 * m * n * k for _, m, n, k, _ in shapes(M))
            print(f"{kern:>5} M={M:6d} " + " ".join(f"{n}:{row[n]:7.1f}" for n in names) + f"  block:{tot:8.1f} us  {flop / tot / 1e6:7.1f} TF")
