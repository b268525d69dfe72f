"""k8 (256-row persistent) vs 128x128 kernel on small token counts:  python tools/gemm_small.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1:
    for name, env in (("auto   ", {}), ("128x128", {"K5_GEMM_V1": "2"}), ("q4     ", {"K5_GEMM_V1": "5"}), ("k8     ", {"K5_GEMM_V1": "8"}), ("w4     ", {"K5_GEMM_V1": "4"})):
        out = subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, **env), capture_output=True, text=True).stdout
        print(name, out.strip())
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch
from kandinsky import _engine as E
BF = torch.bfloat16
res = []
for (M, N, K) in ((3328, 3584, 1792), (1792, 3328, 1792), (3328, 1792, 1792), (3328, 7168, 1792), (3328, 1792, 7168)):
    if True:
        a, w = torch.randn(M, K, device="cuda").to(BF), (torch.randn(N, K, device="cuda") * 0.05).to(BF)
        out = torch.empty(M, N, dtype=BF, device="cuda")
        for _ in range(3): E.gemm(a, w, None, E.EPI_BIAS, out=out)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): E.gemm(a, w, None, E.EPI_BIAS, out=out)
        e.record(); torch.cuda.synchronize()
        res.append(f"{M}x{N}x{K}:{s.elapsed_time(e) / 20 * 1e3:.0f}us")
print(" ".join(res))
