"""One GEMM shape through the C ABI (for rocprofv3 counter runs):  python tools/gemm_only.py M N K [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch
from kandinsky import _engine as E
M, N, K = (int(v) for v in sys.argv[1:4])
it = int(sys.argv[4]) if len(sys.argv) > 4 else 2
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(it):
    E.gemm(a, w, None, E.EPI_BIAS, out=out)
torch.cuda.synchronize()
