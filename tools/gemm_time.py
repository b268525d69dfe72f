"""Time one GEMM shape through the C ABI:  python tools/gemm_time.py M N K [iters]   (K5_GEMM_V1 selects the kernel)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch
from kandinsky import _engine as E
M, N, K = (int(v) for v in sys.argv[1:4])
it = int(sys.argv[4]) if len(sys.argv) > 4 else 20
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    E.gemm(a, w, None, E.EPI_BIAS, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(it):
    E.gemm(a, w, None, E.EPI_BIAS, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / it
print(f"K5_GEMM_V1={os.environ.get('K5_GEMM_V1', '-')} M={M} N={N} K={K}: {ms:.4f} ms  {2*M*N*K/ms/1e9:.1f} TFLOP/s")
if os.environ.get("BLASLT"):
    for _ in range(3): torch.nn.functional.linear(a, w)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): torch.nn.functional.linear(a, w)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    print(f"hipBLASLt M={M} N={N} K={K}: {ms:.4f} ms  {2*M*N*K/ms/1e9:.1f} TFLOP/s")
