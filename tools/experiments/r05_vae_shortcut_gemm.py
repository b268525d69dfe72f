import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "kandinsky-5_amd"))
from kandinsky import _engine as E
M, N, K = 17 * 512 * 768, 128, 256
a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16); b = torch.randn(N, device="cuda")
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for _ in range(3): E.gemm(a, w, b, E.EPI_BIAS, out=out)
torch.cuda.synchronize(); s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): E.gemm(a, w, b, E.EPI_BIAS, out=out)
t.record(); torch.cuda.synchronize()
ms = s.elapsed_time(t) / 10
print(os.environ.get("K5_LIB", "default")[-20:], f"{ms:.3f} ms  {(M * K * 2 + M * N * 2) / ms / 1e9:.2f} TB/s  {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s")
