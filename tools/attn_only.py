import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch
from kandinsky import _engine as E
N, H = 47616, 28
BF = torch.bfloat16
q, k = torch.randn(N, H * 64, device="cuda").to(BF), torch.randn(N, H * 64, device="cuda").to(BF)
vt = torch.randn(H * 64, N, device="cuda").to(BF)
o = torch.empty(N, H * 64, dtype=BF, device="cuda")
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    E.attention(q, k, vt, H, out=o)
torch.cuda.synchronize()
