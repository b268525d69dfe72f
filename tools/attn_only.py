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
bound = None
if os.environ.get("BOUNDED"):   # RMS-normalised heads, as after norm_qk: |q.k| <= 64
    def rmsn(x):
        x = x.float().reshape(N, H, 64)
        return (x / x.pow(2).mean(-1, keepdim=True).sqrt()).reshape(N, H * 64).to(BF)
    q, k, bound = rmsn(q), rmsn(k), 64 * 1.05
def run():
    if os.environ.get("PRESCALED"):   # keys pre-multiplied by log2(e)/8 (what the engine's dense visual self-attention runs)
        E.check(E.lib().k5_attention_bf16_prescaled(q.data_ptr(), kc.data_ptr(), vt.data_ptr(), o.data_ptr(), H, N, N, q.stride(0),
                                                    kc.stride(0), vt.stride(0), o.stride(0), bound, E.stream_ptr()))
    else:
        E.attention(q, k, vt, H, out=o, score_bound=bound)


kc = (k.float() * 0.18033688011112042).to(BF) if os.environ.get("PRESCALED") else None
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    run()
torch.cuda.synchronize()
if os.environ.get("TIME"):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        run()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"attention N={N} bounded={bound is not None}: {ms:.3f} ms  {4.0 * N * N * 64 * H / ms / 1e9:.1f} TFLOP/s")
