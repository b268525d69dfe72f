import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch
from kandinsky import _engine as E
N, H = 47616, 28
BF = torch.bfloat16
def timeit(fn, iters=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
o = torch.empty(N, H * 64, dtype=BF, device="cuda")
for name, mk in (("zeros", lambda *s: torch.zeros(*s, device="cuda", dtype=BF)),
                 ("ones*0.1", lambda *s: torch.full(s, 0.1, device="cuda", dtype=BF)),
                 ("randn", lambda *s: torch.randn(*s, device="cuda").to(BF)),
                 ("randn*0.1", lambda *s: (torch.randn(*s, device="cuda") * 0.1).to(BF)),
                 ("rmsnormed randn (|q|=|k|=8)", None)):
    if mk is None:
        q = torch.randn(N, H, 64, device="cuda"); q = (q / q.pow(2).mean(-1, keepdim=True).sqrt()).reshape(N, -1).to(BF)
        k = torch.randn(N, H, 64, device="cuda"); k = (k / k.pow(2).mean(-1, keepdim=True).sqrt()).reshape(N, -1).to(BF)
        vt = torch.randn(H * 64, N, device="cuda").to(BF)
    else:
        q, k, vt = mk(N, H * 64), mk(N, H * 64), mk(H * 64, N)
    ms = timeit(lambda: E.attention(q, k, vt, H, out=o))
    print(f"{name:32s} {ms:8.3f} ms {4.0 * N * N * 64 * H / ms / 1e9:8.1f} TFLOP/s", flush=True)
    if mk is None:
        ref = o.clone()
        ms = timeit(lambda: E.attention(q, k, vt, H, out=o, score_bound=64 * 1.05))
        print(f"{name + ' BOUNDED':32s} {ms:8.3f} ms {4.0 * N * N * 64 * H / ms / 1e9:8.1f} TFLOP/s  max|diff| vs online-max {(o.float() - ref.float()).abs().max().item():.3g}", flush=True)
