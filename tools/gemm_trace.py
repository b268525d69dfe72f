"""Phase-level timeline of the 256x256 GEMM kernel (needs lib/variants/libk5_trace.so, built with -DK8_TRACE):
s_memtime stamps of block 0, one row per wave: LOAD-begin, reads-issued, reads-landed, barrier-a passed, MFMAs issued,
vmcnt wait done (then barrier-b)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["K5_LIB"] = os.path.join(ROOT, "kandinsky-5_amd", "lib", "variants", "libk5_trace.so")
sys.path.insert(0, os.path.join(ROOT, "kandinsky-5_amd"))
import torch
trace = torch.zeros(8 * 128, dtype=torch.int64, device="cuda")
os.environ["K5_GEMM_TRACE"] = "%x" % trace.data_ptr()
from kandinsky import _engine as E
M = N = K = 4096
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
for _ in range(3):
    E.gemm(a, w, None, E.EPI_BIAS)
torch.cuda.synchronize()
t = trace.cpu().reshape(8, 128)
t0 = int(t[t > 0].min())
names = ["L0", "rd_iss", "rd_done", "barA", "mfma_iss", "vm_done"]
for wv in (0, 4):
    print("wave", wv)
    row = [int(v) - t0 for v in t[wv]]
    for ph in range(8, 20):
        st = row[6 * ph:6 * ph + 6]
        nxt = row[6 * ph + 6]
        print(f"  ph{ph:2d} " + " ".join(f"{n}={v}" for n, v in zip(names, st)) + f" | load={st[2]-st[0]} barA={st[3]-st[2]} mfma={st[4]-st[3]} vmwait={st[5]-st[4]} barB={nxt-st[5]} total={nxt-st[0]}")
