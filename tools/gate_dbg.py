import os, sys
sys.path.insert(0, "/root/repo/kandinsky-5_amd")
import torch
from kandinsky import _engine as E
BF = torch.bfloat16
torch.manual_seed(0)
def run(M, N, K, tile, inplace, bias):
    a = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    rnd = lambda x: x.to(BF).float()
    acc = a.float() @ w.float().t()
    b = rnd(torch.randn(N, device="cuda") * 0.1) if bias else None
    resid = rnd(torch.randn(M, N, device="cuda")); gate = torch.randn(N, device="cuda")
    inner = rnd(acc + (b if bias else 0)); ref = rnd(resid + gate * inner)
    r = resid.to(BF)
    out = r if inplace else torch.empty_like(r)
    got = E.gemm(a, w, b, E.EPI_GATE, resid=r, gate=gate, out=out, kernel=4, token_tile=tile)
    torch.cuda.synchronize()
    err = (got.float() - ref).abs()
    tol = 2e-3 + 3 * 2.0 ** -7 * ref.abs() + gate.abs() * 2.0 ** -7 * inner.abs()
    bad = (err > tol)
    rows = bad.any(dim=1).nonzero().flatten()
    cols = bad.any(dim=0).nonzero().flatten()
    return int(bad.sum()), (rows.min().item(), rows.max().item(), cols.min().item(), cols.max().item()) if bad.any() else None
for (M, N, K) in ((17921, 264, 384), (7937, 960, 512), (17920, 264, 384), (7936, 960, 512), (7944, 960, 512), (7937, 1024, 512)):
    for tile in (256, 192, 128):
        for inplace in (True, False):
            for bias in (False, True):
                print(M, N, K, "tile", tile, "inplace" if inplace else "outofpl", "bias" if bias else "nobias", run(M, N, K, tile, inplace, bias), flush=True)
